"""CPU: the view-parallel path (sharding + bucketed reduce-scatter/all-gather gradient sync) with
world_size 2 over gloo; the same code runs over RCCL on the GPUs."""
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_views_round_robin():
    from goliath_amd import parallel

    assert parallel.shard_views(8, 1, 4) == [1, 5]
    assert parallel.shard_views(3, 2, 8) == [2] and parallel.shard_views(3, 5, 8) == []
    assert sorted(sum((parallel.shard_views(10, r, 3) for r in range(3)), [])) == list(range(10))
    b = parallel.shard_batch({"K": torch.arange(12).reshape(4, 3), "name": "x"}, 1, 2)
    assert b["K"].tolist() == [[3, 4, 5], [9, 10, 11]] and b["name"] == "x"
    # no process group: sync is the identity
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    parallel.GradSync([p]).sync()
    assert p.grad.tolist() == [2.0, 2.0, 2.0]


def test_gradsync_world2_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(HERE, "_dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_config3_world8_gloo():
    """BASELINE config 3 on 8 ranks: 8 views / 8 ranks, and 4 (and 3) views / 8 ranks -- ranks without a
    view issue the same collectives; gradients and the global norm equal the single-process full-batch ones."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(HERE, "_dist_worker8.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env)
    assert r.returncode == 0 and "DIST8_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
