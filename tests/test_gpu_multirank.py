"""GPU (one device): the N-rank path of bench.py before an 8-GPU node is available.

  * two ranks on ONE GPU (`--gpus 2 --share-gpu`): RCCL refuses two ranks on one device, so the ranks exchange gradients
    over gloo with device tensors -- everything else (launcher, per-rank inputs, HIP-graph capture and replay in each rank,
    GradSync's launch_all / wait overlap, max-over-ranks timing, the one JSON line) is the path the driver's SCALE runs take;
  * one rank with a LIVE RCCL communicator (`--force-dist`): process group "nccl" of world size 1, the gradient exchange
    issued as real RCCL collectives, the step captured into a HIP graph and replayed while the communicator and its
    watchdog thread exist (capture_error_mode thread_local) -- capture must not fall back to eager;
  * `--gpus 2` on a 1-GPU box without --share-gpu fails loudly.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
pytestmark = pytest.mark.gpu
SMALL = ["--views", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--grad-floats", "1000000"]


def _run(args, timeout=900):
    e = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e)


def _line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_run_the_whole_step_with_graph_capture():
    line = _line(_run(["--gpus", "2", "--share-gpu"] + SMALL))
    c = line["config"]
    assert line["n_gpus"] == 2 and c["world_size"] == 2 and c["ranks_share_gpu"] is True
    assert c["dist_backend"] == "gloo" and c["rccl_world_size"] == 0
    assert c["launch"].startswith("hip_graph_replay"), c["launch"]
    assert c["grad_exchange_bytes_per_step"] == 4 * (1_000_000 + 250_000 * 3)
    assert line["value"] > 0 and c["intersections_per_view"] > 1e5
    assert "NOT a scaling measurement" in line["note"]


def test_graph_capture_next_to_a_live_rccl_communicator():
    line = _line(_run(["--force-dist"] + SMALL))
    c = line["config"]
    assert line["n_gpus"] == 1 and c["dist_backend"] == "nccl" and c["rccl_world_size"] == 1
    assert c["launch"].startswith("hip_graph_replay"), "capture fell back to eager next to RCCL: " + c["launch"]
    assert c["grad_exchange_ms_alone"] is not None and c["grad_exchange_ms_alone"] > 0


def test_gpus_2_on_a_one_gpu_box_fails_loudly():
    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one GPU visible")
    r = _run(["--gpus", "2"] + SMALL)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "one GPU per rank is required" in r.stderr, r.stderr[-3000:]
