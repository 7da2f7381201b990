"""world_size-8 gloo worker for tests/test_parallel.py::test_config3_world8_gloo: BASELINE config 3 as written -- the batch
of V views sharded over 8 ranks (b % 8 == rank; /root/reference/ca_code/models/rgca.py:119-138 is the view loop that
shards), one gradient exchange per step -- with V = 8 (one view per rank) and V = 4 (ranks 4-7 own NO view: they skip
forward / backward but must issue the same collectives, in the same order, or the job hangs)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goliath_amd import parallel  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 8
    torch.manual_seed(0)  # same "decoder" on every rank
    model = torch.nn.Sequential(torch.nn.Linear(9, 32), torch.nn.LeakyReLU(0.2), torch.nn.Linear(32, 16),
                                torch.nn.LeakyReLU(0.2), torch.nn.Linear(16, 5))
    sync = parallel.GradSync(model.parameters(), bucket_bytes=512)     # several buckets of different sizes
    assert len(sync.buckets) >= 3
    guard = parallel.ExplosionGuard()
    for V in (8, 4, 8, 3):
        g = torch.Generator().manual_seed(100 + V)
        batch = {"x": torch.randn(V, 9, generator=g), "y": torch.randn(V, 5, generator=g), "view": list(range(V))}
        mine = parallel.shard_batch(batch)
        assert mine["view"] == [v for v in range(V) if v % world == rank]
        torch.manual_seed(0)
        ref = torch.nn.Sequential(torch.nn.Linear(9, 32), torch.nn.LeakyReLU(0.2), torch.nn.Linear(32, 16),
                                  torch.nn.LeakyReLU(0.2), torch.nn.Linear(16, 5))
        ((ref(batch["x"]) - batch["y"]) ** 2).sum().div(V).backward()     # the whole batch on one process
        sync.zero_grad()
        if mine["view"]:
            # per-view losses summed over this rank's views, scaled so that the rank AVERAGE is the full-batch mean
            loss = ((model(mine["x"]) - mine["y"]) ** 2).sum() * world / V
            loss.backward()
            assert all(sync._launched), "a rank with views launches every bucket from its hooks"
        else:
            loss = torch.zeros(())
            assert not any(sync._launched)   # no backward: nothing launched yet; finish() issues the same sequence
        norm = parallel.finish_scrub_and_clip(sync, model.parameters(), max_norm=1e9)
        for p, q in zip(model.parameters(), ref.parameters()):
            assert torch.allclose(p.grad, q.grad, atol=1e-6), (V, rank, float((p.grad - q.grad).abs().max()))
        ref_norm = torch.stack([q.grad.pow(2).sum() for q in ref.parameters()]).sum().sqrt()
        assert abs(float(norm) - float(ref_norm)) < 1e-5 * float(ref_norm)   # the same global norm on every rank
        assert guard.exploded(loss) is False                                 # (one more collective, every rank)
    sync.close()
    dist.barrier()
    if rank == 0:
        print("DIST8_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
