"""world_size-2 gloo worker for tests/test_parallel.py (launched with torch.distributed.run)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goliath_amd import parallel  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.manual_seed(0)  # same model on every rank
    model = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.Tanh(), torch.nn.Linear(13, 3))
    B = 6
    g = torch.Generator().manual_seed(1)
    batch = {"x": torch.randn(B, 7, generator=g), "y": torch.randn(B, 3, generator=g), "ids": list(range(B)),
             "cfg": "shared"}
    mine = parallel.shard_batch(batch)
    assert mine["ids"] == list(range(rank, B, world)) and mine["cfg"] == "shared"
    # reference: the whole batch on one process
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.Tanh(), torch.nn.Linear(13, 3))
    ((ref(batch["x"]) - batch["y"]) ** 2).sum().div(B).backward()

    def check(tag):
        for p, q in zip(model.parameters(), ref.parameters()):
            assert torch.allclose(p.grad, q.grad, atol=1e-6), (tag, (p.grad - q.grad).abs().max())

    # (1) plain use: backward first, then sync() adopts the existing .grad tensors.  Tiny buckets -> several collectives
    # per-view loss summed over this rank's views, scaled so that the rank-average equals the full-batch mean
    loss = ((model(mine["x"]) - mine["y"]) ** 2).sum() * world / B
    loss.backward()
    sync = parallel.GradSync(model.parameters(), bucket_bytes=256)
    assert len(sync.buckets) > 1
    sync.sync()
    check("adopt")
    # (2) steady state: gradients are views of the persistent flat buckets, the hooks launch each bucket's collectives
    # DURING backward (overlap), finish() only waits
    for step in range(2):
        sync.zero_grad()
        ptrs = [p.grad.data_ptr() for p in model.parameters()]
        loss = ((model(mine["x"]) - mine["y"]) ** 2).sum() * world / B
        loss.backward()
        assert all(sync._launched), "every bucket was launched by its hooks before finish()"
        sync.finish()
        assert ptrs == [p.grad.data_ptr() for p in model.parameters()], "grads stayed views of the flat buffers"
        check(f"overlap{step}")
    # (3) a parameter that gets no gradient (p.grad is None) + grads replaced behind our back (set_to_none)
    sync.close()
    extra = torch.nn.Parameter(torch.ones(5))
    sync2 = parallel.GradSync(list(model.parameters()) + [extra], bucket_bytes=1 << 20)
    for p in model.parameters():
        p.grad = None
    extra.grad = None
    loss = ((model(mine["x"]) - mine["y"]) ** 2).sum() * world / B
    loss.backward()
    sync2.sync()
    check("unused")
    assert extra.grad is not None and float(extra.grad.abs().max()) == 0.0
    sync2.zero_grad()
    for p in model.parameters():
        p.grad = None                      # what optimizer.zero_grad(set_to_none=True) does
    loss = ((model(mine["x"]) - mine["y"]) ** 2).sum() * world / B
    loss.backward()
    sync2.finish()
    check("set_to_none")
    # (4) launch order is the bucket order on every rank, whatever the local autograd order: rank 1 leaves the LAST
    # parameter (= bucket 0) unused, so its hooks can launch nothing and finish() issues the whole sequence, while rank 0
    # launches everything from its hooks -- the collectives (all of different sizes) must still pair up
    sync2.close()
    extra2 = torch.nn.Parameter(torch.ones(5))
    sync3 = parallel.GradSync(list(model.parameters()) + [extra2], bucket_bytes=16)
    assert len(sync3.buckets) >= 3 and len(sync3.buckets[0]) == 1 and sync3.buckets[0][0] is extra2
    sync3.zero_grad()
    loss = ((model(mine["x"]) - mine["y"]) ** 2).sum() * world / B
    if rank == 0:
        loss = loss + extra2.sum() * 2.0
    loss.backward()
    if rank == 0:
        assert all(sync3._launched)
    else:
        assert not any(sync3._launched), "bucket 0 never became ready here: nothing may overtake it"
    sync3.finish()
    check("ordered")
    assert torch.allclose(extra2.grad, torch.full((5,), 2.0 / world)), extra2.grad
    # (5) a second backward before finish() must not silently add local gradients to averaged buckets
    sync3.zero_grad()
    loss = ((model(mine["x"]) - mine["y"]) ** 2).sum() * world / B + extra2.sum()
    loss.backward(retain_graph=True)
    try:
        loss.backward()
        raise AssertionError("second backward before finish() went unnoticed")
    except RuntimeError as e:
        assert "no_sync" in str(e)
    sync3.finish()
    # (6) accumulation under no_sync(): two half-batches == the full batch
    sync3.zero_grad()
    half = mine["x"].shape[0] // 2
    with sync3.no_sync():
        (((model(mine["x"][:half]) - mine["y"][:half]) ** 2).sum() * world / B).backward()
        assert not any(sync3._launched)
    (((model(mine["x"][half:]) - mine["y"][half:]) ** 2).sum() * world / B + 0.0 * extra2.sum()).backward()
    assert all(sync3._launched)
    sync3.finish()
    check("no_sync")
    # (7) launch_all() / wait() split (what bench.py uses to overlap the exchange with the next step's compute)
    sync3.zero_grad()
    (((model(mine["x"]) - mine["y"]) ** 2).sum() * world / B).backward()
    sync3.launch_all()
    sync3.wait()
    check("split")
    gn = parallel.global_grad_norm(model.parameters())
    gathered = [torch.zeros(()) for _ in range(world)]
    dist.all_gather(gathered, gn)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    m = parallel.sync_mean(torch.tensor(float(rank)))
    assert abs(float(m) - (world - 1) / 2) < 1e-6
    # (8) the reference loop's explosion rollback + NaN scrub + clip (train.py:170-214), view-parallel: every rank takes the
    # same decision from the all-reduced loss, and the clip uses the norm of the AVERAGED gradients
    guard = parallel.ExplosionGuard()
    assert not guard.exploded(torch.tensor(1.0 + rank))                 # first step: the history starts at inf
    assert not guard.exploded(torch.tensor(1.0e4 if rank == 0 else 1.0))   # ... and, as in the reference, the 10x rule only
    for _ in range(32):                                                    # bites once 32 losses have pushed the inf out
        assert not guard.exploded(torch.tensor(1.2))
    # only ONE rank sees a huge loss: the mean is > 10 x the history on every rank -> all roll back together
    assert guard.exploded(torch.tensor(1.0e4 if rank == 0 else 1.0))
    assert guard.exploded(torch.tensor(float("nan") if rank == world - 1 else 1.0))
    assert not guard.exploded(torch.tensor(1.3))                        # the history was not polluted by the exploded steps
    sync3.zero_grad()
    (((model(mine["x"]) - mine["y"]) ** 2).sum() * world / B * 50.0).backward()
    norm = parallel.finish_scrub_and_clip(sync3, list(model.parameters()), max_norm=1.0)
    after = parallel.global_grad_norm(model.parameters())
    gathered = [torch.zeros(()) for _ in range(world)]
    dist.all_gather(gathered, norm)
    assert all(torch.equal(gathered[0], t) for t in gathered) and float(norm) > 1.0
    assert abs(float(after) - 1.0) < 1e-4, float(after)
    dist.barrier()
    if rank == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
