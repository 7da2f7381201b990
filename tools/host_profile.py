"""tools/host_profile.py -- host-side cost of issuing one eager step (run on the GPU box): cProfile over K steps of bench.step,
streams not synchronised inside the loop, so the numbers are pure issue time."""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

cfg = dict(bench.CFG)
dev = torch.device("cuda", 0)
t = bench.make_step_inputs(cfg, dev, 0, 2)
for _ in range(3):
    bench.step(t, cfg, 1)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    bench.step(t, cfg, 1)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"issue time per step {1e3 * (t1 - t0) / K:.3f} ms (with profiler); drain {1e3 * (t2 - t1):.2f} ms")
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
