#!/usr/bin/env python
"""Which framework op launches which small kernel in one eager bench step:  python tools/op_trace.py [--unfused-projection]
(torch.profiler; prints the step's device kernels in launch order with the ATen / autograd op that issued each)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

args = bench.parse_args(["--no-secondary", "--no-cpu-baseline", "--no-graph"] + sys.argv[1:])
cfg = dict(bench.CFG, views_per_gpu=8, fused_projection=not args.unfused_projection)
dev = torch.device("cuda")
t = bench.make_step_inputs(cfg, dev, 0, 2)
for _ in range(3):
    bench.step(t, cfg, 1)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    bench.step(t, cfg, 1)
    torch.cuda.synchronize()
ev = prof.events()
kern = [e for e in ev if e.device_type.name in ("CUDA", "PrivateUse1") or "cuda" in str(e.device_type).lower()]
cpu = [e for e in ev if e not in kern]
kern.sort(key=lambda e: e.time_range.start)
for k in kern:
    # innermost CPU op whose launch correlates (same correlation id is not exposed: use the enclosing op at launch time)
    print("%9.1f us  %-60s" % (k.time_range.elapsed_us(), k.name[:60]))
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
for e in ev:
    if e.name in ("aten::copy_", "aten::cat", "aten::add", "aten::fill_"):
        chain, p = [], e
        while p is not None:
            chain.append(p.name)
            p = p.cpu_parent
        print(" <- ".join(chain))
