"""Experiment: two half-batches of 4 views on two HIP streams vs one batch of 8 on one stream."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = dict(bench.CFG, views_per_gpu=8 // NS)
dev = torch.device("cuda")
ts = [bench.make_inputs(cfg, dev, rank=r) for r in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
def step2():
    for t, s in zip(ts, streams):
        with torch.cuda.stream(s):
            bench.step(t, cfg, 1)
for _ in range(3): step2()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step2()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(NS, "streams x", 8 // NS, "views:", round(80 / dt, 1), "views/s")
