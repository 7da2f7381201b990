#!/usr/bin/env python
"""Per-step kernel timeline out of a rocprofv3 kernel trace of the default bench command (tools/timeline.sh):
python tools/timeline.py gpurun_out/TAG/kernel_trace.csv [--dump]
Finds the graph-replayed steps (groups of kernels that repeat), prints for the median step: span, time with 1 / >= 2
kernels in flight, idle, and with --dump every kernel (start, end, queue)."""
import csv
import sys

csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
grid = lambda r: int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_X"]), 1)
# steps: delimited by raster_bwd kernels on views/2 launches; a step holds exactly two of them in the default command
bw = [i for i, r in enumerate(rows) if "raster_bwd_kernel" in r["Kernel_Name"]]
sf = [i for i, r in enumerate(rows) if "shade_fwd_kernel" in r["Kernel_Name"]]
# pair up consecutive shade_fwd launches (two micro-batches); a step = [first shade_fwd of a pair, next pair's first)
steps = []
for a, b in zip(sf[0::2], sf[2::2]):
    seg = rows[a:b]
    t_first = int(seg[0]["Start_Timestamp"])
    seg = [r for r in seg if int(r["Start_Timestamp"]) - t_first < 6_000_000]   # (drop what follows a pause)
    if sum("raster_bwd_kernel" in r["Kernel_Name"] for r in seg) == 2:
        steps.append(seg)
print("%d steps with two micro-batches found" % len(steps))
stats = []
for seg in steps:
    t0 = int(seg[0]["Start_Timestamp"])
    ev = [((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3) for r in seg]
    end = max(e for s, e in ev)
    pts = sorted([(s, 1) for s, e in ev] + [(e, -1) for s, e in ev])
    depth, last, one, two = 0, 0.0, 0.0, 0.0
    for t, d in pts:
        if depth == 1: one += t - last
        elif depth >= 2: two += t - last
        depth += d; last = t
    stats.append((end, one, two, end - one - two, seg, ev))
# the graph replays of the timed region: kernels of both micro-batches in flight together (the warm-up steps are eager, the
# instrumented pass after the timed region runs the micro-batches on one stream)
replays = [x for x in stats if x[2] > 0.5 * x[0]]
stats = sorted(replays or stats, key=lambda x: x[0])
print("%d of them with two kernels in flight most of the time (graph replays / two-stream steps)" % len(replays))
end, one, two, idle, seg, ev = stats[len(stats) // 2]
print("median step: kernels span %.1f us; 1 kernel in flight %.1f us, >= 2 in flight %.1f us, idle %.1f us; sum of kernel "
      "durations %.1f us" % (end, one, two, idle, sum(e - s for s, e in ev)))
if "--dump" in sys.argv:
    for r, (s, e) in zip(seg, ev):
        print("%8.1f %8.1f %7.1f  q%-2s s%-3s wg%-6d v%-3s %s" % (s, e, e - s, r["Queue_Id"], r["Stream_Id"], grid(r), r["VGPR_Count"], short(r)))
