#!/bin/bash
# quick kernel-time table of the micro1 bench command:  bash tools/quick_kt.sh TAG [extra bench args]
TAG=${1:-q}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -o kt -- python bench.py --micro 1 --no-graph --no-cpu-baseline --no-secondary --steps 10 --warmup 2 "$@" > $OUT/kt.log 2>&1
python - "$OUT" <<'PY'
import csv, sys
csv.field_size_limit(1 << 30)
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/kt/kt_kernel_stats.csv")))
with open(out + "/kernel_stats.csv", "w") as o:
    o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for r in rows:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:100]
        o.write('"%s",%s,%s,%s,%s,%s,%s\n' % (n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]))
for r in rows[:14]:
    print("%-52s %4s %10.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:52], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $OUT/kt
