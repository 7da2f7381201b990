#!/bin/bash
# kernel-time table of a secondary workload:  bash tools/quick_kt_workload.sh TAG WORKLOAD   (run on the GPU box)
TAG=${1:-q}; WL=${2:-urhand}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -o kt -- python $ROOT/bench.py --workload $WL --no-cpu-baseline --steps 10 --warmup 2 > $OUT/kt.log 2>&1)
python - "$OUT" <<'PY'
import csv, sys
csv.field_size_limit(1 << 30)
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/kt/kt_kernel_stats.csv")))
for r in rows[:12]:
    print("%-60s %5s %10.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $OUT/kt
