#!/bin/bash
# Round-end check, run ON THE GPU BOX:  bash tools/final_check.sh TAG
# full gpu suite + smoke + the default bench line (taken with the stamped PMC records present) + rocprofv3 kernel stats of the
# DEFAULT command (two 4-view micro-batches, graph replay) + the 2-rank shared-GPU functional line.
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -o kt -- python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 2 > $OUT/kt.log 2>&1
python - "$OUT" <<'PY'
import csv, sys
csv.field_size_limit(1 << 30)
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/kt/kt_kernel_stats.csv", newline="")))
with open(out + "/default_kernel_stats.csv", "w") as o:
    o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for r in rows[:40]:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:100]
        o.write('"%s",%s,%s,%s,%s,%s,%s\n' % (n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]))
PY
rm -rf $OUT/kt
timeout 300 python bench.py --gpus 2 --share-gpu --no-cpu-baseline --no-secondary > $OUT/bench_2ranks_shared_gpu.json 2> $OUT/bench2.err
cat $OUT/pytest.log; tail -2 $OUT/smoke.log; head -c 600 $OUT/bench.json; echo; head -8 $OUT/default_kernel_stats.csv; head -c 300 $OUT/bench_2ranks_shared_gpu.json
