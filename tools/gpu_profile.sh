#!/bin/bash
# Standard measurement suite, run ON THE GPU BOX from the repo root:  bash tools/gpu_profile.sh TAG [quick]
# Writes gpurun_out/TAG/: bench.json (default command), bench_micro1.json (8 views per launch, eager),
# kernel_stats.csv (rocprofv3 --kernel-trace --stats of the micro1 command), pmc_traffic.csv (FETCH_SIZE /
# WRITE_SIZE, separate passes), pmc_sq.csv (instruction / cycle counters, separate passes), head.txt (commit).
# Copy what should be judged into profiles/ (tracked).
TAG=${1:-prof}; QUICK=${2:-}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cat .git_head 2>/dev/null > $OUT/head.txt
python -c "from goliath_amd import build; print(build.source_digest())" > $OUT/csrc_sha16.txt
M1="python bench.py --micro 1 --no-graph --no-cpu-baseline --no-secondary"
python bench.py > $OUT/bench.json 2> $OUT/bench.err
$M1 --steps 20 --warmup 3 > $OUT/bench_micro1.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -o kt -- $M1 --steps 10 --warmup 2 > $OUT/kt.log 2>&1
cp $OUT/kt/kt_kernel_stats.csv $OUT/kernel_stats_raw.csv 2>/dev/null
python - "$OUT" <<'PY'
import csv, re, sys
out = sys.argv[1]
csv.field_size_limit(1 << 30)
try:
    rows = list(csv.DictReader(open(out + "/kernel_stats_raw.csv", newline="")))
    with open(out + "/kernel_stats.csv", "w") as o:
        o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:100]
            o.write('"%s",%s,%s,%s,%s,%s,%s\n' % (n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]))
except Exception as e:
    print("kernel stats:", e)
PY
rm -rf $OUT/kt $OUT/kernel_stats_raw.csv
if [ -z "$QUICK" ]; then
  S="--steps 3 --warmup 1"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -f csv -d $OUT/pmc_$c -o p -- $M1 $S > $OUT/pmc_$c.log 2>&1
  done
  python tools/pmc_summary.py $OUT/pmc_traffic.csv $OUT/pmc_FETCH_SIZE/p_counter_collection.csv $OUT/pmc_WRITE_SIZE/p_counter_collection.csv > /dev/null
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_sqA -o p -- $M1 $S > $OUT/pmc_sqA.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU -f csv -d $OUT/pmc_sqB -o p -- $M1 $S > $OUT/pmc_sqB.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_sq.csv $OUT/pmc_sqA/p_counter_collection.csv $OUT/pmc_sqB/p_counter_collection.csv > /dev/null
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sqA $OUT/pmc_sqB
fi
ls -la $OUT
