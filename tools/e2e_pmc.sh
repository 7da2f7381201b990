#!/bin/bash
# PMC record of the hot path at the reference-native size (1,048,576 Gaussians, 8 views per launch; rgca.py:385-386), run ON
# THE GPU BOX:  bash tools/e2e_pmc.sh TAG
# Writes gpurun_out/TAG/: e2e_kernel_stats.csv (rocprofv3 --kernel-trace --stats), e2e_pmc_traffic.csv (FETCH_SIZE / WRITE_SIZE,
# separate passes), e2e_pmc_sq.csv (instruction / cycle counters, two passes), bench_e2e.json (the line itself, 60 steps, with
# --segments).  tools/make_profile_record.py --e2e TAG turns them into the stamped profiles/traffic_e2e.json / valu_e2e.json.
TAG=${1:-e2e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -c "from goliath_amd import build; print(build.source_digest())" > $OUT/csrc_sha16.txt
M="python bench.py --workload e2e --no-cpu-baseline --steps 3 --warmup 2"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -o kt -- $M > $OUT/kt.log 2>&1
python - "$OUT" <<'PY'
import csv, sys
csv.field_size_limit(1 << 30)
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/kt/kt_kernel_stats.csv", newline="")))
with open(out + "/e2e_kernel_stats.csv", "w") as o:
    o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for r in rows:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:100]
        o.write('"%s",%s,%s,%s,%s,%s,%s\n' % (n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]))
PY
rm -rf $OUT/kt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -f csv -d $OUT/pmc_$c -o p -- $M > $OUT/pmc_$c.log 2>&1
done
python tools/pmc_summary.py $OUT/e2e_pmc_traffic.csv $OUT/pmc_FETCH_SIZE/p_counter_collection.csv $OUT/pmc_WRITE_SIZE/p_counter_collection.csv > /dev/null
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_sqA -o p -- $M > $OUT/pmc_sqA.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU -f csv -d $OUT/pmc_sqB -o p -- $M > $OUT/pmc_sqB.log 2>&1
python tools/pmc_summary.py $OUT/e2e_pmc_sq.csv $OUT/pmc_sqA/p_counter_collection.csv $OUT/pmc_sqB/p_counter_collection.csv > /dev/null
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sqA $OUT/pmc_sqB
python bench.py --workload e2e --no-cpu-baseline --steps 60 --warmup 3 --segments > $OUT/bench_e2e.json 2> $OUT/bench_e2e.err
ls -la $OUT
