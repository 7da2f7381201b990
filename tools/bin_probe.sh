#!/bin/bash
# bash tools/bin_probe.sh TAG [LIB]   per-kernel split of gol_bin_sort at 250k and 1M (rocprofv3 kernel trace of tools/bin_probe.py)
TAG=${1:?tag}; LIB=$2; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ -n "$LIB" ]; then export GOLIATH_HIP_LIB=$PWD/$LIB; else unset GOLIATH_HIP_LIB; fi
for cfgs in "500 0.0" "1024 -0.9"; do
  set -- $cfgs
  python tools/bin_probe.py --slab $1 --scale-shift $2 2>/dev/null | tee -a $OUT/probe.txt
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -o kt -- python tools/bin_probe.py --slab $1 --scale-shift $2 --iters 10 > $OUT/kt.log 2>&1
  python - "$OUT" <<'PY' | tee -a $OUT/probe.txt
import csv, sys
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1] + "/kt/kt_kernel_stats.csv")))
keep = ("count_lds", "scan_kernel", "scatter_lds", "sort_kernel", "sort_big", "sort_mid", "count_kernel", "scatter_kernel", "fillBuffer")
tot = 0.0
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if any(k in n for k in keep):
        print("   %-40s %4s calls %9.1f us" % (n[:40], r["Calls"], float(r["AverageNs"]) / 1e3))
        if "fillBuffer" not in n:
            tot += float(r["AverageNs"]) / 1e3
print("   sum of the binning kernels %.1f us" % tot)
PY
  rm -rf $OUT/kt
done
