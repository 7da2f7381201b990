#!/bin/bash
# Same-box interleaved A/B of two builds of the library:  bash tools/ab_libs.sh TAG LIB_A LIB_B [reps]
#   ("" = the default goliath_amd/lib/libgoliath_hip.so).  Per repetition and arm: the default command (graph replay, two
#   streams; headline value) and the --micro 1 eager command (8 views per launch; per-call medians), then one rocprofv3
#   kernel-trace of the micro1 command per arm.  Table -> gpurun_out/TAG/ab.txt
TAG=${1:?tag}; A=$2; B=$3; REPS=${4:-3}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() {  # arm lib mode args...
  local arm=$1 lib=$2 mode=$3; shift 3
  if [ -n "$lib" ]; then export GOLIATH_HIP_LIB=$PWD/$lib; else unset GOLIATH_HIP_LIB; fi
  python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['kernels_ms_per_call']
print('$arm $mode value %.1f ms_med %.4f | ' % (d['value'], d['windows']['ms_per_step_median']) + ' '.join('%s %.4f' % (n.replace('gol_',''), v) for n, v in k.items()))
" | tee -a $OUT/ab.txt
}
for r in $(seq $REPS); do
  run A "$A" default; run B "$B" default
  run A "$A" micro1 --micro 1 --no-graph; run B "$B" micro1 --micro 1 --no-graph
done
for arm in A B; do
  lib=$A; [ $arm = B ] && lib=$B
  if [ -n "$lib" ]; then export GOLIATH_HIP_LIB=$PWD/$lib; else unset GOLIATH_HIP_LIB; fi
  echo "== kernel trace, arm $arm ($lib)" | tee -a $OUT/ab.txt
  bash tools/quick_kt.sh $TAG/kt_$arm | tee -a $OUT/ab.txt
done
