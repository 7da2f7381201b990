#!/bin/bash
# Same-box interleaved A/B of bench.py argument sets (default command: graph replay, two streams):
#   bash tools/ab_args.sh TAG REPS "args of arm 1" "args of arm 2" ...     table -> gpurun_out/TAG/ab.txt
TAG=${1:?tag}; REPS=${2:?reps}; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for r in $(seq $REPS); do
  i=0
  for a in "$@"; do
    i=$((i+1))
    python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 $a 2>$OUT/err_$i.txt | python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l)
    print('arm$i [$a] value %.1f ms_med %.4f launch %s' % (d['value'], d['windows']['ms_per_step_median'], d['config'].get('launch')))
except Exception as e:
    print('arm$i [$a] FAILED', e, open('$OUT/err_$i.txt').read()[-1500:])
" | tee -a $OUT/ab.txt
  done
done
