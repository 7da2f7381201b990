#!/bin/bash
# kernel trace of the DEFAULT (graph, two micro-batches) bench command:  bash tools/timeline.sh TAG [extra bench args]
# leaves gpurun_out/TAG/kernel_trace.csv (analyse with tools/timeline.py)
TAG=${1:-tl}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace -f csv -d $OUT/kt -o kt -- python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2 "$@" > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name 'kt_kernel_trace.csv' | head -1) $OUT/kernel_trace.csv
rm -rf $OUT/kt
