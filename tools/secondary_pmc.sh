#!/bin/bash
# PMC records of the secondary workloads (mvp / urhand / sg), run ON THE GPU BOX:  bash tools/secondary_pmc.sh TAG
# Writes gpurun_out/TAG/secondary_{workload}_pmc_sq.csv; tools/make_profile_record.py --secondary TAG turns them into the
# stamped profiles/valu_secondary.json that bench.py's secondary lines use for their instruction-issue rooflines.
TAG=${1:-sec}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -c "from goliath_amd import build; print(build.source_digest())" > $OUT/csrc_sha16.txt
for w in mvp urhand sg; do
  M="python bench.py --workload $w --no-cpu-baseline --steps 2 --warmup 1"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -f csv -d $OUT/a -o p -- $M > $OUT/a.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU -f csv -d $OUT/b -o p -- $M > $OUT/b.log 2>&1
  python tools/pmc_summary.py $OUT/secondary_${w}_pmc_sq.csv $OUT/a/p_counter_collection.csv $OUT/b/p_counter_collection.csv > /dev/null
  rm -rf $OUT/a $OUT/b
  python bench.py --workload $w > $OUT/bench_$w.json 2>> $OUT/bench.err
done
ls -la $OUT
