export TMPDIR=/tmp
OUT=gpurun_out/mvp_pmc; mkdir -p $OUT
M="python bench.py --workload mvp --no-cpu-baseline --steps 2 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -f csv -d $OUT/a -o p -- $M > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU -f csv -d $OUT/b -o p -- $M > $OUT/b.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/c -o p -- $M > $OUT/c.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/d -o p -- $M > $OUT/d.log 2>&1
python tools/pmc_summary.py $OUT/pmc_sq.csv $OUT/a/p_counter_collection.csv $OUT/b/p_counter_collection.csv
python tools/pmc_summary.py $OUT/pmc_traffic.csv $OUT/c/p_counter_collection.csv $OUT/d/p_counter_collection.csv
rm -rf $OUT/a $OUT/b $OUT/c $OUT/d
