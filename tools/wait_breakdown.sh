#!/bin/bash
# What do the latency-bound kernels wait for?  Run ON THE GPU BOX:  bash tools/wait_breakdown.sh TAG [bench args]
# Four rocprofv3 --pmc passes over the micro1 bench command (8 views per launch, eager) -> gpurun_out/TAG/wait_breakdown.csv:
# per kernel the wave-cycle split (issue waits, LDS waits, active cycles per instruction class), the average number of
# outstanding VMEM / LDS instructions (LEVEL counters: level / instructions = average latency in cycles), store-path back
# pressure (TA data FIFO full, EA write-request stalls) and LDS atomic / conflict counts.
TAG=${1:-wb}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -c "from goliath_amd import build; print(build.source_digest())" > $OUT/csrc_sha16.txt
M="python bench.py --micro 1 --no-graph --no-cpu-baseline --no-secondary --steps 3 --warmup 1 $@"
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT" \
  "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" \
  "TCP_PENDING_STALL_CYCLES TCC_EA0_WRREQ_STALL TCC_TAG_STALL TCP_TCP_TA_DATA_STALL_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -f csv -d $OUT/p$i -o p -- $M > $OUT/p$i.log 2>&1
done
python tools/pmc_summary.py $OUT/wait_breakdown.csv $OUT/p1/p_counter_collection.csv $OUT/p2/p_counter_collection.csv $OUT/p3/p_counter_collection.csv $OUT/p4/p_counter_collection.csv > /dev/null
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
ls -la $OUT
