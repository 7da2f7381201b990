#!/bin/bash
# Round-6 profile pass, run ON THE GPU BOX from the repo root:  bash tools/r06_profile.sh
# gpu_profile.sh (kernel trace + FETCH / WRITE + SQ counters of the micro1 command) -> gpurun_out/r06z,
# the same FETCH / WRITE passes for --coherent-uv --smooth-normals -> gpurun_out/r06z_cs, the secondary workloads'
# SQ counters -> gpurun_out/r06z_sec.  tools/make_profile_record.py turns them into the stamped records under profiles/.
export TMPDIR=/tmp
bash tools/gpu_profile.sh r06z > gpurun_out/r06z_profile.log 2>&1
O=gpurun_out/r06z_cs; mkdir -p $O
python -c "from goliath_amd import build; print(build.source_digest())" > $O/csrc_sha16.txt
M="python bench.py --micro 1 --no-graph --no-cpu-baseline --no-secondary --coherent-uv --smooth-normals --steps 3 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c -f csv -d $O/pmc_$c -o p -- $M > $O/pmc_$c.log 2>&1; done
python tools/pmc_summary.py $O/pmc_traffic.csv $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv > /dev/null
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
bash tools/secondary_pmc.sh r06z_sec > gpurun_out/r06z_sec.log 2>&1
ls gpurun_out/r06z gpurun_out/r06z_cs gpurun_out/r06z_sec
