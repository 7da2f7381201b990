export TMPDIR=/tmp
OUT=gpurun_out/shade_pmc; mkdir -p $OUT
for f in "" "--coherent-uv"; do
  tag=$( [ -z "$f" ] && echo rand || echo coh )
  M="python bench.py --micro 1 --no-graph --no-cpu-baseline --no-secondary --steps 3 --warmup 1 $f"
  rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/f_$tag -o p -- $M > $OUT/f_$tag.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/w_$tag -o p -- $M > $OUT/w_$tag.log 2>&1
  python tools/pmc_summary.py $OUT/traffic_$tag.csv $OUT/f_$tag/p_counter_collection.csv $OUT/w_$tag/p_counter_collection.csv > /dev/null
  echo "== $tag"; grep -E "shade_fwd|shade_bwd|scatter|count_lds|sort_k|raster" $OUT/traffic_$tag.csv | cut -d, -f1,8,9
  rm -rf $OUT/f_$tag $OUT/w_$tag
done
