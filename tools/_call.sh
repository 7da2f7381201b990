export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_splat.py tests/test_gpu_edges.py tests/test_gpu_exact_math.py -x -q -m gpu 2>&1 | tail -3
bash tools/quick_kt.sh r04r_m1 2>&1 | grep -E "scatter|count_lds|sort_k"
python bench.py --no-secondary --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph', d['value'], d['kernels_ms_per_call'].get('gol_bin_sort'))"
python bench.py --workload e2e --no-cpu-baseline --steps 20 --warmup 4 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('e2e', d['value'], d['kernels_ms_per_call'].get('gol_bin_sort'))"
