export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_meshraster.py tests/test_gpu_urhand_model.py tests/test_gpu_losses.py tests/test_gpu_splat.py -x -q -m gpu 2>&1 | tail -5
bash tools/quick_kt.sh r04m_m1 2>&1 | grep -E "l1_sum|raster_fwd|scan_kernel|count_lds"
python bench.py --no-secondary --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph', d['value'])"
