export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_splat.py tests/test_gpu_edges.py -x -q -m gpu 2>&1 | tail -2
bash tools/quick_kt.sh r04v_m1 2>&1 | grep -E "scatter|count_lds|sort_k"
GOL_BIN_WGS2=4096 bash tools/quick_kt.sh r04v_m1b 2>&1 | grep -E "scatter"
GOL_BIN_WGS2=1024 bash tools/quick_kt.sh r04v_m1c 2>&1 | grep -E "scatter"
