export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r04k_bench.json 2>gpurun_out/r04k.err; tail -3 gpurun_out/r04k.err
python -c "
import json; d=json.load(open('gpurun_out/r04k_bench.json')); print(d['value'], d['ms_per_step']); r=d['roofline']; print({k:r.get(k) for k in ('kernel','frac','algorithmic_inst','issued_over_algorithmic','algorithmic_frac_of_peak','taken_pixel_gaussian_pairs_per_view')}); print(d['config'].get('pixel_gaussian_pairs_tested_per_view'))"
python bench.py --no-secondary --no-cpu-baseline --no-graph | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('eager', d['value'])"
python bench.py --no-secondary --no-cpu-baseline --views 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('views1', d['value'], d['ms_per_step'])"
