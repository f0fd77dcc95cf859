set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests/test_multi_device_gpu.py tests/test_headless_cli.py tests/test_bench_gpu.py tests/test_let_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -rs > gpurun_out/r03/bench_tests1.log 2>&1 || { tail -60 gpurun_out/r03/bench_tests1.log; exit 1; }
tail -8 gpurun_out/r03/bench_tests1.log
python bench.py > gpurun_out/r03/bench_n1_a.json 2> gpurun_out/r03/bench_n1_a.err || { tail gpurun_out/r03/bench_n1_a.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r03/bench_n1_a.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','n_gpus')}, d['roofline']['frac'], d['config']['lib'])
for r in d['criterion']['rows']: print(r['group'], r['n'], round(r['us_per_step_median'],1), round(r['us_per_step_mean'],1))
for k in ('tree_1m_theta05','tree_4m_theta075_headless'): print(k, {x:d[k][x] for x in ('ms_per_step','walk_ms','build_ms','build_hbm_frac','walk_tflops_frac')}, d[k].get('cpu_baseline'))
"
