set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests/test_tree_gpu.py tests/test_full_size_gpu.py tests/test_let_gpu.py tests/test_headless_cli.py -x -q -m gpu > gpurun_out/r03/sort_tests3.log 2>&1 || { tail -40 gpurun_out/r03/sort_tests3.log; exit 1; }
tail -3 gpurun_out/r03/sort_tests3.log
for cfg in "--bodies 1048576" "--bodies 4000000 --theta 0.75 --seed 0" "--bodies 131072 --theta 0.75" "--bodies 65536 --theta 0.75" "--bodies 100000 --theta 0.75 --init disc --g 0.00001 --dt 0.0016" "--bodies 1000000 --theta 0.75 --init disc --g 0.00001 --dt 0.0016"; do
    echo "# $cfg"
    python tools/bench_tree.py $cfg --warmup 30 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms')})"
done 2>&1 | tee gpurun_out/r03/sort_variants3.txt
