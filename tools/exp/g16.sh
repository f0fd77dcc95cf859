set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests/test_tree_gpu.py tests/test_let_gpu.py -x -q -m gpu > gpurun_out/r03/fill_tests.log 2>&1 || { tail -40 gpurun_out/r03/fill_tests.log; exit 1; }
tail -2 gpurun_out/r03/fill_tests.log
for n in 262144 524288 1048576 2097152; do for th in 0.6 0.75 1.0; do for g in 8 16; do
  echo "# n $n theta $th group $g"
  python tools/bench_tree.py --bodies $n --theta $th --group $g --warmup 30 --steps 20 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms')})"
done; done; done 2>&1 | tee gpurun_out/r03/walk_groups8.txt
for n in 32768 65536 131072; do for th in 0.6 1.0; do for g in 8 16; do
  echo "# n $n theta $th group $g"
  python tools/bench_tree.py --bodies $n --theta $th --group $g --warmup 50 --steps 50 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms')})"
done; done; done 2>&1 | tee -a gpurun_out/r03/walk_groups8.txt
