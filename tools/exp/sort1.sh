set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests/test_tree_gpu.py tests/test_full_size_gpu.py tests/test_let_gpu.py -x -q -m gpu > gpurun_out/r03/sort_tests1.log 2>&1 || { tail -40 gpurun_out/r03/sort_tests1.log; exit 1; }
tail -3 gpurun_out/r03/sort_tests1.log
for cfg in "" "--bodies 4000000 --theta 0.75 --seed 0" "--bodies 131072 --theta 0.75" "--bodies 262144 --theta 0.75" "--bodies 16777216 --theta 0.75 --steps 10"; do
  for t in "tree_sort_hi=0" "tree_sort_hi=1" "tree_sort_spare=4" "tree_sort_spare=3"; do
    echo "# $cfg $t"
    python tools/bench_tree.py $cfg --tune $t --warmup 30 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms')})"
  done
done 2>&1 | tee gpurun_out/r03/sort_variants1.txt
for n in 8192 16384 32768 65536 131072; do for th in 0.5 0.75; do for g in 8 16; do
  echo "# n $n theta $th group $g"
  python tools/bench_tree.py --bodies $n --theta $th --group $g --warmup 50 --steps 50 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms')})"
done; done; done 2>&1 | tee gpurun_out/r03/walk_groups7.txt
