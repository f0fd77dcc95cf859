set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests/test_tree_gpu.py tests/test_full_size_gpu.py -x -q -m gpu > gpurun_out/r03/sort_tests2.log 2>&1 || { tail -40 gpurun_out/r03/sort_tests2.log; exit 1; }
tail -3 gpurun_out/r03/sort_tests2.log
for cfg in "--bodies 1048576" "--bodies 4000000 --theta 0.75 --seed 0" "--bodies 131072 --theta 0.75" "--bodies 32768 --theta 0.75" "--bodies 262144 --theta 0.75" "--bodies 16777216 --theta 0.75 --steps 10" "--bodies 100000 --theta 0.75 --init disc --g 0.00001 --dt 0.0016"; do
  for t in "tree_sort_spare_hi=-1" "tree_sort_spare_hi=-3" "tree_sort_spare_hi=1" "tree_sort_spare_hi=6"; do
    echo "# $cfg $t"
    python tools/bench_tree.py $cfg --tune $t --warmup 30 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms')})"
  done
done 2>&1 | tee gpurun_out/r03/sort_variants2.txt
bash tools/trace_tree.sh exp_s2 --bodies 4000000 --theta 0.75 --seed 0 | tail -22
bash tools/trace_tree.sh exp_s3 --bodies 1048576 | tail -22
