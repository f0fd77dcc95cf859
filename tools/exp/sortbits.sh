set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for cfg in "--bodies 1048576" "--bodies 4000000 --theta 0.75 --seed 0" "--bodies 131072 --theta 0.75" "--bodies 32768 --theta 0.75" "--bodies 16777216 --theta 0.75 --steps 10"; do
  for t in "tree_sort_bits=0" "tree_sort_bits=16" "tree_sort_bits=24" "tree_sort_bits=20"; do
    echo "# $cfg $t"
    python tools/bench_tree.py $cfg --tune $t --warmup 30 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms')})"
  done
done 2>&1 | tee gpurun_out/r03/sort_bits.txt
bash tools/trace_tree.sh exp_b16 --bodies 1048576 --tune tree_sort_bits=16 | tail -22
