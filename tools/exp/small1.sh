set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_tree_gpu.py tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -12 &&
for rep in 1 2; do
for args in "--theta 0.5" "--bodies 4000000 --theta 0.75 --seed 0" "--bodies 262144 --theta 0.75" "--bodies 131072 --theta 0.75" "--bodies 16777216 --theta 0.75 --steps 10"; do
for tune in 0 2; do
echo "gathers=$tune $args: $(timeout -k 10 120 python tools/bench_tree.py $args --tune tree_walk_gathers=$tune 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("walk %.4f build %.4f step %.4f" % (d["walk_kernel_ms"], d["build_ms"], d["ms_per_step"]))')"
done
done
done
