set -o pipefail
cd $GRAFT_REPO_ROOT
V=$PWD/wgpu_n_body_amd/_variants
python -m pytest tests/test_tree_gpu.py tests/test_let_gpu.py tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in head slot head slot; do
  for cfg in "--bodies 1048576" "--bodies 4000000 --theta 0.75 --seed 0" "--bodies 131072 --theta 0.75" "--bodies 8192 --theta 0.75" "--bodies 16777216 --theta 0.75 --steps 10"; do
    echo "# $v $cfg"
    NB_LIB=$V/$v.so python tools/bench_tree.py $cfg --warmup 30 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms')})"
  done
done
