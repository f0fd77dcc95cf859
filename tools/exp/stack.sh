set -o pipefail
cd $GRAFT_REPO_ROOT
V=$PWD/wgpu_n_body_amd/_variants
for v in st896 st1024 st1152 st1280 st896 st1024 st1152 st1280; do
  for cfg in "--bodies 1048576" "--bodies 4000000 --theta 0.75 --seed 0" "--bodies 4194304" "--bodies 16384 --theta 0.75"; do
    echo "# $v $cfg"
    NB_LIB=$V/$v.so python tools/bench_tree.py $cfg --warmup 30 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms')})"
  done
done
