set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
V=$PWD/wgpu_n_body_amd/_variants
for v in v1_mac2 v3 v3_noskip v3_addr64 v3_both v1_mac2 v3; do
  for cfg in "" "--bodies 4000000 --theta 0.75 --seed 0"; do
    echo "# $v $cfg"
    NB_LIB=$V/$v.so python tools/bench_tree.py $cfg --warmup 30 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms')})"
  done
done 2>&1 | tee gpurun_out/r03/walk_variants2.txt
for v in v1_mac2 v3; do
  echo "=== SQ counters $v"
  NB_LIB=$V/$v.so bash tools/profile_tree_pmc.sh exp_$v || exit 1
done 2>&1 | tee gpurun_out/r03/walk_pmc2.txt
