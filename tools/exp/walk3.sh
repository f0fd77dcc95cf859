set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
V=$PWD/wgpu_n_body_amd/_variants
for v in v1_mac2 v4 v4_fmt v4_oldc v4_oldpop v4_oldem v4_allold v1_mac2 v4; do
  for cfg in "" ; do
    echo "# $v $cfg"
    NB_LIB=$V/$v.so python tools/bench_tree.py $cfg --warmup 30 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms')})"
  done
done 2>&1 | tee gpurun_out/r03/walk_variants3.txt
for v in v4 v4_allold; do
  echo "=== SQ counters $v"
  NB_LIB=$V/$v.so bash tools/profile_tree_pmc.sh exp_$v || exit 1
done 2>&1 | tee gpurun_out/r03/walk_pmc3.txt
