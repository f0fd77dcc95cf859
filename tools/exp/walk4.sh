set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
V=$PWD/wgpu_n_body_amd/_variants
for v in v1_mac2 v5 v5_oldc v5_exec v5_exec_oldc v5 v5_exec; do
  for cfg in "" "--bodies 4000000 --theta 0.75 --seed 0" "--bodies 16384 --theta 0.75"; do
    echo "# $v $cfg"
    NB_LIB=$V/$v.so python tools/bench_tree.py $cfg --warmup 30 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms')})"
  done
done 2>&1 | tee gpurun_out/r03/walk_variants4.txt
NB_LIB=$V/v5_exec.so python -m pytest tests/test_tree_gpu.py -x -q -m gpu 2>&1 | tail -5
