set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests/test_tree_gpu.py -x -q -m gpu > gpurun_out/r03/tree_tests1.log 2>&1 || { tail -30 gpurun_out/r03/tree_tests1.log; exit 1; }
tail -3 gpurun_out/r03/tree_tests1.log
for v in base v1_mac2 v2_trim; do
  for cfg in "" "--bodies 4000000 --theta 0.75 --seed 0" "--bodies 131072 --theta 0.75"; do
    echo "# $v $cfg"
    NB_LIB=$PWD/wgpu_n_body_amd/_variants/$v.so python tools/bench_tree.py $cfg --warmup 30 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms','lane_utilisation','batches_step1','visits_per_body_step1')})"
  done
done 2>&1 | tee gpurun_out/r03/walk_variants1.txt
