set -o pipefail
cd $GRAFT_REPO_ROOT
V=$PWD/wgpu_n_body_amd/_variants
for v in f0 f1_eager f0 f1_eager; do
  for cfg in "--bodies 1048576" "--bodies 4000000 --theta 0.75 --seed 0" "--bodies 16777216 --theta 0.75 --steps 10" "--bodies 524288 --theta 0.75"; do
    echo "# $v $cfg"
    NB_LIB=$V/$v.so python tools/bench_tree.py $cfg --warmup 30 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms')})"
  done
done 2>&1 | tee gpurun_out/r03/fill_eager.txt
