set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for cfg in "" "--theta 0.75" "--bodies 4000000 --theta 0.75 --seed 0" "--bodies 131072 --theta 0.75" "--bodies 16384 --theta 0.75"; do
    echo "# $cfg"
    python tools/bench_tree.py $cfg --warmup 30 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms','batches_step1','evaluated_batches_step1','idle_pair_slots_step1','lane_utilisation')})"
done 2>&1 | tee gpurun_out/r03/walk_stats5.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r03/gpu_tests5.log 2>&1; tail -15 gpurun_out/r03/gpu_tests5.log
