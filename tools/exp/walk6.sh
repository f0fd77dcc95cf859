set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests/test_tree_gpu.py -x -q -m gpu -s -k "visualize or dense_core" 2>&1 | tail -12
for g in 4 8 16; do
for cfg in "" "--bodies 4000000 --theta 0.75 --seed 0" "--bodies 32768 --theta 0.75"; do
    echo "# group $g $cfg"
    python tools/bench_tree.py $cfg --group $g --warmup 30 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('ms_per_step_events','walk_kernel_ms','build_ms','lane_utilisation')})"
done; done 2>&1 | tee gpurun_out/r03/walk_groups6.txt
python tools/criterion_sizes.py 2>&1 | tail -15
