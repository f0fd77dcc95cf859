set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 400 python tools/sort_stress.py 240 7 2>&1 | tail -15
timeout -k 10 200 python tools/tree_stress.py 40 2>&1 | tail -5
