set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
echo "== let_stress"; timeout -k 10 300 python tools/let_stress.py 40 2>&1 | tail -4
echo "== multi_stress"; timeout -k 10 300 python tools/multi_stress.py 30 2>&1 | tail -4
echo "== soak_multi"; timeout -k 10 400 python tools/soak_multi.py 2>&1 | tail -8
echo "== sort_stress"; timeout -k 10 200 python tools/sort_stress.py 100 23 2>&1 | tail -3
