set -o pipefail
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
timeout -k 10 300 python -m pytest tests/test_naive_gpu.py -x -q -m gpu -k "sharded_ranks_reproduce" 2>&1 | tail -2
done
