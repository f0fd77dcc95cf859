set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_tree_gpu.py -x -q -m gpu --durations=5 2>&1 | tail -14
