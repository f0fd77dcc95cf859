set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_tree_gpu.py -x -q -m gpu -k "random_cases or closed_form or gathers" 2>&1 | tail -3
