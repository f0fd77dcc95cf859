set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 &&
timeout -k 10 300 python tools/soak_multi.py 2>&1 | tail -3 &&
timeout -k 10 200 python tools/sort_stress.py 60 31 2>&1 | tail -2 &&
timeout -k 10 200 python tools/tree_fuzz.py 90 77 2>&1 | tail -3
