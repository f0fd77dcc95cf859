set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python tools/tree_fuzz.py 300 13 2>&1 | tail -30
