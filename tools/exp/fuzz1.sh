set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python tools/tree_fuzz.py 290 101 2>&1 | grep -v "^error" | tail -8
