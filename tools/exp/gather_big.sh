cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tools/exp/gather_big.py 2>&1 | tail -16
