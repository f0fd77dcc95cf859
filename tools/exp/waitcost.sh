cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/exp/waitcost.py 2>&1 | tail -4
