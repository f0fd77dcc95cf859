set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 &&
timeout -k 10 400 python tools/small_n.py --sizes 1024,8192,16384,32768,65536,131072 --thetas 0.75 - 2>&1 | tail -20
