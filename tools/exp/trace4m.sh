set -o pipefail
cd $GRAFT_REPO_ROOT
bash tools/trace_tree.sh x4m --bodies 4000000 --theta 0.75 --seed 0 2>&1 | tail -24
