set -o pipefail
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout -k 10 600 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -3 || exit 1
done
