set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 &&
timeout -k 10 400 python tools/small_n.py --sizes 8192,16384,32768,65536,131072 --thetas 0.75 - 2>&1 | tail -20 &&
for args in "--theta 0.75" "--theta 0.5" "--bodies 4000000 --theta 0.75 --seed 0"; do
echo "$args: $(timeout -k 10 120 python tools/bench_tree.py $args 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("walk %.4f build %.4f step %.4f" % (d["walk_kernel_ms"], d["build_ms"], d["ms_per_step"]))')"
done
