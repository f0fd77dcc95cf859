set -o pipefail
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for args in "--theta 0.5 --scale 300" "--bodies 131072 --theta 0.75 --scale 300" "--bodies 4000000 --theta 0.75 --seed 0 --scale 300" "--bodies 16777216 --theta 0.75 --steps 10 --scale 300"; do
for lib in _variants/base.so libnbody_hip.so; do
echo "$lib $args: $(NB_LIB=wgpu_n_body_amd/$lib timeout -k 10 120 python tools/bench_tree.py $args 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("walk %.4f build %.4f step %.4f" % (d["walk_kernel_ms"], d["build_ms"], d["ms_per_step"]))')"
done
done
done
