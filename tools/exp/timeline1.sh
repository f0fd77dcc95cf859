set -o pipefail
cd $GRAFT_REPO_ROOT
for v in t_nb; do
for n in 8192 32768 131072; do
NB_LIB=wgpu_n_body_amd/_variants/$v.so timeout -k 10 120 python tools/walk_timeline.py $n 0.75 2>&1 | grep -v "last five" | tail -4 || exit 1
done
done
for args in "--bodies 32768 --theta 0.75" "--bodies 131072 --theta 0.75" "--theta 0.75" "--theta 0.5" "--bodies 4000000 --theta 0.75 --seed 0"; do
for lib in libnbody_hip.so _variants/nb.so; do
echo "$lib $args: $(NB_LIB=wgpu_n_body_amd/$lib timeout -k 10 120 python tools/bench_tree.py $args 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("walk %.4f build %.4f step %.4f" % (d["walk_kernel_ms"], d["build_ms"], d["ms_per_step"]))')"
done
done
