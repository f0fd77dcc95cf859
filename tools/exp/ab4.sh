set -o pipefail
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for n in 32768 49152 65536 98304 131072; do
line="n $n:"
for lib in "$@"; do
line="$line | $lib $(NB_LIB=wgpu_n_body_amd/$lib timeout -k 10 120 python tools/small_n.py --sizes $n --thetas 0.75 - 2>&1 | tail -1 | awk '{print $NF}')"
done
echo "$line"
done
done
