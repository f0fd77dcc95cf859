set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for n in 8192 16384 32768 65536 131072 262144 524288 1048576; do for th in 0.5 0.6 0.75 1.0; do
  line="n $n theta $th:"
  for g in 4 8 16; do
  w=$(python tools/bench_tree.py --bodies $n --theta $th --group $g --warmup 40 --steps 40 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('%.4f' % d['walk_kernel_ms'])")
  line="$line G$g $w"
  done
  echo "$line"
done; done 2>&1 | tee gpurun_out/r03/walk_groups9.txt
