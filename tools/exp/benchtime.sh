set -o pipefail
cd $GRAFT_REPO_ROOT
S=$(date +%s.%N)
python bench.py > gpurun_out/bench_t.json 2> gpurun_out/bench_t.err
E=$(date +%s.%N)
echo "bench.py wall: $(python -c "print(round($E-$S,1))") s"
tail -c 300 gpurun_out/bench_t.json
