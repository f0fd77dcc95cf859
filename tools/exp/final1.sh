set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests -x -q -m gpu -rs > gpurun_out/r03/gpu_tests_final.log 2>&1 || { tail -60 gpurun_out/r03/gpu_tests_final.log; exit 1; }
tail -8 gpurun_out/r03/gpu_tests_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
bash tools/collect_profiles.sh r03
