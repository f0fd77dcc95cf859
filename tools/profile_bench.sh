#!/bin/bash
# Profiles the default bench.py run with rocprofv3 on the GPU box (run through gpurun):
#   pass 1: --kernel-trace --stats          -> per-kernel durations
#   pass 2: --pmc FETCH_SIZE                -> HBM read traffic   (own pass, MI355X_MICROARCH.md)
#   pass 3: --pmc WRITE_SIZE                -> HBM write traffic  (own pass)
# Outputs land in gpurun_out/prof_<tag>/; tools/summarize_profile.py turns them into
# profiles/<tag>_*.{txt,json}.
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --no-cpu-baseline --no-tree --steps 200 --warmup 100"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 || { tail -20 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1 || { tail -20 $OUT/pmc_write.log; exit 1; }
find $OUT -name "*.csv" | head -30
