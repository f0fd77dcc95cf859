#!/bin/bash
# rocprofv3 kernel trace of the Barnes-Hut step benchmark (run through gpurun).
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_tree_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/bench_tree.py --steps 10 --warmup 3 > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
grep '^{' $OUT/trace.log
find $OUT -name "*kernel_stats.csv" | head -2
