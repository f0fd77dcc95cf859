#!/bin/bash
# rocprofv3 kernel trace of a few Barnes-Hut steps: per-kernel durations and the gaps between
# consecutive kernels of one step (run through gpurun).  usage: trace_tree.sh TAG [bench_tree args]
set -o pipefail
TAG=${1:-r03}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_tree_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/tools/bench_tree.py --steps 10 --warmup 3 "$@" > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/t/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last complete step: from the last morton kernel but one to the last
starts = [i for i, r in enumerate(rows) if "morton" in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
tot_busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("nb::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%8.1f us  +%5.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, name))
    prev_end = e
    tot_busy += e - s
print("step: %.1f us wall, %.1f us in kernels, %d launches" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, tot_busy / 1e3, b - a))
PY
