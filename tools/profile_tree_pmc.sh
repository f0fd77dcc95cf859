#!/bin/bash
# SQ counters of the Barnes-Hut walk kernel (run through gpurun).  Two --pmc passes of <= 8 SQ
# counters each, kernel-trace only (no other trace domains).  Prints per-launch averages.
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_tree_pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/tools/bench_tree.py --steps 5 --warmup 2"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1 || { tail -20 $OUT/p1.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1 || { tail -20 $OUT/p2.log; exit 1; }
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    f = glob.glob("$OUT/%s/**/*_counter_collection.csv" % p, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "walk_" in r["Kernel_Name"] and "true" not in r["Kernel_Name"].split(">(")[0]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-24s %16.0f  (avg of %d launches)" % (k, sum(v) / len(v), len(v)))
PY
