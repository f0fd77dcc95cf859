#!/bin/bash
# SQ counters of the Barnes-Hut walk kernel (run through gpurun).  --pmc passes of <= 8 SQ counters
# each, kernel-trace only (no other trace domains).  Prints per-launch averages of the TIMED kernel --
# walk_cells_kernel<8, false, ...>: COUNT = false; the one launch with COUNT = true is the
# statistics step of tools/bench_tree.py -- and, from that statistics step's batch count, the
# instructions per batch.  Fails if no launch matches.
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_tree_pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/tools/bench_tree.py --steps 5 --warmup 2"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1 || { tail -20 $OUT/p1.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1 || { tail -20 $OUT/p2.log; exit 1; }
# (the per-type VALU counters are not in every rocprofv3 build's gfx950 list: this pass may be empty)
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/p3 -- python3 $ARGS > $OUT/p3.log 2>&1 || echo "# pass 3 (per-type VALU counters) not available: $(tail -1 $OUT/p3.log)"
python3 - <<PY
import csv, glob, collections, json, re, sys
TIMED = "walk_cells_kernel<8, false"
batches = None
for line in open("$OUT/p1.log"):
    if line.startswith("{"):
        batches = json.loads(line).get("batches_step1")
found = 0
tot = {}
for p in ("p1", "p2", "p3"):
    fs = glob.glob("$OUT/%s/**/*_counter_collection.csv" % p, recursive=True)
    if not fs:
        continue
    acc = collections.defaultdict(list)
    names = set()
    for r in csv.DictReader(open(fs[0])):
        if TIMED in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            names.add(r["Kernel_Name"].split("(")[0])
    for k, v in sorted(acc.items()):
        tot[k] = sum(v) / len(v)
        found += 1
        print("%-26s %16.0f  (avg of %d launches)" % (k, tot[k], len(v)))
    if p == "p1":
        print("# kernel:", "; ".join(sorted(names)))
if not found:
    sys.exit("no launch of %s in the counter files" % TIMED)
if batches:
    print("# batches per launch (statistics step of the same run): %d" % batches)
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_VALU_TRANS_F32",
              "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_INT32"):
        if k in tot:
            print("%-26s %10.1f per batch" % (k, tot[k] / batches))
PY
