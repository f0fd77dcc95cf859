#!/bin/bash
# HBM traffic of every kernel of the Barnes-Hut step (run through gpurun): FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes (TCC slot budget, MI355X_MICROARCH.md), kernel-trace only.
# Prints per-kernel averages per launch: KiB read (x2 = the gfx950 correction for wide reads),
# KiB written, duration, and the achieved HBM GB/s.
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_tree_hbm_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/tools/bench_tree.py --steps 5 --warmup 2"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1 || { tail -20 $OUT/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1 || { tail -20 $OUT/write.log; exit 1; }
python3 - <<PY
import csv, glob, collections, re
def short(k):
    k = k.replace("(anonymous namespace)::", "")
    m = re.search(r"([A-Za-z_]\w*)\s*(<[^()]*>)?\s*\(", k)
    return (m.group(1) + (m.group(2) or "")) if m else k[:28]
def load(p, name):
    f = glob.glob("$OUT/%s/**/*_counter_collection.csv" % p, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc
def durations(p):
    f = glob.glob("$OUT/%s/**/*_kernel_trace.csv" % p, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return acc
fe, wr, du = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE"), durations("fetch")
print("# Barnes-Hut step, 1,048,576 bodies, theta 0.5: HBM traffic per launch from PMC counters")
print("# (FETCH_SIZE x 2 per MI355X_MICROARCH.md's gfx950 correction; WRITE_SIZE as reported; both KiB)")
print("%-28s %8s %12s %12s %10s %10s" % ("kernel", "launches", "read MB", "written MB", "us", "GB/s"))
for k in sorted(fe, key=lambda k: -sum(du.get(k, [0]))):
    n = len(fe[k])
    rd = 2.0 * sum(fe[k]) / n * 1024 / 1e6
    wt = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0]))) * 1024 / 1e6
    us = sum(du[k]) / len(du[k]) / 1e3 if k in du else float("nan")
    print("%-28s %8d %12.2f %12.2f %10.1f %10.0f" % (k[:28], n, rd, wt, us, (rd + wt) * 1e6 / (us * 1e-6) / 1e9 if us else 0))
PY
