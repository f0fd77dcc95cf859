#!/bin/bash
# rocprofv3 kernel trace of the one-process LET runner (nb_runner_create_multi_let, all ranks on this GPU):
# average duration per kernel.  usage: trace_let.sh N WORLD [export_mode]
set -o pipefail
N=${1:-1048576}; W=${2:-8}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_let
rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<PY
import sys
sys.path.insert(0, "$ROOT")
import wgpu_n_body_amd as nb
sp = nb.SimParams(particle_num=$N)
r = nb.OfflineHeadless(nb.TreeSim, sp, nb.AddParams.TreeSimParams(0.5), lambda p: nb.inits.uniform_init(p, seed=3),
                       device_ids=[0] * $W, let_migrate_every=0)
r.step_n(12)
r.destroy()
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $OUT/run.py > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/t/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    name = r["Name"].replace("nb::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    print("%-46s calls %5s  avg %9.1f us  total %8.1f ms" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
