#!/bin/bash
# Copies what tools/collect_profiles.sh <tag> left under gpurun_out/ into the committed profiles/<tag>_* files.
# Fails -- before copying anything -- if an artefact is missing or empty.
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r03}
O=gpurun_out/${TAG}final
# (the newest: gpurun merges every call's process-id-named files into the same directory)
STATS="$(ls -t $(find gpurun_out/prof_tree_$TAG/trace -name '*kernel_stats.csv') | head -1)"
for f in $O/bench_n1.json "$STATS" $O/tree_hbm_traffic.txt $O/tree_walk_sq_counters.txt $O/criterion.txt \
         $O/criterion_sizes.json $O/tree_bench.txt $O/tree_bench_cpu_baseline.json $O/headless_cli.txt \
         $O/trace_8192.txt $O/trace_16384.txt $O/trace_131072.txt $O/trace_1048576.txt; do
  [ -s "$f" ] || { echo "copy_profiles: '$f' is missing or empty -- nothing copied"; exit 1; }
done
grep -q "SQ_INSTS_VALU" $O/tree_walk_sq_counters.txt || { echo "copy_profiles: no SQ counters of the walk -- nothing copied"; exit 1; }
python tools/summarize_profile.py $TAG > /dev/null
tail -1 $O/bench_n1.json > profiles/${TAG}_bench_n1.json
cp "$STATS" profiles/${TAG}_tree_kernel_stats.csv
cp $O/tree_hbm_traffic.txt profiles/${TAG}_tree_hbm_traffic.txt
cp $O/tree_walk_sq_counters.txt profiles/${TAG}_tree_walk_sq_counters.txt
cp $O/criterion.txt profiles/${TAG}_criterion_sizes.txt
cp $O/criterion_sizes.json profiles/${TAG}_criterion_sizes.json
cp $O/tree_bench.txt profiles/${TAG}_tree_bench.txt
grep '^{' $O/tree_bench_cpu_baseline.json | tail -1 > profiles/${TAG}_tree_bench_cpu_baseline.json
: > profiles/${TAG}_step_timelines.txt
for n in 8192 16384 131072 1048576; do
  echo "=== tools/trace_tree.sh: one Barnes-Hut step at $n bodies (rocprofv3 --kernel-trace; start, gap to the previous kernel, duration) ===" >> profiles/${TAG}_step_timelines.txt
  cat $O/trace_$n.txt >> profiles/${TAG}_step_timelines.txt
done
cp $O/headless_cli.txt profiles/${TAG}_headless_cli.txt
if [ -s $O/walk_timeline.txt ]; then
  { echo "# tools/walk_timeline.py on a -DNB_DIAG_TIMELINE build (8 bodies per wave): every wave of the walk leaves the 100 MHz clock"
    echo "# at its first instruction (launch), at its first batch (start) and after its last (end), and its batch count."
    grep -v "amdgpu.ids" $O/walk_timeline.txt; } > profiles/${TAG}_walk_timeline.txt
fi
if [ -s $O/tree_let_per_rank.json ]; then grep '^{' $O/tree_let_per_rank.json | tail -1 > profiles/${TAG}_tree_let_per_rank.json; fi
if [ -s $O/let_export.txt ]; then
  { echo "# tools/let_export_latency.py: NB_PHASE_LET_BUILD (octree of the rank's bodies + LET export for 7 peers) of ONE rank"
    echo "# with the GPU to itself, 8 Morton domains of a uniform cube, theta 0.5.  Export mode 0: a launch per tree level"
    echo "# (23 dependent launches), mode 1: one launch, a workgroup per peer and root grandchild (default)."
    cat $O/let_export.txt; } > profiles/${TAG}_let_export.txt
fi
if [ -s $O/host_overhead.txt ]; then
  grep -v "amdgpu.ids\|c10d\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/host_overhead.txt > profiles/${TAG}_host_overhead.txt
fi
for f in profiles/${TAG}_*; do [ -s "$f" ] || { echo "copy_profiles: $f came out empty"; exit 1; }; done
git status --short profiles | head -30
