#!/bin/bash
# Everything the committed profiles/<tag>_* files are made from, in one gpurun call:
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r03'
# then, back in the build container, tools/copy_profiles.sh r03.  Every step that leaves an artefact
# checks that the artefact is there and not empty; the first failure ends the call.
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/${TAG}final
rm -rf $O; mkdir -p $O
need() { for f in "$@"; do [ -s "$f" ] || { echo "collect_profiles: $f is missing or empty"; exit 1; }; done; }
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail $O/bench_n1.err; exit 1; }
need $O/bench_n1.json
echo "bench done"
bash tools/profile_bench.sh $TAG > $O/profile_bench.log 2>&1 || { tail $O/profile_bench.log; exit 1; }
bash tools/profile_tree.sh $TAG > $O/profile_tree.log 2>&1 || { tail $O/profile_tree.log; exit 1; }
bash tools/profile_tree_hbm.sh $TAG > $O/tree_hbm_traffic.txt 2>&1 || { tail $O/tree_hbm_traffic.txt; exit 1; }
bash tools/profile_tree_pmc.sh $TAG > $O/tree_walk_sq_counters.txt 2>&1 || { tail $O/tree_walk_sq_counters.txt; exit 1; }
need $O/tree_hbm_traffic.txt $O/tree_walk_sq_counters.txt
grep -q "SQ_INSTS_VALU" $O/tree_walk_sq_counters.txt || { echo "collect_profiles: no SQ counters of the walk"; exit 1; }
echo "profiles done"
python tools/criterion_sizes.py > $O/criterion.txt 2>&1 && cp gpurun_out/criterion_sizes.json $O/criterion_sizes.json
need $O/criterion.txt $O/criterion_sizes.json
for cfg in "" "--theta 0.75" "--bodies 4194304" "--bodies 4000000 --theta 0.75 --seed 0" "--bodies 16777216 --theta 0.75 --steps 10" "--bodies 26843545 --theta 0.75 --steps 5" "--bodies 100000 --theta 0.75 --init disc --g 0.00001 --dt 0.0016" "--mode 0"; do
  echo "# bench_tree.py $cfg" >> $O/tree_bench.txt
  python tools/bench_tree.py $cfg --warmup 30 >> $O/tree_bench.txt 2>&1 || { tail -5 $O/tree_bench.txt; exit 1; }
done
python tools/bench_tree.py --cpu-baseline --steps 5 > $O/tree_bench_cpu_baseline.json 2>&1
need $O/tree_bench.txt $O/tree_bench_cpu_baseline.json
echo "tree bench done"
for n in 8192 16384 131072 1048576; do
  bash tools/trace_tree.sh ${TAG}_$n --bodies $n --theta $([ $n = 1048576 ] && echo 0.5 || echo 0.75) > $O/trace_$n.txt 2>&1 || { tail $O/trace_$n.txt; exit 1; }
  need $O/trace_$n.txt
done
# when the waves of the walk ran (a -DNB_DIAG_TIMELINE build of the same sources: tools/build_variant.sh timeline -DNB_DIAG_TIMELINE)
if [ -s wgpu_n_body_amd/_variants/timeline.so ]; then
  for n in 8192 32768 131072 1048576; do
    NB_LIB=wgpu_n_body_amd/_variants/timeline.so python tools/walk_timeline.py $n 0.75 >> $O/walk_timeline.txt 2>&1 || { tail $O/walk_timeline.txt; exit 1; }
  done
  NB_LIB=wgpu_n_body_amd/_variants/timeline.so python tools/walk_timeline.py 1048576 0.5 >> $O/walk_timeline.txt 2>&1 || true
fi
./wgpu_n_body_amd/headless > $O/headless_cli.txt 2>&1
./wgpu_n_body_amd/headless --sim naive --n 65536 --steps 5 --devices 0,0,0,0,0,0,0,0 >> $O/headless_cli.txt 2>&1
need $O/headless_cli.txt
python tools/host_overhead.py > $O/host_overhead.txt 2>&1 || true
python tools/bench_tree_let.py --bodies 4194304 > $O/tree_let_per_rank.json 2> $O/tree_let.err || true
for n in 65536 262144 1048576 4194304; do python tools/let_export_latency.py $n 8 | grep "^n "; done > $O/let_export.txt 2>&1 || true
echo "all done"; ls -la $O
