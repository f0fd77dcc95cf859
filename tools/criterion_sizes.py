"""The reference's own criterion benchmark, on the HIP engine.

benches/benchmark.rs:11-50 defines two groups, `naive` and `tree`, over
N in {8192, 16384, 32768, 65536, 131072}: time per `runner.step()` with
Throughput::Elements(N), uniform_init, SimParams::default, theta = 0.75 for the tree group.
The reference commits no results; this prints the same table for this engine (steady state:
warm-up steps first, then the mean over `--steps` synchronous runner.step() calls, like
criterion's b.iter(|| runner.step()))."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_n_body_amd as nb  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--warmup", type=int, default=100)
args = ap.parse_args()

KB = 8192
rows = []
for group, sim_type, add in (("naive", nb.NaiveSim, nb.AddParams.NaiveSimParams()),
                             ("tree", nb.TreeSim, nb.AddParams.TreeSimParams(0.75))):
    for size in (KB, KB * 2, KB * 4, KB * 8, KB * 16):
        sp = nb.SimParams(particle_num=size)                      # ..SimParams::default()
        runner = nb.OfflineHeadless(sim_type, sp, add, lambda p: nb.inits.uniform_init(p, seed=size))
        for _ in range(args.warmup):
            runner.step()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            runner.step()                                         # encode -> submit -> cleanup -> wait
        dt = (time.perf_counter() - t0) / args.steps
        runner.destroy()
        row = dict(group=group, n=size, us_per_step=dt * 1e6, elements_per_s=size / dt)
        if group == "naive":
            row["pairs_per_s"] = size * (size - 1) / dt
        rows.append(row)
        print(f"{group}/{size:<7d} time: {dt*1e6:10.1f} us   thrpt: {size/dt/1e6:9.2f} Melem/s"
              + (f"   {size*(size-1)/dt/1e12:6.3f} Tpairs/s" if group == "naive" else ""), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/criterion_sizes.json", "w"), indent=1)
