"""Criterion-style loop (time per synchronous runner.step()) for the tree group at the reference's
bench sizes, under tuning variants: usage small_n.py [KEY=VALUE,KEY=VALUE ...] (one variant per argument;
"-" = defaults), after optional --sizes A,B,.. and --thetas X,Y."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_n_body_amd as nb  # noqa: E402

argv = sys.argv[1:]
sizes = [8192, 16384, 32768, 65536, 131072]
thetas = [0.75]
while argv and argv[0].startswith("--"):
    if argv[0] == "--sizes":
        sizes = [int(v) for v in argv[1].split(",")]
    elif argv[0] == "--thetas":
        thetas = [float(v) for v in argv[1].split(",")]
    argv = argv[2:]
variants = argv or ["-"]
for theta in thetas:
    for size in sizes:
        runners = []
        for var in variants:
            sp = nb.SimParams(particle_num=size)
            runner = nb.OfflineHeadless(nb.TreeSim, sp, nb.AddParams.TreeSimParams(theta),
                                        lambda p: nb.inits.uniform_init(p, seed=size))
            if var != "-":
                for kv in var.split(","):
                    k, v = kv.split("=")
                    runner.sim.set_tuning(k, int(v))
            runners.append(runner)
        best = [1e9] * len(runners)
        for rep in range(6):            # the variants take turns: box and clock drift hit them alike
            for i, runner in enumerate(runners):
                for _ in range(50):
                    runner.step()
                t0 = time.perf_counter()
                for _ in range(200):
                    runner.step()
                best[i] = min(best[i], (time.perf_counter() - t0) / 200)
        for runner in runners:
            runner.destroy()
        print("n %7d theta %.2f: " % (size, theta)
              + " | ".join("%s %.1f" % (v, b * 1e6) for v, b in zip(variants, best)), flush=True)
