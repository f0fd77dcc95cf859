"""Barnes-Hut step benchmark (BASELINE.json configs[2]: 1,048,576 bodies, theta 0.5, 1 GPU).
Prints one JSON line: ms/step, walk-kernel ms, bodies/s, counted node visits/s.  Secondary to
bench.py (the headline all-pairs metric); used for profiles/ and DESIGN.md."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_n_body_amd as nb  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bodies", type=int, default=1 << 20)
ap.add_argument("--theta", type=float, default=0.5)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--init", default="uniform")
ap.add_argument("--seed", type=int, default=3)
ap.add_argument("--scale", type=float, default=1.0, help="positions (and velocities) multiplied by this: a cube wider than [-1, 1]")
ap.add_argument("--g", type=float, default=None)
ap.add_argument("--dt", type=float, default=None)
ap.add_argument("--bpw", type=int, default=None, help="bodies per wave of the walk (default: automatic)")
ap.add_argument("--mode", type=int, default=None, help="walk kernel: 1 cells across the lanes (default), 0 bodies across the lanes")
ap.add_argument("--group", type=int, default=None, help="bodies per wave of mode 1 (4/8/16)")
ap.add_argument("--rounds", type=int, default=None, help="256-body rounds per workgroup of the cells kernels")
ap.add_argument("--sort", type=int, default=None, help="sort: 1 one-sweep (default), 0 three kernels per digit")
ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="any tuning key of nb_sim_set_tuning")
ap.add_argument("--cpu-baseline", action="store_true",
                help="also time the CPU oracle: the reference's serial BFS build + DFS reorder "
                     "(src/sims/tree.rs:417-602, single thread as in the reference) and the "
                     "restated walk (OpenMP), on the same particles")
args = ap.parse_args()

sp = nb.SimParams(particle_num=args.bodies)
if args.g is not None or args.dt is not None:
    sp = nb.SimParams(particle_num=args.bodies, g=args.g if args.g is not None else sp.g,
                      dt=args.dt if args.dt is not None else sp.dt)
init = getattr(nb.inits, args.init + "_init")(sp, seed=args.seed)
if args.scale != 1.0:
    f = nb.as_floats(init)
    f[:, 0:6] *= args.scale
sim = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(args.theta), init)
if args.mode is not None:
    sim.set_tuning("tree_walk_mode", args.mode)
if args.group is not None:
    sim.set_tuning("tree_walk_group", args.group)
if args.sort is not None:
    sim.set_tuning("tree_sort_mode", args.sort)
if args.rounds is not None:
    sim.set_tuning("tree_cell_rounds", args.rounds)
if args.bpw is not None:
    sim.set_tuning("tree_walk_bpw", args.bpw)
for kv in args.tune:
    key, value = kv.split("=")
    sim.set_tuning(key, int(value))
sim.set_tuning("tree_count_visits", 1)
sim.encode(); sim.wait()
c0 = sim.debug_buffer("counters", np.uint64).copy()
sim.set_tuning("tree_count_visits", 0)
for _ in range(args.warmup):
    sim.encode()
sim.wait()
t0 = time.perf_counter()
tot, walk = sim.encode_n_timed(args.steps)
wall = time.perf_counter() - t0
tree, rw = sim.read_tree()
out = nb.as_floats(sim.dest_particle_slice())
assert np.isfinite(out).all()
cpu = None
if args.cpu_baseline:
    from oracle import oracle as O   # bench-only use of the checker, as a reported baseline
    s0 = nb.as_floats(init)
    t0 = time.perf_counter()
    tree_ref, _rw = O.tree_build(s0)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    order_ref = O.tree_dfs_order(tree_ref, args.bodies)
    sorted_ref = s0[order_ref]
    t_sort = time.perf_counter() - t0
    threads = min(O.max_threads(), 16)
    O.set_threads(threads)
    t0 = time.perf_counter()
    O.tree_step_f32(s0, sp.g, sp.e, sp.dt, args.theta, flags=O.INTENDED)
    t_step = time.perf_counter() - t0
    cpu = {"kind": "port", "build_tree_s_1thread": t_build, "sort_particles_s_1thread": t_sort,
           "full_step_s": t_step, "walk_threads": threads,
           "note": "oracle restatement of tree.rs:417-602 (serial, as the reference) and of "
                   "tree.wgsl (intended semantics, OpenMP); full_step includes build + reorder"}

print(json.dumps({
    "metric": "Barnes-Hut step", "bodies": args.bodies, "theta": args.theta, "init": args.init,
    "ms_per_step": wall / args.steps * 1e3, "ms_per_step_events": tot / args.steps,
    "walk_kernel_ms": walk, "build_ms": tot / args.steps - walk,
    "bodies_per_s": args.bodies * args.steps / wall, "nodes": int(len(tree)),
    "visits_per_body_step1": float(c0[0]) / args.bodies, "accepted_per_body_step1": float(c0[1]) / args.bodies,
    "lane_visits_per_s": float(c0[0]) / (walk * 1e-3),
    "cells_per_wave_step1": float(c0[2]) / ((args.bodies + 63) // 64), "stack_high_water": int(c0[3]), "leaf_fraction_of_wave_cells": float(c0[4]) / float(c0[2]) if c0[2] else None,
    "lane_utilisation": (float(c0[0]) / float(c0[7]) if c0[7] else float(c0[0]) / (64.0 * float(c0[2]))) if c0[2] else None,
    "batches_step1": int(c0[6]), "evaluated_batches_step1": int(c0[9]), "idle_pair_slots_step1": int(c0[8]), "mode": args.mode, "group": args.group, "steps": args.steps, "warmup": args.warmup, "cpu_baseline": cpu}))
