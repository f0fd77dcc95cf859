#!/usr/bin/env python3
"""bench.py -- the headline benchmark: pair interactions/s of the all-pairs step at 65,536
bodies (BASELINE.json configs[1]) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 200 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --host native          # one process, no torchrun: nb_runner_create_multi

A "step" is one NaiveSim step (nb_sim_encode: kick-drift, all-pairs force accumulation,
second kick) over all 65,536 bodies; with N > 1 the bodies are partitioned by index range,
one rank per GPU, and each step ends with the exchange of the new position/mass slices (strong
scaling: the problem stays 65,536 bodies).  Two hosts drive the same kernels:
  --host rccl   (default) one process per GPU over torch.distributed, in-place RCCL all-gather per step
                (wgpu_n_body_amd/sharded.py) -- what the driver's torchrun line runs;
  --host native one process, a host thread per GPU inside the library (nb_runner_create_multi, the
                `extern "C"` boundary a Rust host binds): the finish kernel stores the slices into the
                peers' buffers through peer access, one HIP event per rank and step.
Inputs are synthetic (the seeded uniform_init the reference's own criterion bench uses,
benches/benchmark.rs:24) and are resident in HBM before the timed region starts.

Rank 0 prints ONE JSON line, the same schema for every N and host.  `value` = N*(N-1)*K / wall seconds
over all GPUs.  Extra keys ride along (none of them inside the timed region):
  criterion                  the reference's own benchmark, benches/benchmark.rs:12-49 (N = 1)
  tree_1m_theta05, tree_4m_theta075_headless      Barnes-Hut legs (N = 1)
  native_host, config3_262144_allpairs, config4_4m_let_theta05   (N > 1) the one-process runner on the same
                             N devices, BASELINE configs[3] and configs[4], per-rank kernel / wait times
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_PAIR = 20            # SURVEY 8(d) / BASELINE.md: the fixed N-body convention
PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: FP32 vector == FP32 matrix (MFMA f32) peak
HBM_PEAK_BPS = 8.0e12         # MI355X_MICROARCH.md: HBM3E spec
BUILD_BYTES_PER_BODY = 267    # keys + index through the sort passes, 40 B reorder, 52 B x ~1.5 nodes
G, E, DT = 0.000001, 0.0001, 0.016


def cpu_baseline(nb, init_floats, n, target_seconds=12.0):
    """The CPU oracle (fp32 restatement of naive.wgsl, OpenMP over bodies) on a bounded
    sample of the SAME workload: the step of the first `m` bodies against all n bodies
    (m sized for ~target_seconds of CPU work; whole steps are repeated if one is shorter).
    Threads: the box's CPU share for one GPU (16) unless NB_CPU_THREADS says otherwise."""
    from oracle import oracle as O
    threads = int(os.environ.get("NB_CPU_THREADS", "0")) or min(O.max_threads(), 16)
    O.set_threads(threads)
    probe = min(n, 64 * threads)
    O.naive_step_f32(init_floats, G, E, DT, 0, min(n, 16 * threads))       # page in / spin up
    t0 = time.perf_counter()
    O.naive_step_f32(init_floats, G, E, DT, 0, probe)
    dt_probe = max(time.perf_counter() - t0, 1e-6)
    m = int(min(n, max(probe, probe * target_seconds / dt_probe)))
    m = max(16, (m // 16) * 16)
    reps = 1 if m < n else int(min(20, max(1, round(target_seconds / (dt_probe * n / probe)))))
    t0 = time.perf_counter()
    for _ in range(reps):
        O.naive_step_f32(init_floats, G, E, DT, 0, m)
    secs = time.perf_counter() - t0
    pairs = reps * m * (n - 1)
    return {"value": pairs / secs, "unit": "pairs/s", "cores": threads, "kind": "port",
            "isa": O.isa(),
            "sample": f"{reps} x the all-pairs step of the first {m} of {n} bodies against all "
                      f"{n} ({pairs:.3e} pairs in {secs:.1f} s on {threads} threads); a full "
                      f"step would take {secs / reps * n / m * 1e3:.0f} ms"}


def tree_cpu_baseline(nb, np, init, n, theta, sample_every=64):
    """The CPU side of a Barnes-Hut step: the oracle's restatement of the reference's own CPU code --
    build_tree + sort_particles, src/sims/tree.rs:417-602, serial as in the reference -- timed once, and its
    restatement of tree.wgsl's walk (OpenMP) on every `sample_every`-th body, scaled to all bodies."""
    from oracle import oracle as O
    threads = int(os.environ.get("NB_CPU_THREADS", "0")) or min(O.max_threads(), 16)
    s0 = nb.as_floats(init)
    t0 = time.perf_counter()
    tree, rw = O.tree_build(s0)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    order = O.tree_dfs_order(tree, n)
    sorted_src = s0[order]
    t_sort = time.perf_counter() - t0
    O.set_threads(threads)
    sample = np.arange(0, n, sample_every)
    t0 = time.perf_counter()
    O.tree_walk_indices(sorted_src, tree, rw, G, E, DT, theta, sample, order)
    t_walk = (time.perf_counter() - t0) * n / len(sample)
    return {"value": 1.0 / (t_build + t_sort + t_walk), "unit": "steps/s", "kind": "port", "cores": threads,
            "build_tree_s_1thread": t_build, "sort_particles_s_1thread": t_sort, "walk_s": t_walk,
            "ms_per_step": (t_build + t_sort + t_walk) * 1e3,
            "sample": f"build_tree + sort_particles of all {n} bodies once on 1 thread (the reference's are serial); "
                      f"the walk of every {sample_every}th body on {threads} threads, scaled to {n}"}


def tree_leg(nb, np, n, theta, seed, steps, device, warmup, with_cpu=False):
    """One Barnes-Hut configuration: ms/step over `steps` steps after `warmup` untimed steps (the
    clock needs ~50 ms of back-to-back work to settle, as for the headline; HIP events around the
    whole steps and around the walk kernel alone), visit / accept counts of one step."""
    sp = nb.SimParams(particle_num=n, g=G, e=E, dt=DT)
    init = nb.inits.uniform_init(sp, seed=seed)
    sim = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(theta), init, nb.Placement(device_id=device))
    sim.set_tuning("tree_count_visits", 1)
    sim.encode()
    sim.wait()
    c = sim.debug_buffer("counters", np.uint64).copy()
    sim.set_tuning("tree_count_visits", 0)
    for _ in range(warmup):
        sim.encode()
    sim.wait()
    t0 = time.perf_counter()
    ms_total, ms_walk = sim.encode_n_timed(steps)
    wall = time.perf_counter() - t0
    assert np.isfinite(nb.as_floats(sim.dest_particle_slice())).all()
    sim.destroy()
    accepted = float(c[1]) / n
    walk_tflops = FLOP_PER_PAIR * float(c[1]) / (ms_walk * 1e-3) / 1e12
    build_ms = ms_total / steps - ms_walk
    out = {"bodies": n, "theta": theta, "init": f"uniform_init seed {seed}", "steps": steps, "warmup": warmup,
           "ms_per_step": wall / steps * 1e3, "ms_per_step_events": ms_total / steps,
           "walk_ms": ms_walk, "build_ms": build_ms,
           "bodies_per_s": n * steps / wall,
           "visits_per_body": float(c[0]) / n, "accepted_per_body": accepted,
           "lane_utilisation": (float(c[0]) / float(c[7])) if c[7] else None,
           # useful work of the walk: 20 FLOP per ACCEPTED (body, cell) interaction
           "walk_tflops": walk_tflops, "walk_tflops_frac": walk_tflops / PEAK_FP32_TFLOPS,
           # the build against the HBM roofline: SURVEY 8(d)'s algorithmic bytes per body over the build's time
           "build_algorithmic_bytes": BUILD_BYTES_PER_BODY * n,
           "build_hbm_frac": BUILD_BYTES_PER_BODY * n / (build_ms * 1e-3) / HBM_PEAK_BPS}
    if with_cpu:
        out["cpu_baseline"] = tree_cpu_baseline(nb, np, init, n, theta)
    return out


def criterion_rows(nb, np, device, iters=200, warmup=100):
    """The reference's only defined benchmark (benches/benchmark.rs:12-49): groups `naive` and `tree`,
    N = 8,192 ... 131,072, uniform_init, SimParams::default, theta 0.75; what is timed is one synchronous
    `runner.step()` per call (encode -> submit -> cleanup -> wait, offline_headless.rs:38-44), here through
    nb_runner_step.  Median and mean of `iters` calls after `warmup` calls, microseconds."""
    rows = []
    for group, sim_type, add in (("naive", nb.NaiveSim, nb.AddParams.NaiveSimParams()),
                                 ("tree", nb.TreeSim, nb.AddParams.TreeSimParams(0.75))):
        for size in (8192, 16384, 32768, 65536, 131072):
            sp = nb.SimParams(particle_num=size)                      # ..SimParams::default()
            runner = nb.OfflineHeadless(sim_type, sp, add, lambda p: nb.inits.uniform_init(p, seed=size),
                                        device_id=device)
            for _ in range(warmup):
                runner.step()
            ts = np.empty(iters)
            for k in range(iters):
                t0 = time.perf_counter()
                runner.step()
                ts[k] = time.perf_counter() - t0
            runner.destroy()
            med = float(np.median(ts))
            row = {"group": group, "n": size, "us_per_step_median": med * 1e6, "us_per_step_mean": float(ts.mean()) * 1e6,
                   "elements_per_s": size / med, "iterations": iters}
            if group == "naive":
                row["pairs_per_s"] = size * (size - 1) / med
            rows.append(row)
    return rows


def native_run(nb, np, sim_type, add, sp, init, device_ids, steps, warmup, let_migrate_every=None):
    """`steps` steps through the one-process runner (nb_runner_create_multi / _multi_let) on device_ids:
    wall time of one unprofiled nb_runner_step_n(steps), then the same batch again with the per-rank timing
    marks on (kernels vs waits for the peers' events).  -> (seconds, per-rank kernel ms per step, wait ms per step)"""
    runner = nb.OfflineHeadless(sim_type, sp, add, lambda _p: init, device_ids=list(device_ids),
                                let_migrate_every=let_migrate_every)
    world = len(device_ids)
    if warmup:
        runner.step_n(warmup)
    t0 = time.perf_counter()
    runner.step_n(steps)
    wall = time.perf_counter() - t0
    runner.set_profiling(True)
    runner.step_n(steps)
    k, w = runner.rank_times(world)
    runner.set_profiling(False)
    state = nb.as_floats(runner.read_particles())
    assert np.isfinite(state).all(), "non-finite state after the timed steps"
    runner.destroy()
    return wall, [x / steps for x in k], [x / steps for x in w]


def native_extra(nb, np, what, device_ids, steps, warmup):
    """An extra key measured through the one-process runner; a failure is reported in the key, not raised
    (the headline line must survive a platform on which, say, peer access is not available)."""
    try:
        if what == "headline":
            n = 65536
            sp = nb.SimParams(particle_num=n, g=G, e=E, dt=DT)
            init = nb.inits.uniform_init(sp, seed=2)
            wall, k, w = native_run(nb, np, nb.NaiveSim, None, sp, init, device_ids, steps, warmup)
            return {"workload": "65,536-body all-pairs (BASELINE configs[1]), nb_runner_create_multi",
                    "ms_per_step": wall / steps * 1e3, "pairs_per_s": n * (n - 1) * steps / wall,
                    "rank_kernel_ms": k, "rank_wait_ms": w, "steps": steps, "warmup": warmup}
        if what == "config3":
            n = 262144
            sp = nb.SimParams(particle_num=n, g=G, e=E, dt=DT)
            init = nb.inits.uniform_init(sp, seed=4)
            wall, k, w = native_run(nb, np, nb.NaiveSim, None, sp, init, device_ids, steps, warmup)
            return {"workload": "262,144-body all-pairs (BASELINE configs[3]), nb_runner_create_multi",
                    "ms_per_step": wall / steps * 1e3, "pairs_per_s": n * (n - 1) * steps / wall,
                    "rank_kernel_ms": k, "rank_wait_ms": w, "steps": steps, "warmup": warmup}
        if what == "config4":
            n = 4194304
            sp = nb.SimParams(particle_num=n, g=G, e=E, dt=DT)
            init = nb.inits.uniform_init(sp, seed=5)
            wall, k, w = native_run(nb, np, nb.TreeSim, nb.AddParams.TreeSimParams(0.5), sp, init, device_ids, steps,
                                    warmup, let_migrate_every=1)
            return {"workload": "4,194,304-body Barnes-Hut theta 0.5, Morton domains + LET exchange (BASELINE "
                                "configs[4]), nb_runner_create_multi_let, migration every step",
                    "ms_per_step": wall / steps * 1e3, "bodies_per_s": n * steps / wall,
                    "rank_kernel_ms": k, "rank_wait_ms": w, "steps": steps, "warmup": warmup}
    except Exception as exc:  # noqa: BLE001 -- reported, see the docstring
        return {"error": f"{type(exc).__name__}: {exc}"}
    raise ValueError(what)


def run_extras_child(gpus, steps, warmup, limit_s=300):
    """`bench.py --gpus N --host native --extras-only` as a child process -> its dict, or {"error": ...}."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(gpus), "--steps", str(steps), "--warmup", str(warmup),
           "--host", "native", "--extras-only"]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=limit_s, env=env)
    except subprocess.TimeoutExpired:
        return {"error": f"the one-process runner did not finish within {limit_s} s (child process killed)"}
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": f"child exit {p.returncode}: {(p.stderr or p.stdout)[-400:]}"}
    return json.loads(lines[-1])


def hbm_traffic_from_profile(n):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/*_pmc_summary.json, collected as MI355X_MICROARCH.md prescribes); null if the
    summary does not cover this N."""
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.json")), reverse=True):
            d = json.load(open(path))
            if int(d.get("n", -1)) == n and d.get("hbm_bytes_per_launch") is not None:
                return float(d["hbm_bytes_per_launch"])
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the chip needs ~50 steps (~70 ms) of back-to-back launches before its clock
    # settles (profiles/r01_warmup.txt: 1.51 ms -> 1.26 ms per step), so warm up past that
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--bodies", type=int, default=65536)
    ap.add_argument("--host", choices=("rccl", "native"), default="rccl",
                    help="rccl: one process per GPU (torchrun when --gpus > 1); native: one process, nb_runner_create_multi")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tree", action="store_true", help="skip the Barnes-Hut legs (extra keys of the JSON line)")
    ap.add_argument("--no-criterion", action="store_true", help="skip the reference's criterion rows")
    ap.add_argument("--no-extras", action="store_true", help="N > 1: skip native_host / config3 / config4")
    ap.add_argument("--extras-only", action="store_true",
                    help="(internal) print only the one-process runner's extra keys for --gpus N, as JSON")
    ap.add_argument("--variant", type=int, default=None, help="all-pairs kernel variant override")
    args = ap.parse_args()

    import numpy as np
    import torch  # first: libnbody_hip.so then binds to the HIP runtime torch already loaded
    import torch.distributed as dist

    import wgpu_n_body_amd as nb
    from wgpu_n_body_amd.sharded import ShardedNaiveSim

    if args.extras_only:
        # the child process of a torchrun rank 0 (below): the one-process runner on the N devices, nothing else
        same = os.environ.get("NB_BENCH_SAME_DEVICE") == "1"
        ids = [0] * args.gpus if same else list(range(args.gpus))
        out = {"native_host": native_extra(nb, np, "headline", ids, args.steps, max(args.warmup, 20)),
               "config3": native_extra(nb, np, "config3", ids, 20, 5),
               "config4": native_extra(nb, np, "config4", ids, 20, 5)}
        print(json.dumps(out), flush=True)
        return

    native = args.host == "native"
    world = 1 if native else int(os.environ.get("WORLD_SIZE", "1"))
    rank = 0 if native else int(os.environ.get("RANK", "0"))
    local_rank = 0 if native else int(os.environ.get("LOCAL_RANK", "0"))
    if not native and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"(or --host native for one process)")
    if not torch.cuda.is_available() or nb.device_count() == 0:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU fallback")
    # Rehearsal knobs (tests only): NB_DIST_BACKEND=gloo and NB_BENCH_SAME_DEVICE=1 let several
    # ranks share one GPU, which RCCL refuses; the driver's runs use neither.
    backend = os.environ.get("NB_DIST_BACKEND", "nccl")
    same_device = os.environ.get("NB_BENCH_SAME_DEVICE") == "1"
    if same_device:
        local_rank = 0
    device_ids = [0] * args.gpus if same_device else list(range(args.gpus))
    if native and max(device_ids) >= nb.device_count():
        raise SystemExit(f"--host native --gpus {args.gpus}: only {nb.device_count()} device(s) visible")
    torch.cuda.set_device(local_rank)
    host_group = None
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            host_group = dist.new_group(backend="gloo")   # host-side barriers: no kernel parked on a GPU
        else:
            dist.init_process_group(backend)

    # The chip needs ~50-100 ms of back-to-back launches before its clock settles
    # (profiles/r01_warmup.txt: 1.51 -> 1.26 ms per step over the first ~50 steps).  If the caller
    # asks for fewer warm-up steps than that, extra untimed steps are run first so that the K
    # timed steps measure the sustained rate; they are reported as `prewarm_steps`.
    prewarm = max(0, 120 - args.warmup)

    n = args.bodies
    n_gpus = args.gpus
    sp = nb.SimParams(particle_num=n, g=G, e=E, dt=DT)
    init = nb.inits.uniform_init(sp, seed=2)          # identical bytes on every rank
    K, W = args.steps, args.warmup

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    ms_kernel = None
    ms_cold = None
    ranks_info = None
    sim = None
    if native:
        # one process: the timed region is one nb_runner_step_n(K) -- the library's own threads enqueue the
        # K steps on every device and return when every stream has drained (its barrier is inside).  The same
        # code for N = 1 (device_ids = [0]: nb_runner_create_multi hands a one-device list to nb_runner_create)
        runner = nb.OfflineHeadless(nb.NaiveSim, sp, None, lambda _p: init, device_ids=device_ids)
        t0 = time.perf_counter()
        runner.step_n(max(W, 1))
        ms_cold = (time.perf_counter() - t0) / max(W, 1) * 1e3
        if prewarm:
            runner.step_n(prewarm)
        t0 = time.perf_counter()
        runner.step_n(K)
        wall = time.perf_counter() - t0
        state = nb.as_floats(runner.read_particles())
        runner.set_profiling(True)                     # after the timed region: the same batch with timing marks
        runner.step_n(K)
        kms, wms = runner.rank_times(n_gpus)
        runner.destroy()
        ranks_info = {"rank_kernel_ms": [x / K for x in kms], "rank_wait_ms": [x / K for x in wms]}
        ms_kernel = max(kms) / K
    elif world == 1:
        sim = nb.NaiveSim.from_particles(sp, None, init, nb.Placement(device_id=local_rank))
        if args.variant is not None:
            sim.set_tuning("naive_variant", args.variant)
        # the rate with exactly the warm-up the caller asked for (clock still ramping), reported
        # beside the sustained one so that the extra pre-warm steps hide nothing
        sim.encode()
        sim.wait()
        t0 = time.perf_counter()
        for _ in range(max(W, 1)):
            sim.encode()
        sim.wait()
        ms_cold = (time.perf_counter() - t0) / max(W, 1) * 1e3
        for _ in range(prewarm):
            sim.encode()
        sim.wait()
        sync_all()
        t0 = time.perf_counter()
        # K steps back to back on the simulator's stream, HIP events around each launch
        _ms_total, ms_kernel = sim.encode_n_timed(K)
        sync_all()
        wall = time.perf_counter() - t0
        state = nb.as_floats(sim.dest_particle_slice())
    else:
        sim = ShardedNaiveSim(sp, init, rank, world, local_rank, variant=args.variant)
        for _ in range(prewarm + W):
            sim.encode()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(K):
            sim.encode()
        sync_all()
        wall = time.perf_counter() - t0
        sim.wait()
        state = nb.as_floats(sim.read_particles())

    wall_t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t.item())

    # sanity: the state is finite after the run (a diverged run is not a benchmark)
    assert np.isfinite(state).all(), "non-finite state after the timed steps"

    if world > 1:
        # kernel-only duration of one rank's share (both halves + integrator), measured after
        # the timed region and the sanity check, without the exchange: the state is not used
        # again, only the launch durations are
        with torch.cuda.stream(sim.stream):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                sim.sim.encode()
            ev[1].record()
        sim.stream.synchronize()
        ms_kernel = ev[0].elapsed_time(ev[1]) / 10

    # BASELINE configs[3] through the RCCL host (every rank takes part; not in the timed region above)
    config3_rccl = None
    if world > 1 and not args.no_extras:
        try:
            n3 = 262144
            sp3 = nb.SimParams(particle_num=n3, g=G, e=E, dt=DT)
            sim3 = ShardedNaiveSim(sp3, nb.inits.uniform_init(sp3, seed=4), rank, world, local_rank)
            for _ in range(5):
                sim3.encode()
            sync_all()
            t0 = time.perf_counter()
            for _ in range(20):
                sim3.encode()
            sync_all()
            w3 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
            dist.all_reduce(w3, op=dist.ReduceOp.MAX)
            sim3.wait()
            sim3.destroy()
            config3_rccl = {"ms_per_step": float(w3.item()) / 20 * 1e3, "pairs_per_s": n3 * (n3 - 1) * 20 / float(w3.item()),
                            "steps": 20, "warmup": 5}
        except Exception as exc:  # noqa: BLE001
            config3_rccl = {"error": f"{type(exc).__name__}: {exc}"}

    if sim is not None:
        sim.destroy()
        sim = None

    if rank == 0:
        pairs_per_step = n * (n - 1)
        value = pairs_per_step * K / wall
        per_rank = nb.shard_bodies_per_rank(n, n_gpus)
        local_pairs = min(per_rank, n) * (n - 1)              # pairs one launch evaluates
        achieved = FLOP_PER_PAIR * local_pairs / (ms_kernel * 1e-3) / 1e12
        traffic = hbm_traffic_from_profile(n) if n_gpus == 1 else None
        exchange = ("" if n_gpus == 1 else
                    " + peer stores of the float4 slices from the finish kernel, one event per rank and step" if native
                    else " + RCCL all-gather of float4 positions per step")
        out = {
            "metric": "body-pair interactions/sec, 64k-body all-pairs",
            "value": value, "unit": "pairs/s", "n_gpus": n_gpus, "steps": K, "warmup": W,
            "prewarm_steps": prewarm,
            "ms_per_step": wall / K * 1e3, "ms_per_step_cold": ms_cold, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n}-body naive all-pairs step (BASELINE configs[1]), "
                                   f"uniform_init seed 2, g=1e-6 e=1e-4 dt=0.016",
                       "bodies": n, "pairs_per_step": pairs_per_step,
                       "parallelism": f"body-range shard x{n_gpus}" + exchange,
                       "host": ("native: one process, nb_runner_create_multi over device_ids " + str(device_ids) if native else
                                "rccl: one process per GPU, torch.distributed" if n_gpus > 1 else "one process, one device (nb_sim_*)"),
                       "lib": nb.version(),
                       "kernel_variant": (nb.naive_variants()[args.variant]
                                          if args.variant is not None else "auto")},
            "roofline": {"bound": "mfma", "bound_detail": "fp32_valu_issue (the kernel issues no MFMA; the "
                         "schema's compute side)", "achieved": achieved, "peak": PEAK_FP32_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                         "traffic": traffic,
                         "hbm_gbps": (traffic / (ms_kernel * 1e-3) / 1e9) if traffic else None,
                         "hbm_peak_gbps": 8000.0,
                         "kernel": "nb::naive_step_kernel", "kernel_ms": ms_kernel,
                         "flop_per_pair": FLOP_PER_PAIR,
                         # the instruction-issue ceiling of this force law on this chip (DESIGN.md
                         # section 4: 12 packed + 4 transcendental issues per 2 pairs at the measured
                         # 1.85 / 3.41 ns per wave-instruction and SIMD): 3.59e12 pairs/s
                         "issue_ceiling_frac": (achieved * 1e12 / FLOP_PER_PAIR) / 3.59e12,
                         "note": "compute-bound on FP32 VALU issue; 157.3 TFLOP/s is both the "
                                 "vector and the f32-MFMA dense peak"},
        }
        if ranks_info:
            out["ranks"] = ranks_info
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nb, nb.as_floats(init), n)
        if n_gpus == 1 and not args.no_criterion:
            out["criterion"] = {"source": "benches/benchmark.rs:12-49: runner.step() per call, uniform_init, "
                                          "SimParams::default, theta 0.75", "rows": criterion_rows(nb, np, local_rank)}
        if n_gpus == 1 and not args.no_tree:
            # Barnes-Hut beside the headline, measured in the same run: BASELINE configs[2]
            # (1,048,576 bodies, theta 0.5) and the reference's own headless configuration
            # (src/bin/headless.rs:15-27: 4,000,000 bodies, theta 0.75, uniform_init)
            out["tree_1m_theta05"] = tree_leg(nb, np, 1 << 20, 0.5, 3, 40, local_rank, 60,
                                              with_cpu=not args.no_cpu_baseline)
            out["tree_4m_theta075_headless"] = tree_leg(nb, np, 4000000, 0.75, 0, 20, local_rank, 20)
        if n_gpus > 1 and not args.no_extras:
            if native:
                ex = {"config3": native_extra(nb, np, "config3", device_ids, 20, 5),
                      "config4": native_extra(nb, np, "config4", device_ids, 20, 5)}
            else:
                # The other ranks' processes are parked at a host-side barrier below: their GPUs are idle.  The
                # one-process runner has never met two real devices: it runs in a CHILD process with a time limit,
                # so that a fault or a hang there costs its keys and not the line.
                ex = run_extras_child(args.gpus, K, W)
                out["native_host"] = ex.get("native_host", ex)
            out["config3_262144_allpairs"] = {"rccl_host": config3_rccl, "native_host": ex.get("config3", ex)}
            out["config4_4m_let_theta05"] = {"native_host": ex.get("config4", ex)}
        print(json.dumps(out), flush=True)

    if world > 1:
        if host_group is not None:
            dist.barrier(group=host_group)
        else:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
