#!/usr/bin/env python3
"""bench.py -- the headline benchmark: pair interactions/s of the all-pairs step at 65,536
bodies (BASELINE.json configs[1]) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 200 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one NaiveSim step (nb_sim_encode: kick-drift, all-pairs force accumulation,
second kick) over all 65,536 bodies; with N > 1 the bodies are partitioned by index range,
one rank per GPU, and each step ends with the in-place RCCL all-gather of the new
position/mass slices (strong scaling: the problem stays 65,536 bodies).  Inputs are
synthetic (the seeded uniform_init the reference's own criterion bench uses,
benches/benchmark.rs:24) and are resident in HBM before the timed region starts.

Rank 0 prints ONE JSON line.  `value` = N*(N-1)*K / wall seconds over all GPUs.
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


def tree_leg(nb, np, n, theta, seed, steps, device, warmup):
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
    return {"bodies": n, "theta": theta, "init": f"uniform_init seed {seed}", "steps": steps, "warmup": warmup,
            "ms_per_step": wall / steps * 1e3, "ms_per_step_events": ms_total / steps,
            "walk_ms": ms_walk, "build_ms": ms_total / steps - ms_walk,
            "bodies_per_s": n * steps / wall,
            "visits_per_body": float(c[0]) / n, "accepted_per_body": accepted,
            "lane_utilisation": (float(c[0]) / float(c[7])) if c[7] else None,
            # useful work of the walk: 20 FLOP per ACCEPTED (body, cell) interaction
            "walk_tflops": walk_tflops, "walk_tflops_frac": walk_tflops / PEAK_FP32_TFLOPS}


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tree", action="store_true", help="skip the Barnes-Hut legs (extra keys of the JSON line)")
    ap.add_argument("--variant", type=int, default=None, help="all-pairs kernel variant override")
    args = ap.parse_args()

    import numpy as np
    import torch  # first: libnbody_hip.so then binds to the HIP runtime torch already loaded
    import torch.distributed as dist

    import wgpu_n_body_amd as nb
    from wgpu_n_body_amd.sharded import ShardedNaiveSim

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available() or nb.device_count() == 0:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU fallback")
    # Rehearsal knobs (tests only): NB_DIST_BACKEND=gloo and NB_BENCH_SAME_DEVICE=1 let several
    # ranks share one GPU, which RCCL refuses; the driver's runs use neither.
    backend = os.environ.get("NB_DIST_BACKEND", "nccl")
    if os.environ.get("NB_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    # The chip needs ~50-100 ms of back-to-back launches before its clock settles
    # (profiles/r01_warmup.txt: 1.51 -> 1.26 ms per step over the first ~50 steps).  If the caller
    # asks for fewer warm-up steps than that, extra untimed steps are run first so that the K
    # timed steps measure the sustained rate; they are reported as `prewarm_steps`.
    prewarm = max(0, 120 - args.warmup)

    n = args.bodies
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
    if world == 1:
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

    wall_t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t.item())

    # sanity: the state is finite after the run (a diverged run is not a benchmark)
    state = nb.as_floats(sim.read_particles() if world > 1 else sim.dest_particle_slice())
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

    if rank == 0:
        pairs_per_step = n * (n - 1)
        value = pairs_per_step * K / wall
        per_rank = nb.shard_bodies_per_rank(n, world)
        local_pairs = min(per_rank, n) * (n - 1)              # pairs one launch evaluates
        achieved = FLOP_PER_PAIR * local_pairs / (ms_kernel * 1e-3) / 1e12
        traffic = hbm_traffic_from_profile(n) if world == 1 else None
        out = {
            "metric": "body-pair interactions/sec, 64k-body all-pairs",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": K, "warmup": W,
            "prewarm_steps": prewarm,
            "ms_per_step": wall / K * 1e3, "ms_per_step_cold": ms_cold, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n}-body naive all-pairs step (BASELINE configs[1]), "
                                   f"uniform_init seed 2, g=1e-6 e=1e-4 dt=0.016",
                       "bodies": n, "pairs_per_step": pairs_per_step,
                       "parallelism": f"body-range shard x{world}" + (
                           " + RCCL all-gather of float4 positions per step" if world > 1 else ""),
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
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nb, nb.as_floats(init), n)
        if world == 1 and not args.no_tree:
            # Barnes-Hut beside the headline, measured in the same run: BASELINE configs[2]
            # (1,048,576 bodies, theta 0.5) and the reference's own headless configuration
            # (src/bin/headless.rs:15-27: 4,000,000 bodies, theta 0.75, uniform_init)
            out["tree_1m_theta05"] = tree_leg(nb, np, 1 << 20, 0.5, 3, 40, local_rank, 60)
            out["tree_4m_theta075_headless"] = tree_leg(nb, np, 4000000, 0.75, 0, 20, local_rank, 20)
        print(json.dumps(out), flush=True)

    sim.destroy()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
