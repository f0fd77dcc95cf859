"""Per-rank cost of the LET Barnes-Hut step (BASELINE configs[4]: 4,194,304 bodies, theta 0.5,
8 ranks), measured on ONE GPU: all `world` domain simulators live on this device, the exchanges
are device-to-device copies (untimed: on a node they are RCCL collectives over xGMI), and the
three phases of every rank are timed with HIP events.  All ranks share one stream, so a rank's
kernels have the GPU to themselves, and within a phase the ranks run back to back, which keeps
the clock up (a GPU that idles between short kernels runs them markedly slower).

Prints one JSON line: per-rank phase times (mean / max over ranks), LET sizes, and next to it
the replicated-tree scheme (every rank builds the full tree, walks 1/world of the bodies) and
the single-GPU step for the same bodies.

    python tools/bench_tree_let.py [--bodies 4194304] [--world 8] [--theta 0.5] [--steps 5]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (first: torch bundles its own HIP runtime)

import wgpu_n_body_amd as nb  # noqa: E402
from wgpu_n_body_amd.sharded import morton_domains  # noqa: E402

META, BUILD, WALK, MIGRATE = 2, 3, 4, 5
REC = 32


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", type=int, default=1 << 22)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--theta", type=float, default=0.5)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--init", default="uniform")
    ap.add_argument("--migrate-every", type=int, default=1)
    ap.add_argument("--skip-baselines", action="store_true")
    ap.add_argument("--export-mode", type=int, default=None, help="tree_let_export_mode: 1 one launch (default), 0 a launch per level")
    ap.add_argument("--count-visits", action="store_true", help="one extra counted step at the end")
    args = ap.parse_args()
    W, n = args.world, args.bodies
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    sp = nb.SimParams(particle_num=n)
    p = getattr(nb.inits, args.init + "_init")(sp, seed=args.seed)

    t0 = time.perf_counter()
    order, cuts, splits, ref_bound = morton_domains(p, W, with_owners=True)
    t_order = time.perf_counter() - t0
    one = torch.cuda.Stream(dev)   # all ranks on ONE stream: each rank's kernels run alone, back to back
    streams = [one for _ in range(W)]
    sims = []
    capacity = int(1.25 * max(cuts[r + 1] - cuts[r] for r in range(W))) + 4096
    mig_cap = max(1024, capacity // 8)
    for r in range(W):
        mine = p[order[cuts[r]:cuts[r + 1]]]
        padded = np.zeros(capacity, dtype=mine.dtype)
        padded[:len(mine)] = mine
        s = nb.TreeSim.from_particles(nb.SimParams(particle_num=capacity), nb.AddParams.TreeSimParams(args.theta),
                                      padded, nb.Placement(0, 0, 1, streams[r].cuda_stream))
        s.set_tuning("tree_let_world", W)
        s.set_tuning("tree_let_rank", r)
        s.set_tuning("tree_let_active", len(mine))
        s.set_tuning("tree_let_cap", 2 * capacity + 64)
        if args.export_mode is not None:
            s.set_tuning("tree_let_export_mode", args.export_mode)
        s.let_set_owners(splits, ref_bound, mig_cap)
        sims.append(s)

    def view(ptr, nbytes):
        from wgpu_n_body_amd.sharded import _DevicePtr
        return torch.as_tensor(_DevicePtr(ptr, nbytes // 4), device=dev)

    regs = [[s.exchange_region(k) for k in range(7)] for s in sims]
    views = [[view(ptr, tot) for (ptr, _o, _l, tot) in rg] for rg in regs]

    def all_gather(k):
        torch.cuda.synchronize()
        for src in range(W):
            _p, off, ln, _t = regs[src][k]
            for dst in range(W):
                if dst != src:
                    views[dst][k][off // 4:(off + ln) // 4].copy_(views[src][k][off // 4:(off + ln) // 4])
        torch.cuda.synchronize()

    # This tool leaves the GPU idle while the host shuffles the exchanges, and an idling MI355X
    # drops its clock (a walk measured 1.1 ms right after start-up and 5 ms ten iterations later
    # with identical wave-cycle counts).  On a node every GPU is busy all the time, so the clock
    # is pulled up with ~40 ms of all-pairs work on the same stream right before each timed phase.
    spw = nb.SimParams(particle_num=65536)
    warm = nb.NaiveSim.from_particles(spw, None, nb.inits.uniform_init(spw, seed=1),
                                      nb.Placement(0, 0, 1, one.cuda_stream))

    def keep_warm():
        for _ in range(32):
            warm.encode()

    def all_to_all(counts, k_send, k_recv, R):
        received = []
        for me in range(W):
            offs, recv = 0, []
            for r in range(W):
                c = 0 if r == me else int(counts[r, me])
                recv.append(c)
                if c:
                    seg = regs[r][k_send][2] // 4
                    views[me][k_recv][offs * R:(offs + c) * R].copy_(views[r][k_send][me * seg:me * seg + c * R])
                offs += c
            received.append(recv)
        torch.cuda.synchronize()
        return received

    def matrix(k):
        all_gather(k)
        return views[0][k].view(torch.int32).cpu().numpy().astype(np.int64).reshape(W, W)

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(8)] for _ in range(W)]
    acc = np.zeros((W, 4))
    migrated = 0
    let_records = None
    for it in range(args.warmup + args.steps):
        do_mig = args.migrate_every > 0 and it > 0 and it % args.migrate_every == 0
        if do_mig:
            keep_warm()
            for r, s in enumerate(sims):
                with torch.cuda.stream(streams[r]):
                    ev[r][6].record()
                    s.encode_phase(MIGRATE)
                    ev[r][7].record()
            mc = matrix(4)
            received = all_to_all(mc, 5, 6, 12)
            for me, s in enumerate(sims):
                s.let_set_arrivals(int(mc[me, me]), received[me])
            if it >= args.warmup:
                migrated += int(mc.sum() - np.trace(mc))
        keep_warm()
        for r, s in enumerate(sims):
            with torch.cuda.stream(streams[r]):
                ev[r][0].record()
                s.encode_phase(META)
                ev[r][1].record()
        all_gather(0)
        keep_warm()
        for r, s in enumerate(sims):
            with torch.cuda.stream(streams[r]):
                ev[r][2].record()
                s.encode_phase(BUILD)
                ev[r][3].record()
        counts = matrix(1)
        received = all_to_all(counts, 2, 3, 8)
        for me, s in enumerate(sims):
            s.let_set_imports(received[me])
        keep_warm()
        for r, s in enumerate(sims):
            with torch.cuda.stream(streams[r]):
                ev[r][4].record()
                s.encode_phase(WALK)
                ev[r][5].record()
        torch.cuda.synchronize()
        if os.environ.get("NB_LET_TRACE"):
            print("iter", it, "walk ms per rank", [round(ev[r][4].elapsed_time(ev[r][5]), 2) for r in range(W)],
                  "build", [round(ev[r][2].elapsed_time(ev[r][3]), 2) for r in range(W)], flush=True)
        if it >= args.warmup:
            for r in range(W):
                acc[r] += [ev[r][0].elapsed_time(ev[r][1]), ev[r][2].elapsed_time(ev[r][3]),
                           ev[r][4].elapsed_time(ev[r][5]), ev[r][6].elapsed_time(ev[r][7]) if do_mig else 0.0]
        let_records = counts
    acc /= args.steps
    off = ~np.eye(W, dtype=bool)
    imported = np.array([let_records[:, me].sum() - let_records[me, me] for me in range(W)])
    out = {
        "metric": "Barnes-Hut LET step, per-rank cost measured on one GPU",
        "bodies": n, "world": W, "theta": args.theta, "init": args.init, "bodies_per_rank": n // W,
        "ms_meta": {"mean": acc[:, 0].mean(), "max": acc[:, 0].max()},
        "ms_build_and_export": {"mean": acc[:, 1].mean(), "max": acc[:, 1].max()},
        "ms_walk": {"mean": acc[:, 2].mean(), "max": acc[:, 2].max()},
        "ms_migrate": {"mean": acc[:, 3].mean(), "max": acc[:, 3].max()},
        "migrate_every": args.migrate_every, "bodies_migrated_per_step": migrated / args.steps,
        "ms_rank_total": {"mean": acc.sum(axis=1).mean(), "max": acc.sum(axis=1).max()},
        "ms_walk_per_rank": [round(float(x), 3) for x in acc[:, 2]],
        "let_records_per_pair": {"mean": float(let_records[off].mean()), "max": int(let_records[off].max())},
        "imported_MB_per_rank": {"mean": float(imported.mean() * REC / 1e6), "max": float(imported.max() * REC / 1e6)},
        "replicated_scheme_MB_received_per_rank": 48.0 * n * (W - 1) / W / 1e6,
        "host_morton_order_s": t_order,
    }
    if args.count_visits:
        for s in sims:
            s.set_tuning("tree_count_visits", 1)
        c_before = [s.debug_buffer("counters", np.uint64).copy() for s in sims]
        for s in sims:
            s.encode_phase(META)
        all_gather(0)
        for s in sims:
            s.encode_phase(BUILD)
        counts = matrix(1)
        received = all_to_all(counts, 2, 3, 8)
        for me, s in enumerate(sims):
            s.let_set_imports(received[me])
        for s in sims:
            s.encode_phase(WALK)
        cs = [s.debug_buffer("counters", np.uint64) - b for s, b in zip(sims, c_before)]
        out["visits_per_body"] = [float(c[0]) / (n / W) for c in cs]
        out["wave_cells_per_wave"] = [float(c[2]) / (n / W / 64) for c in cs]
        out["longest_wave_cells"] = [int(s.debug_buffer("counters", np.uint64)[5]) for s in sims]
    for s in sims:
        s.destroy()
    warm.destroy()
    if not args.skip_baselines:
        single = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(args.theta), p)
        for _ in range(3):
            single.encode()
        single.wait()
        tot, walk = single.encode_n_timed(args.steps)
        out["single_gpu_ms_per_step"] = tot / args.steps
        out["single_gpu_walk_ms"] = walk
        single.destroy()
        # the replicated-tree scheme, one rank of it with the GPU to itself: full build + 1/W of the walk
        # (only the first step of such a simulator is meaningful here: nobody all-gathers the other
        # ranks' slices for it, so time one step of a fresh simulator, a few times over)
        for _ in range(4):
            rep = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(args.theta), p,
                                            nb.Placement(0, W // 2, W))
            tot, walk = rep.encode_n_timed(1)
            rep.destroy()
        out["replicated_scheme_rank_ms"] = tot
        out["replicated_scheme_rank_walk_ms"] = walk
    print(json.dumps(out))


if __name__ == "__main__":
    main()
