"""Worker for tests/test_naive_gpu.py::test_sharded_naive_sim_two_ranks_one_gpu: one rank of a
2-process run of the PRODUCT sharded path (wgpu_n_body_amd.sharded.ShardedNaiveSim: HIP local
step + in-place all-gather through torch.distributed) with both ranks on cuda:0 and the gloo
backend standing in for RCCL (RCCL refuses two ranks on one device).
argv: out_dir n steps [naive|naive-overlap|tree|tree-overlap|let]
"let": the LET Barnes-Hut class (LetTreeSim) for tests/test_let_gpu.py; writes rank<r>.npy with
the rank's own bodies."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import wgpu_n_body_amd as nb  # noqa: E402
from wgpu_n_body_amd.sharded import LetTreeSim, ShardedNaiveSim, ShardedTreeSim  # noqa: E402


def rccl_at_world_one(out_dir, n, steps):
    """The product's RCCL code paths on ONE rank (RCCL refuses several ranks per device): the
    collectives are issued all the same (force_exchange) -- all_gather_into_tensor on views of
    library-owned device memory, all_to_all on zero-length device views -- and every sharded class
    must reproduce its single simulator bit for bit."""
    import json
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
    torch.cuda.set_device(0)
    sp = nb.SimParams(particle_num=n)
    init = nb.inits.uniform_init(sp, seed=91)
    report = {}
    # all-pairs
    ShardedNaiveSim.force_exchange = True
    a = ShardedNaiveSim(sp, init, 0, 1, 0)
    b = nb.NaiveSim.from_particles(sp, None, init)
    for _ in range(steps):
        a.encode()
        b.encode()
    report["naive"] = bool(np.array_equal(a.read_particles(), b.dest_particle_slice()))
    a.destroy(); b.destroy()
    # replicated tree (three all-gathers)
    t = ShardedTreeSim(sp, 0.5, init, 0, 1, 0)
    t.world = 1
    one = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(0.5), init)
    import wgpu_n_body_amd.sharded as sh
    for _ in range(steps):
        # (world 1: ShardedTreeSim.encode skips its gathers; issue them by hand on the same views)
        t.encode()
        with torch.cuda.stream(t.stream):
            for k in range(t.sim.exchange_count()):
                ptr, off, ln, tot = t.sim.exchange_region(k)
                full = t._view(ptr, tot)
                dist.all_gather_into_tensor(full, full[off // 4:(off + ln) // 4])
        one.encode()
    report["tree"] = bool(np.array_equal(t.read_particles(), one.dest_particle_slice()))
    t.destroy()
    # LET protocol: all-gathers of region 0 / 1 / 4, all-to-all of (empty) segments, fixed-stride path
    LetTreeSim.force_exchange = True
    one2 = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(0.5), init)
    let = LetTreeSim(sp, 0.5, init, 0, 1, 0, migrate_every=2)
    syncs = []
    for _ in range(steps + 3):
        before = let.host_syncs
        let.encode()
        one2.encode()
        syncs.append(let.host_syncs - before)
    f = lambda x: nb.as_floats(x)[np.argsort(nb.as_floats(x)[:, 0], kind="stable")]
    report["let"] = bool(np.array_equal(f(let.read_local()), f(one2.dest_particle_slice())))
    report["let_async_used"] = bool(let._use_async())
    report["let_syncs_per_step"] = syncs
    let.destroy(); one.destroy(); one2.destroy()
    json.dump(report, open(os.path.join(out_dir, "rccl_world1.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def main():
    out_dir, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    mode = sys.argv[4] if len(sys.argv) > 4 else "naive"
    if mode == "rccl-world1":
        return rccl_at_world_one(out_dir, n, steps)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    sp = nb.SimParams(particle_num=n)
    rank_mode = mode
    if mode in ("let", "let-overlap", "let-rebalance", "let-async"):
        init = nb.inits.uniform_init(sp, seed=27).copy()   # = tests/test_let_gpu.py::tagged(nb, n, 27)
        nb.as_floats(init)[:, 9] = 1.0 + np.arange(n, dtype=np.float32) / np.float32(2 * n)
        if mode == "let-async":   # fixed-stride exchange, counts consumed on the device; migrate every 4th step
            sim = LetTreeSim(sp, 0.5, init, rank, world, 0, migrate_every=4, async_exchange=True)
        else:
            sim = LetTreeSim(sp, 0.5, init, rank, world, 0, overlap=(rank_mode == "let-overlap"))
        syncs = []
        for k in range(steps):
            if mode == "let-rebalance" and k == 2:
                sim.rebalance()
            before = sim.host_syncs
            sim.encode()
            sim.cleanup()
            syncs.append(sim.host_syncs - before)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), nb.as_floats(sim.read_local()))
        np.save(os.path.join(out_dir, f"syncs{rank}.npy"), np.array(syncs))
        if mode == "let-async":   # the collective read-out (tensor collectives) returns every body once
            everything = nb.as_floats(sim.read_particles())
            assert len(everything) == n and len(np.unique(everything[:, 9])) == n
        dist.barrier()
        sim.destroy()
        dist.destroy_process_group()
        return
    init = nb.inits.uniform_init(sp, seed=77)
    if mode.startswith("tree"):
        sim = ShardedTreeSim(sp, 0.5, init, rank, world, 0, overlap=(mode == "tree-overlap"))
        sim.lo, sim.hi = 0, n
    else:
        sim = ShardedNaiveSim(sp, init, rank, world, 0, variant=1, overlap=(mode == "naive-overlap"))
        sim.sim.set_tuning("naive_jsplit", 1)
    for _ in range(steps):
        sim.encode()
        sim.cleanup()
    sim.wait()
    got = nb.as_floats(sim.read_particles())
    np.savez(os.path.join(out_dir, f"gpu_rank{rank}.npz"), state=got, lo=sim.lo, hi=sim.hi)
    dist.barrier()
    sim.destroy()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
