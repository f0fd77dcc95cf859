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


def main():
    out_dir, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    mode = sys.argv[4] if len(sys.argv) > 4 else "naive"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    sp = nb.SimParams(particle_num=n)
    rank_mode = mode
    if mode in ("let", "let-overlap", "let-rebalance"):
        init = nb.inits.uniform_init(sp, seed=27).copy()   # = tests/test_let_gpu.py::tagged(nb, n, 27)
        nb.as_floats(init)[:, 9] = 1.0 + np.arange(n, dtype=np.float32) / np.float32(2 * n)
        sim = LetTreeSim(sp, 0.5, init, rank, world, 0, overlap=(rank_mode == "let-overlap"))
        for k in range(steps):
            if mode == "let-rebalance" and k == 2:
                sim.rebalance()
            sim.encode()
            sim.cleanup()
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), nb.as_floats(sim.read_local()))
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
