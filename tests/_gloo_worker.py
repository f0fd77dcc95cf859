"""Worker for tests/test_sharded_gloo.py: one rank of a CPU (gloo) run of the sharded
all-pairs host logic (wgpu_n_body_amd.sharded.ShardedStepper), with the CPU oracle standing
in for the HIP local step.  argv: out_dir n steps kind seed"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402  (tests may use the oracle)
from tests.helpers import DT, E, G, make_state  # noqa: E402
from wgpu_n_body_amd.sharded import ShardedStepper, ShardPlan  # noqa: E402


class OracleSharded(ShardedStepper):
    """Same host logic as ShardedNaiveSim; the local step is the CPU oracle on this rank's
    body range (vel/acc kept locally, only positions/masses are exchanged)."""

    def __init__(self, plan, rank, state):
        posm = [torch.zeros(plan.padded, 4, dtype=torch.float32) for _ in range(2)]
        super().__init__(plan, rank, posm)
        self.n = plan.n
        self.va = state[:, 3:9].copy()           # only rows [lo,hi) are ever used/valid
        posm[0][: self.n] = torch.from_numpy(state[:, [0, 1, 2, 9]])
        posm[1].copy_(posm[0])

    def _step_remote(self, src, dst):
        full = np.zeros((self.n, 10), np.float32)
        full[:, [0, 1, 2, 9]] = src[: self.n].numpy()
        full[self.lo:self.hi, 3:9] = self.va[self.lo:self.hi]
        out = O.naive_step_f32(full, G, E, DT, self.lo, self.hi)
        dst[self.lo:self.hi] = torch.from_numpy(out[self.lo:self.hi][:, [0, 1, 2, 9]])
        self.va[self.lo:self.hi] = out[self.lo:self.hi, 3:9]


def main():
    out_dir, n, steps, kind, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    O.set_threads(2)
    state = make_state(kind, n, seed)
    sim = OracleSharded(ShardPlan(n, world), rank, state)
    for _ in range(steps):
        sim.encode()
    sim.finish_exchange()
    dist.barrier()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), posm=sim.posm[sim.cur][:n].numpy(),
             va=sim.va, lo=sim.lo, hi=sim.hi, step_num=sim.step_num)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
