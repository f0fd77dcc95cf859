"""Multi-GPU all-pairs: one process per GPU, bodies partitioned by contiguous index range,
one in-place all-gather of the new position/mass slices per step (RCCL over xGMI through
torch.distributed).  There is no reference counterpart: the reference is single-adapter
(src/runners/offline_headless.rs:22-31).

Why this shards with exactly one exchange: naive.wgsl's update is Jacobi-style -- body i's
step reads only the PREVIOUS step's positions of every body (naive.wgsl:34, src buffer) and
writes only its own slot of the dst buffer (naive.wgsl:68), and the two buffers ping-pong
(naive.rs:113-132).  So each rank advances its own bodies from the full old position array,
then every rank needs every other rank's new positions: an all-gather of float4{x,y,z,m}
slices, 16 B per body.  Velocities and accelerations never leave their owner.

torch is plumbing here (device memory, streams, the process group); the step itself is the
HIP kernel behind the C ABI, enqueued on torch's current stream so the collective orders
after it without host synchronisation.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import (AddParams, NaiveSim, Placement, SimParams, TreeSim, as_particles,
               shard_bodies_per_rank)


@dataclass(frozen=True)
class ShardPlan:
    """Which bodies each rank owns: rank r has [r*per, min(n, (r+1)*per)); `per` is a multiple
    of the kernels' tile granule so every rank's slice has the same padded length."""
    n: int
    world: int

    @property
    def per_rank(self) -> int:
        return shard_bodies_per_rank(self.n, self.world)

    @property
    def padded(self) -> int:
        return self.per_rank * self.world

    def range(self, rank: int):
        per = self.per_rank
        return min(self.n, rank * per), min(self.n, (rank + 1) * per)


class ShardedStepper:
    """Host logic of the sharded step, independent of where the local step runs.

    Subclasses provide `_local_step(src, dst)`: advance this rank's bodies reading the full
    `src` position buffer and writing rows [lo, hi) of `dst`.  `posm` are two torch tensors
    of shape [padded, 4] that ping-pong, exactly like the reference's two particle buffers.
    """

    def __init__(self, plan: ShardPlan, rank: int, posm, group=None):
        self.plan, self.rank, self.group = plan, rank, group
        self.posm = posm
        self.cur = 0          # posm[cur] holds the current positions of ALL bodies
        self.step_num = 0
        self.lo, self.hi = plan.range(rank)

    def _local_step(self, src, dst) -> None:  # pragma: no cover - abstract
        raise NotImplementedError

    def exchange(self, buf) -> None:
        """In-place all-gather: every rank contributes rows [rank*per, (rank+1)*per)."""
        import torch.distributed as dist
        if self.plan.world == 1:
            return
        per = self.plan.per_rank
        mine = buf[self.rank * per:(self.rank + 1) * per]
        dist.all_gather_into_tensor(buf.view(-1), mine.reshape(-1), group=self.group)

    def encode(self, events=None) -> None:
        """One sharded step.  `events`: optional (start, end) timing events recorded around
        the local step only (bench.py's kernel-duration measurement)."""
        src, dst = self.posm[self.cur], self.posm[self.cur ^ 1]
        if events:
            events[0].record()
        self._local_step(src, dst)
        if events:
            events[1].record()
        self.exchange(dst)
        self.cur ^= 1
        self.step_num += 1


class ShardedNaiveSim(ShardedStepper):
    """The product path: nb_naive.hip for the local step, RCCL for the exchange."""

    def __init__(self, sim_params: SimParams, particles, rank: int, world: int,
                 device_index: int, group=None, variant: Optional[int] = None):
        import torch
        plan = ShardPlan(sim_params.particle_num, world)
        dev = torch.device("cuda", device_index)
        posm = [torch.zeros(plan.padded, 4, dtype=torch.float32, device=dev) for _ in range(2)]
        super().__init__(plan, rank, posm, group)
        self._torch = torch
        self._dev = dev
        # A dedicated (non-null) stream: the kernels and the collective's stream dependencies
        # are both expressed on it.  (torch's default stream is the null stream, whose handle
        # 0 means "create your own" to nb_placement.)
        self.stream = torch.cuda.Stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))  # after the zero fills above
        self.sim = NaiveSim.from_particles(
            sim_params, None, as_particles(particles),
            Placement(device_index, rank, world, self.stream.cuda_stream,
                      (posm[0].data_ptr(), posm[1].data_ptr())))
        if variant is not None:
            self.sim.set_tuning("naive_variant", variant)
        # the simulator wrote the initial positions into its `cur` buffer (index 0)
        ptr = self.sim.exchange_region()[0]
        self.cur = 0 if ptr == posm[0].data_ptr() else 1

    def _local_step(self, src, dst) -> None:
        self.sim.encode()            # async, on self.stream
        ptr = self.sim.exchange_region()[0]
        assert ptr == dst.data_ptr(), "ping-pong out of sync with the simulator"

    def encode(self, events=None) -> None:
        with self._torch.cuda.stream(self.stream):   # the collective orders against it
            super().encode(events)

    def cleanup(self) -> None:
        self.sim.cleanup()

    def wait(self) -> None:
        self.stream.synchronize()

    def read_particles(self) -> np.ndarray:
        """All positions/masses + this rank's velocities/accelerations (zero elsewhere)."""
        self._torch.cuda.synchronize(self._dev)
        return self.sim.dest_particle_slice()

    def destroy(self) -> None:
        self.sim.destroy()


class _DevicePtr:
    """Expose library-owned device memory to torch (zero-copy) via __cuda_array_interface__."""

    def __init__(self, ptr: int, nfloats: int):
        self.__cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4",
                                         "data": (ptr, False), "version": 3, "strides": None}


class ShardedTreeSim:
    """Barnes-Hut on several GPUs, SURVEY 8(e) step 1: replicated tree, partitioned walk.

    Every rank holds the full state, builds the identical octree (the build is deterministic)
    and walks only its contiguous range of the SORTED bodies; then the range's new positions,
    velocities and accelerations are all-gathered in place (three collectives: the next step
    re-sorts all bodies, so all three arrays must be complete everywhere).  The tree build is
    not sped up by more GPUs -- the spatial domain decomposition + LET exchange of the north
    star is the next step."""

    def __init__(self, sim_params: SimParams, theta: float, particles, rank: int, world: int,
                 device_index: int, group=None):
        import torch
        self._torch = torch
        self.rank, self.world, self.group = rank, world, group
        self._dev = torch.device("cuda", device_index)
        self.stream = torch.cuda.Stream(self._dev)
        self.sim = TreeSim.from_particles(
            sim_params, AddParams.TreeSimParams(theta), as_particles(particles),
            Placement(device_index, rank, world, self.stream.cuda_stream))
        self._views = {}
        self.step_num = 0

    def _view(self, ptr: int, total_bytes: int):
        t = self._views.get(ptr)
        if t is None:
            t = self._torch.as_tensor(_DevicePtr(ptr, total_bytes // 4), device=self._dev)
            self._views[ptr] = t
        return t

    def encode(self) -> None:
        import torch.distributed as dist
        with self._torch.cuda.stream(self.stream):
            self.sim.encode()
            if self.world > 1:
                for k in range(self.sim.exchange_count()):
                    ptr, off, ln, tot = self.sim.exchange_region(k)
                    full = self._view(ptr, tot)
                    dist.all_gather_into_tensor(full, full[off // 4:(off + ln) // 4], group=self.group)
        self.step_num += 1

    def cleanup(self) -> None:
        self.sim.cleanup()

    def wait(self) -> None:
        self.stream.synchronize()

    def read_particles(self) -> np.ndarray:
        self._torch.cuda.synchronize(self._dev)
        return self.sim.dest_particle_slice()

    def destroy(self) -> None:
        self._views.clear()
        self.sim.destroy()
