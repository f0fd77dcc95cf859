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

from . import (AddParams, NaiveSim, NBodyError, Placement, SimParams, TreeSim, as_floats, as_particles,
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
    """Host logic of the sharded step, independent of where the local work runs.

    One step of rank r:   [own-tiles half]  ->  wait for the previous exchange  ->
                          [other-tiles half + integrate]  ->  start the exchange (async)  ->
                          [own-tiles half of the NEXT step, overlapping the exchange]
    The own-tiles half only reads this rank's slice of the position buffer, which the rank
    itself wrote, so it is safe while the in-place all-gather is still filling the other
    slices.  Subclasses provide `_step_local(src)` (may be a no-op) and `_step_remote(src, dst)`
    (everything else; must leave rows [lo, hi) of `dst` final).  `posm` are two torch tensors
    of shape [padded, 4] that ping-pong, exactly like the reference's two particle buffers.
    """

    overlap = True
    force_exchange = False    # issue the collective even at world == 1 (tools/host_overhead.py)

    def __init__(self, plan: ShardPlan, rank: int, posm, group=None):
        self.plan, self.rank, self.group = plan, rank, group
        self.posm = posm
        self.cur = 0          # posm[cur] holds the current positions of ALL bodies
        self.step_num = 0
        self.lo, self.hi = plan.range(rank)
        self._work = None            # the exchange in flight, if any
        self._local_issued = False   # own-tiles half of the next step already enqueued

    def _step_local(self, src) -> None:
        pass

    def _step_remote(self, src, dst) -> None:  # pragma: no cover - abstract
        raise NotImplementedError

    def start_exchange(self, buf) -> None:
        """In-place all-gather: every rank contributes rows [rank*per, (rank+1)*per)."""
        import torch.distributed as dist
        if self.plan.world == 1 and not self.force_exchange:
            return
        per = self.plan.per_rank
        mine = buf[self.rank * per:(self.rank + 1) * per]
        self._work = dist.all_gather_into_tensor(buf.view(-1), mine.reshape(-1), group=self.group,
                                                 async_op=True)

    def finish_exchange(self) -> None:
        """Order everything enqueued from now on after the exchange in flight."""
        if self._work is not None:
            self._work.wait()
            self._work = None

    def encode(self, events=None) -> None:
        """One sharded step.  `events`: optional (start, end) timing events recorded around
        the post-exchange half of the step."""
        src, dst = self.posm[self.cur], self.posm[self.cur ^ 1]
        if not self._local_issued:
            self._step_local(src)
        self.finish_exchange()                 # src is complete from here on
        if events:
            events[0].record()
        self._step_remote(src, dst)
        if events:
            events[1].record()
        self._local_issued = False
        self.start_exchange(dst)
        self.cur ^= 1
        self.step_num += 1
        if self.overlap and (self.plan.world > 1 or self.force_exchange):
            self._step_local(dst)              # next step's first half, beside the all-gather
            self._local_issued = True


class ShardedNaiveSim(ShardedStepper):
    """The product path: nb_naive.hip for the local work, RCCL for the exchange."""

    def __init__(self, sim_params: SimParams, particles, rank: int, world: int,
                 device_index: int, group=None, variant: Optional[int] = None,
                 overlap: bool = True):
        import torch
        plan = ShardPlan(sim_params.particle_num, world)
        dev = torch.device("cuda", device_index)
        posm = [torch.zeros(plan.padded, 4, dtype=torch.float32, device=dev) for _ in range(2)]
        super().__init__(plan, rank, posm, group)
        self.overlap = overlap
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

    def _step_local(self, src) -> None:
        if self.overlap:
            self.sim.encode_phase(0)     # async, on self.stream: the rank's own j tiles

    def _step_remote(self, src, dst) -> None:
        if self.overlap:
            self.sim.encode_phase(1)     # the other j tiles, the integrator, the ping-pong flip
        else:
            self.sim.encode()            # the whole step in one go (no overlap with the exchange)

    def encode(self, events=None) -> None:
        with self._torch.cuda.stream(self.stream):   # the collective orders against it
            super().encode(events)

    def cleanup(self) -> None:
        self.sim.cleanup()

    def wait(self) -> None:
        with self._torch.cuda.stream(self.stream):
            self.finish_exchange()
        self.stream.synchronize()

    def read_particles(self) -> np.ndarray:
        """All positions/masses + this rank's velocities/accelerations (zero elsewhere)."""
        self.wait()
        self._torch.cuda.synchronize(self._dev)
        return self.sim.dest_particle_slice()

    def destroy(self) -> None:
        self.wait()
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
    re-sorts all bodies, so all three arrays must be complete everywhere).

    The build needs only positions and masses, so the step runs in two halves
    (nb_sim_encode_phase) and the host loop is software-pipelined: positions are gathered
    first, the NEXT step's sort + tree build is enqueued as soon as they have landed, and the
    velocity/acceleration gathers run beside it; only the walk waits for them.

    The tree build itself is not sped up by more GPUs and the whole state crosses the links
    every step: LetTreeSim below (spatial domains + LET exchange) is the scheme that scales."""

    def __init__(self, sim_params: SimParams, theta: float, particles, rank: int, world: int,
                 device_index: int, group=None, overlap: bool = True):
        import torch
        self._torch = torch
        self.rank, self.world, self.group, self.overlap = rank, world, group, overlap
        self._dev = torch.device("cuda", device_index)
        self.stream = torch.cuda.Stream(self._dev)
        self.sim = TreeSim.from_particles(
            sim_params, AddParams.TreeSimParams(theta), as_particles(particles),
            Placement(device_index, rank, world, self.stream.cuda_stream))
        self._views = {}
        self._works = []          # [positions, velocities, accelerations] gathers in flight
        self._build_issued = False
        self.step_num = 0

    def _view(self, ptr: int, total_bytes: int):
        t = self._views.get(ptr)
        if t is None:
            t = self._torch.as_tensor(_DevicePtr(ptr, total_bytes // 4), device=self._dev)
            self._views[ptr] = t
        return t

    def _wait(self, first: int, last: int) -> None:
        """Order what is enqueued next after gathers first..last-1 (0 = positions)."""
        for k in range(first, min(last, len(self._works))):
            if self._works[k] is not None:
                self._works[k].wait()
                self._works[k] = None

    def encode(self) -> None:
        import torch.distributed as dist
        with self._torch.cuda.stream(self.stream):
            if not self._build_issued:
                self._wait(0, 1)                  # positions complete
                self.sim.encode_phase(0)          # bound, keys, sort, reorder, tree build
            self._wait(0, 3)                      # velocities and accelerations complete
            self.sim.encode_phase(1)              # reorder v/a, walk + integrate this rank's range
            self._build_issued = False
            self._works = []
            if self.world > 1:
                for k in range(self.sim.exchange_count()):      # 0 positions, 1 vel, 2 acc
                    ptr, off, ln, tot = self.sim.exchange_region(k)
                    full = self._view(ptr, tot)
                    self._works.append(dist.all_gather_into_tensor(
                        full, full[off // 4:(off + ln) // 4], group=self.group, async_op=True))
                if self.overlap:
                    self._wait(0, 1)              # next step's build, beside the v/a gathers
                    self.sim.encode_phase(0)
                    self._build_issued = True
        self.step_num += 1

    def cleanup(self) -> None:
        self.sim.cleanup()

    def wait(self) -> None:
        with self._torch.cuda.stream(self.stream):
            self._wait(0, 3)
        self.stream.synchronize()

    def read_particles(self) -> np.ndarray:
        self.wait()
        self._torch.cuda.synchronize(self._dev)
        return self.sim.dest_particle_slice()

    def destroy(self) -> None:
        self.wait()
        self._views.clear()
        self.sim.destroy()


_TRACE = bool(__import__("os").environ.get("NB_LET_TRACE"))


def _now() -> float:
    import time
    return time.perf_counter()


def _morton_keys(particles: np.ndarray) -> np.ndarray:
    """63-bit Morton keys of the positions quantised to 21 bits per axis in the cube
    [-bound, bound]^3, bound = max |coordinate| (as a float32, the value the device is given):
    the same arithmetic as let_ref_key in nb_tree.hip."""
    f = as_floats(particles)
    pos = f[:, 0:3].astype(np.float64)
    bound = float(np.float32(max(float(np.abs(f[:, 0:3]).max()), 1e-30)))
    q = np.clip((pos + bound) / (2.0 * bound) * 2097152.0, 0.0, 2097151.0).astype(np.uint64)

    def spread(v):  # insert two zero bits after each of the 21 low bits
        v = v & np.uint64(0x1fffff)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v

    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))


def morton_order(particles: np.ndarray) -> np.ndarray:
    """Indices that sort bodies along a Morton (Z-order) curve of their positions -- used only to
    hand every rank a compact spatial domain at start-up (any space-filling order would do; the
    octree itself is keyed on the device by the reference's own descent)."""
    return np.argsort(_morton_keys(particles), kind="stable")


def morton_domains(particles: np.ndarray, world: int, slack: float = 0.02, with_owners: bool = False):
    """(order, cuts): bodies in Morton order and `world`+1 cut positions; rank r owns
    order[cuts[r]:cuts[r+1]].  Each cut starts at the equal-count position and snaps to the
    border of the COARSEST octree cell that lies within `slack` x (bodies per rank) of it: a
    domain that ends on a cell border has a tight bounding box, while a handful of bodies from
    the next cell would stretch the box -- and with it every peer's export -- across that cell.
    with_owners: also return (splits, ref_bound) for nb_sim_let_set_owners -- the first key of every
    domain but the first, and the half-width of the cube the keys were quantised in."""
    keys = _morton_keys(particles)
    order = np.argsort(keys, kind="stable")
    skeys = keys[order]
    n = len(order)
    cuts = [0]
    tol = max(1, int(slack * n / max(world, 1)))
    for r in range(1, world):
        c0 = (n * r) // world
        best = c0
        lo, hi = max(c0 - tol, cuts[-1] + 1), min(c0 + tol, n - 1)
        if lo <= hi and n > 1:
            for level in range(1, 22):                 # coarsest first
                shift = np.uint64(3 * (21 - level))
                seg = skeys[lo - 1:hi + 1] >> shift
                change = np.nonzero(seg[1:] != seg[:-1])[0]          # border between lo-1+j and lo+j
                if len(change):
                    cand = lo + change
                    best = int(cand[np.argmin(np.abs(cand - c0))])
                    break
        cuts.append(max(best, cuts[-1]))
    cuts.append(n)
    if not with_owners:
        return order, cuts
    # an empty tail domain owns nothing: its border is "past every key"
    splits = [int(skeys[c]) if c < n else (1 << 63) for c in cuts[1:-1]]
    pos = as_floats(particles)[:, 0:3]
    return order, cuts, splits, max(float(np.abs(pos).max()), 1e-30)


def exchange_segments(send, seg: int, recv, counts: np.ndarray, me: int, world: int, R: int, group=None,
                      sync=None, force: bool = False) -> list:
    """The all-to-all-v of the LET protocol.  `send` holds `world` segments of `seg` elements;
    counts[r][q] records of R elements each go from rank r to rank q.  Moves counts[r][me]
    records of every rank r's segment `me` into `recv`, packed in rank order (nothing for
    r == me), and returns the per-rank record counts received.
    RCCL: one grouped all-to-all on device views.  Other backends (gloo: the CPU tests and the
    one-GPU rehearsal): point-to-point through host memory (`sync` first, if the data is the
    product of enqueued device work).  force: issue the collective even at world == 1 (zero-length
    views; how the RCCL branch is exercised on a one-GPU box)."""
    import torch
    import torch.distributed as dist
    recv_counts = [0 if r == me else int(counts[r, me]) for r in range(world)]
    send_counts = [0 if q == me else int(counts[me, q]) for q in range(world)]
    offs = np.concatenate([[0], np.cumsum(recv_counts)])
    outs = [recv[int(offs[r]) * R:(int(offs[r]) + recv_counts[r]) * R] for r in range(world)]
    ins = [send[q * seg:q * seg + send_counts[q] * R] for q in range(world)]
    if world == 1 and not force:
        return recv_counts
    if dist.get_backend(group) == "nccl":
        dist.all_to_all(outs, ins, group=group)
        return recv_counts
    if sync is not None:
        sync()
    host_in = [x.cpu() for x in ins]
    host_out = [torch.empty(recv_counts[r] * R, dtype=recv.dtype) for r in range(world)]
    ops = []
    for peer in range(world):
        if peer == me:
            continue
        if send_counts[peer]:
            ops.append(dist.P2POp(dist.isend, host_in[peer], peer, group=group))
        if recv_counts[peer]:
            ops.append(dist.P2POp(dist.irecv, host_out[peer], peer, group=group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for r in range(world):
        if recv_counts[r]:
            outs[r].copy_(host_out[r])
    return recv_counts


class LetTreeSim:
    """Barnes-Hut on several GPUs, SURVEY 8(e) step 2: Morton-range domains, local octrees and a
    locally-essential-tree (LET) exchange (nb_sim_encode_phase(NB_PHASE_LET_*), include/nbody.h).

    Start-up: the bodies are ordered along a Morton curve and cut into `world` near-equal runs
    whose ends sit on octree-cell borders (morton_domains); rank r keeps run r for good
    (velocities and accelerations never leave their owner).  Per step:
      1. NB_PHASE_LET_META  -> all-gather 32 B per rank (local bound, box of the drifted bodies);
      2. NB_PHASE_LET_BUILD -> every rank builds the octree of ITS bodies inside the global root
         cube and prunes it against each peer's box; all-gather the `world` export counts, read
         them on the host (the one host synchronisation of the step), all-to-all the segments;
      3. NB_PHASE_LET_WALK  -> walk own tree + the imported trees, integrate.
    What a body feels is the sum of per-domain Barnes-Hut walks (each with the reference's
    per-body acceptance test); the pruning is decision-exact, so the result does not depend on
    the peers' boxes or on the exchange -- only on which bodies share a domain.

    Migration: a domain is a range of Morton keys in the start-up cube; every `migrate_every`
    steps (default: every step) the bodies that left their rank's range are handed to the new
    owner (NB_PHASE_LET_MIGRATE: one more small all-gather, host read and all-to-all).  Without
    it the leavers sort to the ends of their old rank's tree order, where they form spatially
    incoherent waves whose walks are ten times longer than the rest.  `rebalance()` (collective,
    host-side) re-cuts the domains from the current positions when the load has drifted."""

    META, BUILD, WALK, MIGRATE, WALK_OWN = 2, 3, 4, 5, 6
    HEADROOM = 1.25     # body capacity of a rank relative to its start-up share
    STRIDE_MARGIN = 1.5  # fixed-stride exchange: records per peer = margin x the largest count seen + 1024
    force_exchange = False   # issue the collectives even at world == 1 (one-GPU rehearsal of the RCCL path)

    def __init__(self, sim_params: SimParams, theta: float, particles, rank: int, world: int,
                 device_index: int, group=None, let_cap: Optional[int] = None, migrate_every: int = 1,
                 overlap: bool = False, async_exchange: Optional[bool] = None):
        import torch
        self._torch = torch
        self.rank, self.world, self.group = rank, world, group
        self.theta = float(theta)
        self.params = sim_params
        self._dev = torch.device("cuda", device_index)
        self._device_index = device_index
        self.stream = torch.cuda.Stream(self._dev)
        self.sim = None
        self.step_num = 0
        self._let_cap = let_cap
        self.migrate_every = int(migrate_every)
        # overlap: the rank's own tree is walked (NB_PHASE_LET_WALK_OWN) on the main stream while a
        # high-priority side stream gathers the export counts, reads them and runs the all-to-all.
        # Bit-identical to the plain order.  Off by default: the walk keeps every wave slot of
        # the GPU occupied (8 per SIMD), and up to 524,288 bodies per rank ALL its workgroups are
        # resident at once, so the collective's kernels only get a CU when the walk drains --
        # nothing is hidden until a rank holds more bodies than that (and two gloo ranks sharing
        # one GPU, the only rehearsal available here, run slower with it).
        self.overlap = bool(overlap)
        self.side = torch.cuda.Stream(self._dev, priority=-1)   # its small kernels must not queue behind the walk
        # async_exchange: steps between migrations issue NO host synchronisation -- the export counts
        # are all-gathered and consumed on the device (nb_sim_let_set_import_stride) and a fixed
        # number of records per peer is moved, sized from the counts of two steps ago (read back
        # asynchronously) plus a margin.  Default: on under RCCL, off under gloo (whose exchange
        # goes through host memory anyway).
        self.async_exchange = async_exchange
        self.host_syncs = 0          # blocking host reads of the CURRENT step's data issued by encode()
        self._readbacks = {}         # step -> (pinned counts tensor, event)
        self._known_counts = {}      # step -> counts matrix (numpy), from a synchronous step or a read-back
        self._adopt(as_particles(particles))

    # -- domain set-up -------------------------------------------------------------------------
    def _adopt(self, particles: np.ndarray) -> None:
        # a new domain cut: the export counts seen so far say nothing about it, so the next two steps
        # read their counts on the host again before a fixed stride is planned from them
        self._known_counts.clear()
        self._readbacks.clear()
        order, cuts, splits, ref_bound = morton_domains(particles, self.world, with_owners=True)
        self.counts = [cuts[r + 1] - cuts[r] for r in range(self.world)]
        mine = particles[order[cuts[self.rank]:cuts[self.rank + 1]]]
        if self.sim is not None:
            self.sim.destroy()
        capacity = int(self.HEADROOM * max(self.counts)) + 4096
        padded = np.zeros(capacity, dtype=mine.dtype)
        padded[:len(mine)] = mine
        sp = SimParams(particle_num=capacity, g=self.params.g, e=self.params.e, dt=self.params.dt)
        self.sim = TreeSim.from_particles(sp, AddParams.TreeSimParams(self.theta), padded,
                                          Placement(self._device_index, 0, 1, self.stream.cuda_stream))
        # a peer can need at most this rank's whole octree (< 2 n + 1 nodes for distinct bodies)
        cap = self._let_cap or (2 * capacity + 64)
        self.sim.set_tuning("tree_let_world", self.world)
        self.sim.set_tuning("tree_let_rank", self.rank)
        self.sim.set_tuning("tree_let_active", len(mine))
        self.sim.set_tuning("tree_let_cap", int(cap))
        self.cap = int(cap)
        self.mig_cap = max(1024, capacity // 8)       # leavers per destination per migration
        self.sim.let_set_owners(splits, ref_bound, self.mig_cap)
        t = self._torch
        self._views = []
        for k in range(self.sim.exchange_count()):
            ptr, off, ln, tot = self.sim.exchange_region(k)
            self._views.append((t.as_tensor(_DevicePtr(ptr, tot // 4), device=self._dev), off // 4, ln // 4))

    # -- one step --------------------------------------------------------------------------------
    def _all_gather(self, k: int) -> None:
        import torch.distributed as dist
        full, off, ln = self._views[k]
        if self.world > 1 or self.force_exchange:
            dist.all_gather_into_tensor(full, full[off:off + ln], group=self.group)

    def _exchange_segments(self, counts: np.ndarray, k_send: int, k_recv: int, R: int) -> list:
        send, _, seg = self._views[k_send]
        recv, _, _ = self._views[k_recv]
        return exchange_segments(send, seg, recv, counts, self.rank, self.world, R, self.group,
                                 sync=lambda: self._torch.cuda.current_stream(self._dev).synchronize(),
                                 force=self.force_exchange)

    def _counts_matrix(self, k: int) -> np.ndarray:
        t = self._torch
        self._all_gather(k)
        self.host_syncs += 1          # a blocking read of this step's counts
        return self._views[k][0].view(t.int32).cpu().numpy().astype(np.int64).reshape(self.world, self.world)

    def _use_async(self) -> bool:
        if self.async_exchange is not None:
            return bool(self.async_exchange)
        import torch.distributed as dist
        return (self.world > 1 or self.force_exchange) and dist.is_initialized() and \
            dist.get_backend(self.group) == "nccl"

    def _planned_stride(self):
        """Records per peer for this step's fixed-stride exchange, or None when the counts of two
        steps ago are not known yet (the first two steps, and the two after the domains were cut anew:
        _adopt; a migration hands over a few dozen bodies, which the stride's margin covers).  Every rank
        derives it from the same all-gathered matrix of the same step, so all ranks agree."""
        k = self.step_num - 2
        if k in self._readbacks:                     # complete by now unless the host runs > 2 steps ahead
            pinned, ev = self._readbacks.pop(k)
            ev.synchronize()
            self._known_counts[k] = pinned.numpy().astype(np.int64).reshape(self.world, self.world).copy()
        for old in [j for j in self._known_counts if j < k]:
            del self._known_counts[old]
        c = self._known_counts.get(k)
        if c is None:
            return None
        off = c - np.diag(np.diag(c))
        need = int(off.max(initial=0))
        stride = int(need * self.STRIDE_MARGIN) + 1024
        return min((stride + 63) // 64 * 64, self.cap)

    def _queue_counts_readback(self) -> None:
        """Copy this step's all-gathered counts matrix to pinned host memory without waiting."""
        t = self._torch
        src = self._views[1][0].view(t.int32)
        pinned = t.empty(src.shape, dtype=t.int32, pin_memory=True)
        pinned.copy_(src, non_blocking=True)
        ev = t.cuda.Event()
        ev.record(t.cuda.current_stream(self._dev))
        self._readbacks[self.step_num] = (pinned, ev)

    def migrate(self) -> None:
        """Hand the bodies that left this rank's key range to their new owners (collective)."""
        t = self._torch
        with t.cuda.stream(self.stream):
            self.sim.encode_phase(self.MIGRATE)
            counts = self._counts_matrix(4)
            if (counts - np.diag(np.diag(counts))).max(initial=0) > self.mig_cap:
                raise NBodyError("LET migration: more leavers for one rank than the segment holds "
                                 f"({int(counts.max())} > {self.mig_cap}); rebalance() first")
            recv = self._exchange_segments(counts, 5, 6, 12)
            self.sim.let_set_arrivals(int(counts[self.rank, self.rank]), recv)
        self.counts = [int(counts[:, r].sum()) for r in range(self.world)]
        self.last_migration = counts
        self.host_syncs += 1

    def encode(self) -> None:
        t = self._torch
        if self.migrate_every > 0 and self.step_num > 0 and self.step_num % self.migrate_every == 0:
            self.migrate()
        with t.cuda.stream(self.stream):
            self.sim.encode_phase(self.META)
            self._all_gather(0)
            self.sim.encode_phase(self.BUILD)
            if self.overlap and self.world > 1:
                built = t.cuda.Event()
                built.record(self.stream)
                self.sim.encode_phase(self.WALK_OWN)         # main stream: needs nothing from the peers
                with t.cuda.stream(self.side):               # side stream: counts -> host -> all-to-all
                    self.side.wait_event(built)
                    _t0 = _now()
                    counts = self._counts_matrix(1)
                    _t1 = _now()
                    recv_counts = self._exchange_segments(counts, 2, 3, 8)
                    _t2 = _now()
                    arrived = t.cuda.Event()
                    arrived.record(self.side)
                self.stream.wait_event(arrived)
            else:
                stride = self._planned_stride() if self._use_async() else None
                if stride is not None:
                    # no host round trip: counts all-gathered and consumed on the device, a fixed
                    # number of records per peer on the wire
                    self._all_gather(1)
                    fixed = np.full((self.world, self.world), stride, dtype=np.int64)
                    np.fill_diagonal(fixed, 0)
                    self._exchange_segments(fixed, 2, 3, 8)
                    self._queue_counts_readback()
                    self.sim.let_set_import_stride(stride)
                    self.sim.encode_phase(self.WALK)
                    self.last_stride = stride
                    self.step_num += 1
                    return
                _t0 = _now()
                counts = self._counts_matrix(1)
                _t1 = _now()
                recv_counts = self._exchange_segments(counts, 2, 3, 8)
                _t2 = _now()
            if _TRACE:
                print(f"[let rank {self.rank} step {self.step_num}] counts {(_t1 - _t0) * 1e3:.2f} ms, "
                      f"exchange {(_t2 - _t1) * 1e3:.2f} ms", flush=True)
            self.last_counts = counts
            self._known_counts[self.step_num] = counts
            self.sim.let_set_imports(recv_counts)
            self.sim.encode_phase(self.WALK)
        self.step_num += 1

    def cleanup(self) -> None:
        self.sim.cleanup()

    def wait(self) -> None:
        self.stream.synchronize()
        self.sim.wait()   # ... and the device status words: a peer with more records than the fixed stride is
                          # raised here (nbody.h: at wait), not at the next read-back

    # -- read-out ----------------------------------------------------------------------------------
    def read_local(self) -> np.ndarray:
        """This rank's bodies (post-step), in the rank's current sorted order."""
        self.wait()
        return self.sim.dest_particle_slice()

    def read_particles(self) -> np.ndarray:
        """All bodies, rank by rank (collective).  The order is not the input order: like the
        reference's TreeSim, every step leaves the bodies in tree order."""
        import torch.distributed as dist
        mine = self.read_local()
        if self.world == 1:
            return mine
        # tensor collectives (no pickling): the body counts, then the 40-byte rows padded to the
        # largest count -- on the device under RCCL, on the host under gloo
        t = self._torch
        dev = self._dev if dist.get_backend(self.group) == "nccl" else t.device("cpu")
        nloc = t.tensor([len(mine)], dtype=t.int64, device=dev)
        counts = t.zeros(self.world, dtype=t.int64, device=dev)
        dist.all_gather_into_tensor(counts, nloc, group=self.group)
        counts = [int(c) for c in counts.cpu()]
        width = max(counts) * 10
        row = t.zeros(width, dtype=t.float32, device=dev)
        row[:len(mine) * 10] = t.from_numpy(as_floats(mine).reshape(-1).copy()).to(dev)
        out = t.zeros(self.world * width, dtype=t.float32, device=dev)
        dist.all_gather_into_tensor(out, row, group=self.group)
        host = out.cpu().numpy().reshape(self.world, width)
        return np.concatenate([host[r, :counts[r] * 10].reshape(-1, 10).view(mine.dtype).reshape(-1)
                               for r in range(self.world)])

    def rebalance(self) -> None:
        """Re-cut the domains from the current positions (collective, host-side: gathers all
        bodies, re-orders them along the Morton curve and re-creates the local simulator)."""
        self._adopt(self.read_particles().copy())

    def destroy(self) -> None:
        self.wait()
        self._views = []
        if self.sim is not None:
            self.sim.destroy()
            self.sim = None
