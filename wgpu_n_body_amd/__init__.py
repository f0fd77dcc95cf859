"""wgpu_n_body_amd -- MI355X-native drop-in for the simulation hot path of
arpan-dhatt/wgpu-n-body (its `sims` + `runners::OfflineHeadless` + `inits` API).

Python host-side mirror of the reference's interface for this path, over the C ABI of
include/nbody.h (libnbody_hip.so: hand-written HIP kernels for gfx950).  Names follow
the reference crate:

    reference (Rust)                               here
    ---------------------------------------------  -----------------------------------
    sims::SimParams            (sims/mod.rs:51-71)  SimParams
    sims::AddParams            (sims/mod.rs:18-23)  AddParams.NaiveSimParams / .TreeSimParams
    sims::Particle             (sims/mod.rs:9-16)   PARTICLE_DTYPE (numpy, 40 B)
    trait sims::Simulator      (sims/mod.rs:73-90)  Simulator (new/encode/dest_particle_slice/
                                                    sim_params/cleanup)
    sims::NaiveSim, TreeSim    (sims/mod.rs:4-5)    NaiveSim, TreeSim
    runners::OfflineHeadless   (offline_headless.rs) OfflineHeadless
    inits::{uniform,disc,spherical}_init (inits.rs) inits.uniform_init / disc_init / spherical_init

There is no CPU fallback: importing works without a GPU (so the ABI can be checked), but
constructing a simulator raises NBodyError(NB_ERR_NO_DEVICE) when no HIP device exists.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import (NB_NAIVE_SIM_PARAMS, NB_TREE_SIM_PARAMS, OCTANT_DTYPE, PARTICLE_DTYPE,
                   NBodyError, check)

__all__ = ["SimParams", "AddParams", "Placement", "Simulator", "NaiveSim", "TreeSim",
           "OfflineHeadless", "inits", "PARTICLE_DTYPE", "OCTANT_DTYPE", "NBodyError",
           "PARTICLES_PER_GROUP", "device_count", "version", "shard_bodies_per_rank",
           "shard_padded_bodies", "naive_variants"]

PARTICLES_PER_GROUP = 64  # sims/mod.rs:7


@dataclass(frozen=True)
class SimParams:
    """`struct SimParams` with `Default` (sims/mod.rs:51-71)."""
    particle_num: int = 10000
    g: float = 0.000001
    e: float = 0.0001
    dt: float = 0.016

    def to_c(self) -> _lib.nb_sim_params:
        return _lib.nb_sim_params(int(self.particle_num), float(self.g), float(self.e),
                                  float(self.dt))


@dataclass(frozen=True)
class AddParams:
    """`enum AddParams` (sims/mod.rs:18-23)."""
    kind: int = NB_NAIVE_SIM_PARAMS
    theta: float = 0.0

    @staticmethod
    def NaiveSimParams() -> "AddParams":
        return AddParams(NB_NAIVE_SIM_PARAMS, 0.0)

    @staticmethod
    def TreeSimParams(theta: float) -> "AddParams":
        return AddParams(NB_TREE_SIM_PARAMS, float(theta))

    def to_c(self) -> _lib.nb_add_params:
        return _lib.nb_add_params(int(self.kind), float(self.theta))


@dataclass(frozen=True)
class Placement:
    """Device + body shard of a simulator (nb_placement; no reference counterpart -- the
    reference is single-adapter, offline_headless.rs:22-31)."""
    device_id: int = 0
    rank: int = 0
    world: int = 1
    stream: int = 0                      # hipStream_t as an integer, 0 = own stream
    posm: Sequence[int] = (0, 0)         # optional caller-owned ping-pong buffers

    def to_c(self) -> _lib.nb_placement:
        p = _lib.nb_placement()
        p.device_id, p.rank, p.world = int(self.device_id), int(self.rank), int(self.world)
        p.stream = C.c_void_p(int(self.stream) or None)
        p.posm[0] = C.c_void_p(int(self.posm[0]) or None)
        p.posm[1] = C.c_void_p(int(self.posm[1]) or None)
        return p


InitFn = Callable[[SimParams], np.ndarray]


def version() -> str:
    return _lib.lib().nb_version().decode()


def device_count() -> int:
    return int(_lib.lib().nb_device_count())


def shard_bodies_per_rank(particle_num: int, world: int) -> int:
    return int(_lib.lib().nb_shard_bodies_per_rank(particle_num, world))


def shard_padded_bodies(particle_num: int, world: int) -> int:
    return int(_lib.lib().nb_shard_padded_bodies(particle_num, world))


def naive_variants() -> list:
    L = _lib.lib()
    return [L.nb_naive_variant_name(i).decode() for i in range(L.nb_naive_variant_count())]


def as_particles(a) -> np.ndarray:
    """View/convert an (n,10) float32 array or a structured array as PARTICLE_DTYPE[n]."""
    a = np.asarray(a)
    if a.dtype == PARTICLE_DTYPE:
        return np.ascontiguousarray(a)
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 10:
        raise ValueError("particles must be PARTICLE_DTYPE[n] or float32[n,10]")
    return a.view(PARTICLE_DTYPE).reshape(-1)


def as_floats(p: np.ndarray) -> np.ndarray:
    """PARTICLE_DTYPE[n] -> float32[n,10] view (px py pz vx vy vz ax ay az mass)."""
    return np.ascontiguousarray(p).view(np.float32).reshape(-1, 10)


class _Inits:
    """`mod inits` (src/inits.rs): seeded equivalents of the three distributions.  Each
    function has the reference's shape `fn(&SimParams) -> Vec<Particle>`; the seed (the
    reference uses the unseedable thread_rng) is an optional keyword."""

    @staticmethod
    def _run(name: str, sim_params: SimParams, seed: int) -> np.ndarray:
        out = np.zeros(sim_params.particle_num, dtype=PARTICLE_DTYPE)
        cp = sim_params.to_c()
        s = C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF)
        getattr(_lib.lib(), name)(C.byref(cp), out.ctypes.data, C.cast(C.byref(s), C.c_void_p))
        return out

    def uniform_init(self, sim_params: SimParams, seed: int = 0) -> np.ndarray:
        return self._run("nb_init_uniform", sim_params, seed)        # inits.rs:6-27

    def disc_init(self, sim_params: SimParams, seed: int = 0) -> np.ndarray:
        return self._run("nb_init_disc", sim_params, seed)           # inits.rs:29-54

    def spherical_init(self, sim_params: SimParams, seed: int = 0) -> np.ndarray:
        return self._run("nb_init_spherical", sim_params, seed)      # inits.rs:56-83


inits = _Inits()


def _init_trampoline(init_fn: InitFn, sim_params: SimParams):
    """Wrap a Python `init_fn(sim_params) -> particles` as an nb_init_fn C callback."""
    err = []

    def trampoline(_params_ptr, out_ptr, _user):
        try:
            arr = as_particles(init_fn(sim_params))
            if arr.shape[0] != sim_params.particle_num:
                raise ValueError(f"init_fn returned {arr.shape[0]} particles, expected "
                                 f"{sim_params.particle_num}")
            C.memmove(out_ptr, arr.ctypes.data, arr.nbytes)
        except BaseException as ex:  # never unwind through C
            err.append(ex)

    return _lib.NB_INIT_FN(trampoline), err


class Simulator:
    """`trait Simulator` (sims/mod.rs:73-90) over an nb_sim handle."""

    KIND = NB_NAIVE_SIM_PARAMS

    def __init__(self, handle: int, borrowed: bool = False):
        self._h = C.c_void_p(handle)
        self._borrowed = borrowed  # owned by an nb_runner

    # -- Simulator::new(device, sim_params, add_params, mappable_primary_buffers, init_fn) --
    @classmethod
    def new(cls, sim_params: SimParams, add_params: Optional[AddParams], init_fn: InitFn,
            placement: Optional[Placement] = None) -> "Simulator":
        L = _lib.lib()
        if add_params is None:
            add_params = AddParams(cls.KIND, 0.0)
        if add_params.kind != cls.KIND and cls is not Simulator:
            # the reference's TreeSim::new accepts any AddParams and falls back to theta
            # 0.75 with a warning (tree.rs:42-51); mirror that
            add_params = AddParams(cls.KIND, 0.0)
        sp, ap = sim_params.to_c(), add_params.to_c()
        pl = (placement or Placement()).to_c()
        cb, err = _init_trampoline(init_fn, sim_params)
        h = C.c_void_p()
        rc = L.nb_sim_create(C.byref(h), C.byref(sp), C.byref(ap), C.byref(pl),
                             C.cast(cb, C.c_void_p), None)
        if err:
            if rc == 0:
                L.nb_sim_destroy(h)
            raise err[0]
        check(rc)
        return cls(h.value)

    @classmethod
    def from_particles(cls, sim_params: SimParams, add_params: Optional[AddParams], particles,
                       placement: Optional[Placement] = None) -> "Simulator":
        L = _lib.lib()
        if add_params is None:
            add_params = AddParams(cls.KIND, 0.0)
        arr = as_particles(particles)
        sp, ap = sim_params.to_c(), add_params.to_c()
        pl = (placement or Placement()).to_c()
        h = C.c_void_p()
        check(L.nb_sim_create_from_particles(C.byref(h), C.byref(sp), C.byref(ap), C.byref(pl),
                                             arr.ctypes.data, arr.shape[0]))
        return cls(h.value)

    # -- Simulator::encode (+ queue.submit): enqueue one step, do not wait --
    def encode(self) -> None:
        check(_lib.lib().nb_sim_encode(self._h))

    def encode_phase(self, phase: int) -> None:
        """Half a step (nb_sim_encode_phase): 0 = own bodies' tiles, 1 = the rest + integrate."""
        check(_lib.lib().nb_sim_encode_phase(self._h, int(phase)))

    def let_set_imports(self, counts) -> None:
        """LET protocol (nb_sim_let_set_imports): records received from every rank this step."""
        arr = (C.c_uint32 * len(counts))(*[int(c) for c in counts])
        check(_lib.lib().nb_sim_let_set_imports(self._h, arr, len(counts)))

    def let_set_import_stride(self, stride: int) -> None:
        """LET protocol without a host round trip (nb_sim_let_set_import_stride): this step's imports
        are fixed-stride segments, their counts are read on the device."""
        check(_lib.lib().nb_sim_let_set_import_stride(self._h, int(stride)))

    def let_set_owners(self, splits, ref_bound: float, seg_cap: int) -> None:
        """LET migration (nb_sim_let_set_owners): rank r owns reference keys [splits[r-1], splits[r])."""
        world = len(splits) + 1
        arr = (C.c_ulonglong * max(1, len(splits)))(*[int(x) for x in splits])
        check(_lib.lib().nb_sim_let_set_owners(self._h, arr, world, float(ref_bound), int(seg_cap)))

    def let_set_arrivals(self, stay: int, counts) -> None:
        """LET migration (nb_sim_let_set_arrivals): bodies kept, bodies received from every rank."""
        arr = (C.c_uint32 * len(counts))(*[int(c) for c in counts])
        check(_lib.lib().nb_sim_let_set_arrivals(self._h, int(stay), arr, len(counts)))

    # -- Simulator::cleanup --
    def cleanup(self) -> None:
        check(_lib.lib().nb_sim_cleanup(self._h))

    # -- device.poll(Maintain::Wait) --
    def wait(self) -> None:
        check(_lib.lib().nb_sim_wait(self._h))

    # -- Simulator::sim_params --
    def sim_params(self) -> SimParams:
        sp = _lib.nb_sim_params()
        check(_lib.lib().nb_sim_sim_params(self._h, C.byref(sp)))
        return SimParams(sp.particle_num, sp.g, sp.e, sp.dt)

    # -- Simulator::dest_particle_slice: here the POST-step state, copied to the host --
    def dest_particle_slice(self) -> np.ndarray:
        n = self.sim_params().particle_num
        out = np.zeros(n, dtype=PARTICLE_DTYPE)
        check(_lib.lib().nb_sim_read_particles(self._h, out.ctypes.data, n))
        return out

    read_particles = dest_particle_slice

    def write_particles(self, particles) -> None:
        arr = as_particles(particles)
        check(_lib.lib().nb_sim_write_particles(self._h, arr.ctypes.data, arr.shape[0]))

    def step_num(self) -> int:
        v = C.c_uint64()
        check(_lib.lib().nb_sim_step_num(self._h, C.byref(v)))
        return int(v.value)

    def encode_n_timed(self, n: int):
        """n steps back to back -> (ms_total, ms_mean_force_kernel), HIP-event timed."""
        a, b = C.c_float(), C.c_float()
        check(_lib.lib().nb_sim_encode_n_timed(self._h, n, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def exchange_region(self, index: int = 0):
        """(device_ptr, offset_bytes, slice_bytes, total_bytes) of exchange region `index`."""
        p, o, s, t = C.c_void_p(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        check(_lib.lib().nb_sim_exchange_region_i(self._h, int(index), C.byref(p), C.byref(o),
                                                  C.byref(s), C.byref(t)))
        return int(p.value or 0), int(o.value), int(s.value), int(t.value)

    def exchange_count(self) -> int:
        c = C.c_int()
        check(_lib.lib().nb_sim_exchange_count(self._h, C.byref(c)))
        return int(c.value)

    def set_tuning(self, key: str, value: int) -> None:
        check(_lib.lib().nb_sim_set_tuning(self._h, key.encode(), int(value)))

    def debug_buffer(self, name: str, dtype) -> np.ndarray:
        """Testing hook (nb_sim_debug_buffer): a named internal device buffer as a numpy array."""
        nbytes = C.c_size_t()
        check(_lib.lib().nb_sim_debug_buffer(self._h, name.encode(), None, 0, C.byref(nbytes)))
        out = np.zeros(nbytes.value // np.dtype(dtype).itemsize, dtype=dtype)
        check(_lib.lib().nb_sim_debug_buffer(self._h, name.encode(), out.ctypes.data, out.nbytes,
                                             C.byref(nbytes)))
        return out

    def read_tree(self):
        """TreeSim: (octants[n_nodes], root_width) of the tree the last step built."""
        n = self.sim_params().particle_num
        cap = max(4 * n, 8)
        buf = np.zeros(cap, dtype=OCTANT_DTYPE)
        nn, rw = C.c_size_t(), C.c_float()
        check(_lib.lib().nb_sim_read_tree(self._h, buf.ctypes.data, cap, C.byref(nn), C.byref(rw)))
        return buf[: nn.value].copy(), float(rw.value)

    def tree_node_count(self) -> int:
        """TreeSim: number of octants of the tree the last step built (no copy of the tree)."""
        nn, rw = C.c_size_t(), C.c_float()
        check(_lib.lib().nb_sim_read_tree(self._h, None, 0, C.byref(nn), C.byref(rw)))
        return int(nn.value)

    def destroy(self) -> None:
        if self._h and not self._borrowed:
            _lib.lib().nb_sim_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class NaiveSim(Simulator):
    """`NaiveSim` (sims/naive.rs): all-pairs O(N^2)."""
    KIND = NB_NAIVE_SIM_PARAMS


class TreeSim(Simulator):
    """`TreeSim` (sims/tree.rs): Barnes-Hut octree."""
    KIND = NB_TREE_SIM_PARAMS


class OfflineHeadless:
    """`OfflineHeadless<T: Simulator>` (runners/offline_headless.rs:4-45) over nb_runner.

    OfflineHeadless(NaiveSim, sim_params, add_params, init_fn) ~
    OfflineHeadless::<NaiveSim>::new(sim_params, add_params, init_fn)."""

    def __init__(self, sim_type, sim_params: SimParams, add_params: Optional[AddParams],
                 init_fn: InitFn, device_id: int = -1, device_ids: Optional[Sequence[int]] = None,
                 let_migrate_every: Optional[int] = None):
        """device_ids: several GPUs of this process (nb_runner_create_multi; all-pairs: body ranges +
        peer stores, TreeSim: replicated tree + partitioned walk): rank r owns a contiguous body
        range on device_ids[r]; a device id may repeat.  let_migrate_every (TreeSim, with device_ids):
        Morton domains + LET exchange instead (nb_runner_create_multi_let)."""
        L = _lib.lib()
        if add_params is None or add_params.kind != sim_type.KIND:
            add_params = AddParams(sim_type.KIND, 0.0)
        sp, ap = sim_params.to_c(), add_params.to_c()
        cb, err = _init_trampoline(init_fn, sim_params)
        h = C.c_void_p()
        if device_ids is not None:
            ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
            if let_migrate_every is not None:
                rc = L.nb_runner_create_multi_let(C.byref(h), C.byref(sp), C.byref(ap), C.cast(cb, C.c_void_p), None,
                                                  ids, len(device_ids), int(let_migrate_every))
            else:
                rc = L.nb_runner_create_multi(C.byref(h), C.byref(sp), C.byref(ap), C.cast(cb, C.c_void_p), None,
                                              ids, len(device_ids))
        else:
            rc = L.nb_runner_create(C.byref(h), C.byref(sp), C.byref(ap), C.cast(cb, C.c_void_p), None,
                                    int(device_id))
        if err:
            if rc == 0:
                L.nb_runner_destroy(h)
            raise err[0]
        check(rc)
        self._h = h
        sim_h = L.nb_runner_sim(h)                 # NULL for a several-GPU runner
        self.sim = sim_type(sim_h, borrowed=True) if sim_h else None

    @classmethod
    def new(cls, sim_type, sim_params, add_params, init_fn, device_id: int = -1, device_ids=None,
            let_migrate_every=None):
        return cls(sim_type, sim_params, add_params, init_fn, device_id, device_ids, let_migrate_every)

    def step_num(self) -> int:
        v = C.c_uint64()
        check(_lib.lib().nb_runner_step_num(self._h, C.byref(v)))
        return int(v.value)

    def step(self) -> None:
        """offline_headless.rs:38-44: encode -> submit -> cleanup -> poll(Wait)."""
        check(_lib.lib().nb_runner_step(self._h))

    def step_n(self, n: int) -> None:
        check(_lib.lib().nb_runner_step_n(self._h, int(n)))

    def set_profiling(self, on: bool = True) -> None:
        """Measurement: timing events around every rank's kernels and its waits for the peers."""
        check(_lib.lib().nb_runner_set_profiling(self._h, 1 if on else 0))

    def rank_times(self, world: int):
        """(kernel_ms[world], wait_ms[world]) of the last step_n() with profiling on."""
        k, w = (C.c_float * world)(), (C.c_float * world)()
        check(_lib.lib().nb_runner_rank_times(self._h, k, w, world))
        return [float(x) for x in k], [float(x) for x in w]

    def read_particles(self) -> np.ndarray:
        n = self.sim_params().particle_num
        out = np.zeros(n, dtype=PARTICLE_DTYPE)
        check(_lib.lib().nb_runner_read_particles(self._h, out.ctypes.data, n))
        return out

    def sim_params(self) -> SimParams:
        sp = _lib.nb_sim_params()
        check(_lib.lib().nb_runner_sim_params(self._h, C.byref(sp)))
        return SimParams(sp.particle_num, sp.g, sp.e, sp.dt)

    def destroy(self) -> None:
        if self._h:
            if self.sim is not None:
                self.sim._h = C.c_void_p()
            _lib.lib().nb_runner_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
