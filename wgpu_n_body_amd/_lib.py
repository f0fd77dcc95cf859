"""ctypes declarations for libnbody_hip.so (include/nbody.h).  Fails loudly if the HIP
library is missing: there is no CPU fallback in the product."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# NB_LIB: an alternative build of the same library (kernel tuning experiments); default in-tree
LIB_PATH = os.environ.get("NB_LIB") or os.path.join(_PKG, "libnbody_hip.so")

# `struct Particle`, src/sims/mod.rs:9-16 -- 40 bytes
PARTICLE_DTYPE = np.dtype(
    [("position", "<f4", (3,)), ("velocity", "<f4", (3,)), ("acceleration", "<f4", (3,)),
     ("mass", "<f4")]
)
assert PARTICLE_DTYPE.itemsize == 40
# `struct Octant`, src/sims/tree.rs:605-622 -- 52 bytes
OCTANT_DTYPE = np.dtype(
    [("cog", "<f4", (3,)), ("mass", "<f4"), ("bodies", "<u4"), ("children", "<u4", (8,))]
)
assert OCTANT_DTYPE.itemsize == 52


class nb_sim_params(C.Structure):
    _fields_ = [("particle_num", C.c_uint32), ("g", C.c_float), ("e", C.c_float),
                ("dt", C.c_float)]


class nb_add_params(C.Structure):
    _fields_ = [("kind", C.c_int32), ("theta", C.c_float)]


class nb_placement(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32),
                ("stream", C.c_void_p), ("posm", C.c_void_p * 2)]


assert C.sizeof(nb_sim_params) == 16 and C.sizeof(nb_add_params) == 8

NB_INIT_FN = C.CFUNCTYPE(None, C.POINTER(nb_sim_params), C.c_void_p, C.c_void_p)

NB_OK, NB_ERR_INVALID, NB_ERR_NO_DEVICE, NB_ERR_HIP, NB_ERR_ALLOC, NB_ERR_UNSUPPORTED = range(6)
NB_NAIVE_SIM_PARAMS, NB_TREE_SIM_PARAMS = 0, 1

# every symbol include/nbody.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "nb_last_error", "nb_version", "nb_device_count",
    "nb_init_uniform", "nb_init_disc", "nb_init_spherical",
    "nb_shard_bodies_per_rank", "nb_shard_padded_bodies",
    "nb_sim_create", "nb_sim_create_from_particles", "nb_sim_encode", "nb_sim_encode_phase",
    "nb_sim_let_set_imports", "nb_sim_let_set_import_stride", "nb_sim_let_set_owners", "nb_sim_let_set_arrivals", "nb_sim_cleanup",
    "nb_sim_wait", "nb_sim_sim_params", "nb_sim_read_particles", "nb_sim_write_particles",
    "nb_sim_read_tree", "nb_sim_exchange_region", "nb_sim_exchange_count",
    "nb_sim_exchange_region_i", "nb_sim_step_num", "nb_sim_encode_n_timed",
    "nb_sim_set_tuning", "nb_sim_debug_buffer", "nb_naive_variant_count", "nb_naive_variant_name", "nb_sim_destroy",
    "nb_runner_create", "nb_runner_create_multi", "nb_runner_create_multi_let", "nb_runner_step_num", "nb_runner_step", "nb_runner_step_n", "nb_runner_read_particles",
    "nb_runner_set_profiling", "nb_runner_rank_times",
    "nb_runner_sim_params", "nb_runner_sim", "nb_runner_destroy",
]


class NBodyError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"nbody_hip error {code}: {message}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Load libnbody_hip.so.  Raises (no fallback) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m wgpu_n_body_amd.build` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
    P = C.POINTER
    L.nb_last_error.restype = C.c_char_p
    L.nb_version.restype = C.c_char_p
    L.nb_device_count.restype = C.c_int
    for name in ("nb_init_uniform", "nb_init_disc", "nb_init_spherical"):
        f = getattr(L, name)
        f.argtypes = [P(nb_sim_params), vp, vp]
        f.restype = None
    L.nb_shard_bodies_per_rank.argtypes = [sz, C.c_int]
    L.nb_shard_bodies_per_rank.restype = sz
    L.nb_shard_padded_bodies.argtypes = [sz, C.c_int]
    L.nb_shard_padded_bodies.restype = sz
    L.nb_sim_create.argtypes = [P(vp), P(nb_sim_params), P(nb_add_params), P(nb_placement), vp, vp]
    L.nb_sim_create_from_particles.argtypes = [P(vp), P(nb_sim_params), P(nb_add_params),
                                               P(nb_placement), vp, sz]
    for name in ("nb_sim_encode", "nb_sim_cleanup", "nb_sim_wait", "nb_sim_destroy"):
        getattr(L, name).argtypes = [vp]
    L.nb_sim_encode_phase.argtypes = [vp, C.c_int]
    L.nb_sim_let_set_imports.argtypes = [vp, P(C.c_uint32), C.c_int]
    L.nb_sim_let_set_import_stride.argtypes = [vp, C.c_uint32]
    L.nb_sim_let_set_owners.argtypes = [vp, P(C.c_ulonglong), C.c_int, C.c_float, C.c_uint32]
    L.nb_sim_let_set_arrivals.argtypes = [vp, C.c_uint32, P(C.c_uint32), C.c_int]
    L.nb_sim_sim_params.argtypes = [vp, P(nb_sim_params)]
    L.nb_sim_read_particles.argtypes = [vp, vp, sz]
    L.nb_sim_write_particles.argtypes = [vp, vp, sz]
    L.nb_sim_read_tree.argtypes = [vp, vp, sz, P(sz), P(C.c_float)]
    L.nb_sim_exchange_region.argtypes = [vp, P(vp), P(sz), P(sz), P(sz)]
    L.nb_sim_exchange_count.argtypes = [vp, P(C.c_int)]
    L.nb_sim_exchange_region_i.argtypes = [vp, C.c_int, P(vp), P(sz), P(sz), P(sz)]
    L.nb_sim_step_num.argtypes = [vp, P(u64)]
    L.nb_sim_encode_n_timed.argtypes = [vp, C.c_int, P(C.c_float), P(C.c_float)]
    L.nb_sim_set_tuning.argtypes = [vp, C.c_char_p, C.c_int]
    L.nb_sim_debug_buffer.argtypes = [vp, C.c_char_p, vp, sz, P(sz)]
    L.nb_naive_variant_count.restype = C.c_int
    L.nb_naive_variant_name.argtypes = [C.c_int]
    L.nb_naive_variant_name.restype = C.c_char_p
    L.nb_runner_create.argtypes = [P(vp), P(nb_sim_params), P(nb_add_params), vp, vp, C.c_int]
    L.nb_runner_create_multi.argtypes = [P(vp), P(nb_sim_params), P(nb_add_params), vp, vp, P(C.c_int), C.c_int]
    L.nb_runner_create_multi_let.argtypes = [P(vp), P(nb_sim_params), P(nb_add_params), vp, vp, P(C.c_int), C.c_int,
                                             C.c_int]
    L.nb_runner_step_num.argtypes = [vp, P(u64)]
    L.nb_runner_step.argtypes = [vp]
    L.nb_runner_step_n.argtypes = [vp, C.c_int]
    L.nb_runner_read_particles.argtypes = [vp, vp, sz]
    L.nb_runner_set_profiling.argtypes = [vp, C.c_int]
    L.nb_runner_rank_times.argtypes = [vp, P(C.c_float), P(C.c_float), C.c_int]
    L.nb_runner_sim_params.argtypes = [vp, P(nb_sim_params)]
    L.nb_runner_sim.argtypes = [vp]
    L.nb_runner_sim.restype = vp
    L.nb_runner_destroy.argtypes = [vp]
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != NB_OK:
        raise NBodyError(rc, lib().nb_last_error().decode(errors="replace"))
