"""ctypes loader for the CPU oracle (oracle/libnbody_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under wgpu_n_body_amd/ imports this.
PARITY UNPINNED -- see the header of nbody_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnbody_oracle.so")

SELF_BY_IDENTITY = 1
LEAF_IS_BODY = 2
CHECKED_STACK = 4
INTENDED = 7  # what the HIP kernels implement
LITERAL = 0   # tree.wgsl as written, defects included

OCTANT_DTYPE = np.dtype(
    [("cog", "<f4", (3,)), ("mass", "<f4"), ("bodies", "<u4"), ("children", "<u4", (8,))]
)
assert OCTANT_DTYPE.itemsize == 52


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")] + [
        os.path.join(_HERE, "Makefile")
    ]
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs
    )
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        f32p, f64p, u32p, u64p = (C.POINTER(C.c_float), C.POINTER(C.c_double),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint64))
        L.nbo_isa.restype = C.c_char_p
        L.nbo_max_threads.restype = C.c_int
        L.nbo_set_threads.argtypes = [C.c_int]
        L.nbo_naive_step_f32.argtypes = [f32p, f32p, C.c_uint32, C.c_float, C.c_float, C.c_float,
                                         C.c_uint32, C.c_uint32]
        L.nbo_naive_step_f32.restype = None
        L.nbo_naive_step_f64.argtypes = [f64p, f64p, C.c_uint32, C.c_double, C.c_double, C.c_double,
                                         C.c_uint32, C.c_uint32]
        L.nbo_naive_step_f64.restype = None
        L.nbo_tree_bound.argtypes = [f32p, C.c_uint32]
        L.nbo_tree_bound.restype = C.c_float
        L.nbo_tree_build.argtypes = [f32p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, f32p]
        L.nbo_tree_build.restype = C.c_int64
        L.nbo_tree_dfs_order.argtypes = [C.c_void_p, C.c_uint32, u32p]
        L.nbo_tree_dfs_order.restype = None
        L.nbo_tree_step_f32.argtypes = [f32p, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float,
                                        C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32,
                                        C.POINTER(C.c_int64), f32p, f32p, u32p, f32p, u64p]
        L.nbo_tree_step_f32.restype = C.c_int
        L.nbo_tree_walk_indices.argtypes = [f32p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_float, u32p,
                                            C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32,
                                            u32p, C.c_uint32, f32p, u64p]
        L.nbo_tree_walk_indices.restype = C.c_int
        _lib = L
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _as_state(particles) -> np.ndarray:
    """Accept an (n,10) float array or a structured nb_particle array; return (n,10) float32."""
    a = np.asarray(particles)
    if a.dtype.fields is not None:
        a = a.view(np.float32).reshape(-1, 10)
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 10
    return a


def naive_step_f32(state, g, e, dt, i_lo=0, i_hi=None) -> np.ndarray:
    """One literal-fp32 all-pairs step.  Rows outside [i_lo,i_hi) of the result are zero."""
    src = _as_state(state)
    n = src.shape[0]
    i_hi = n if i_hi is None else i_hi
    dst = np.zeros_like(src)
    lib().nbo_naive_step_f32(src.ctypes.data_as(C.POINTER(C.c_float)),
                             dst.ctypes.data_as(C.POINTER(C.c_float)), n,
                             np.float32(g), np.float32(e), np.float32(dt), i_lo, i_hi)
    return dst


def naive_step_f64(state, g, e, dt, i_lo=0, i_hi=None) -> np.ndarray:
    """One binary64 step on a binary64 state; g,e,dt are widened from their fp32 values."""
    src = np.ascontiguousarray(state, dtype=np.float64)
    n = src.shape[0]
    i_hi = n if i_hi is None else i_hi
    dst = np.zeros_like(src)
    g, e, dt = (float(np.float32(x)) for x in (g, e, dt))
    lib().nbo_naive_step_f64(src.ctypes.data_as(C.POINTER(C.c_double)),
                             dst.ctypes.data_as(C.POINTER(C.c_double)), n, g, e, dt, i_lo, i_hi)
    return dst


def naive_run_f32(state, g, e, dt, steps) -> np.ndarray:
    s = _as_state(state).copy()
    for _ in range(steps):
        s = naive_step_f32(s, g, e, dt)
    return s


def naive_run_f64(state, g, e, dt, steps) -> np.ndarray:
    s = np.asarray(_as_state(state), dtype=np.float64)
    for _ in range(steps):
        s = naive_step_f64(s, g, e, dt)
    return s


def tree_build(state, max_depth=64):
    """Literal BFS octree of src/sims/tree.rs:417-546 -> (octants[n_nodes], root_width)."""
    src = _as_state(state)
    n = src.shape[0]
    cap = max(4 * n, 8)
    tree = np.zeros(cap, dtype=OCTANT_DTYPE)
    rw = C.c_float(0)
    nodes = lib().nbo_tree_build(src.ctypes.data_as(C.POINTER(C.c_float)), n, tree.ctypes.data,
                                 cap, max_depth, C.byref(rw))
    if nodes < 0:
        raise RuntimeError("oracle tree build failed (coincident bodies / cap exceeded)")
    return tree[:nodes].copy(), float(rw.value)


def tree_dfs_order(tree, n) -> np.ndarray:
    order = np.zeros(n, dtype=np.uint32)
    t = np.ascontiguousarray(tree)
    lib().nbo_tree_dfs_order(t.ctypes.data, n, order.ctypes.data_as(C.POINTER(C.c_uint32)))
    return order


def tree_step_f32(state, g, e, dt, theta, flags=INTENDED, max_depth=64):
    """One Barnes-Hut step -> dict(dst, sorted_src, order, tree, root_width, stats)."""
    src = _as_state(state)
    n = src.shape[0]
    cap = max(4 * n, 8)
    tree = np.zeros(cap, dtype=OCTANT_DTYPE)
    nodes = C.c_int64(0)
    rw = C.c_float(0)
    sorted_src = np.zeros_like(src)
    dst = np.zeros_like(src)
    order = np.zeros(n, dtype=np.uint32)
    stats = np.zeros(5, dtype=np.uint64)
    f32p = C.POINTER(C.c_float)
    rc = lib().nbo_tree_step_f32(src.ctypes.data_as(f32p), n, np.float32(g), np.float32(e),
                                 np.float32(dt), np.float32(theta), flags, tree.ctypes.data, cap,
                                 max_depth, C.byref(nodes), C.byref(rw),
                                 sorted_src.ctypes.data_as(f32p),
                                 order.ctypes.data_as(C.POINTER(C.c_uint32)),
                                 dst.ctypes.data_as(f32p),
                                 stats.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc != 0:
        raise RuntimeError("oracle tree step failed (coincident bodies / cap exceeded)")
    return dict(dst=dst, sorted_src=sorted_src, order=order, tree=tree[: nodes.value].copy(),
                root_width=float(rw.value),
                stats=dict(visits=int(stats[0]), accepted=int(stats[1]), high_water=int(stats[2]),
                           overflowed=int(stats[3]), bad_index=int(stats[4])))


def tree_walk_indices(sorted_src, tree, root_width, g, e, dt, theta, indices, order=None, flags=INTENDED):
    """Walk + integrate the sorted bodies `indices` of an already built tree (tree.wgsl:41-111 per
    body) -> (rows[len(indices),10], stats).  order[k] = source index of sorted body k (the leaves
    name their body by source index, tree.rs:532); None: sorted_src is in source order."""
    src = _as_state(sorted_src)
    n = src.shape[0]
    t = np.ascontiguousarray(tree)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    out = np.zeros((len(idx), 10), dtype=np.float32)
    stats = np.zeros(5, dtype=np.uint64)
    f32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    od = None if order is None else np.ascontiguousarray(order, dtype=np.uint32)
    lib().nbo_tree_walk_indices(src.ctypes.data_as(f32p), n, t.ctypes.data, len(t), np.float32(root_width),
                                od.ctypes.data_as(u32p) if od is not None else None, np.float32(g),
                                np.float32(e), np.float32(dt), np.float32(theta), flags,
                                idx.ctypes.data_as(u32p), len(idx), out.ctypes.data_as(f32p),
                                stats.ctypes.data_as(C.POINTER(C.c_uint64)))
    return out, dict(visits=int(stats[0]), accepted=int(stats[1]), high_water=int(stats[2]),
                     overflowed=int(stats[3]), bad_index=int(stats[4]))


def tree_walk_window(sorted_src, tree, root_width, g, e, dt, theta, lo, hi, order=None, flags=INTENDED):
    return tree_walk_indices(sorted_src, tree, root_width, g, e, dt, theta, np.arange(lo, hi), order, flags)


def isa() -> str:
    return lib().nbo_isa().decode()


def max_threads() -> int:
    return lib().nbo_max_threads()


def set_threads(n: int) -> None:
    lib().nbo_set_threads(n)
