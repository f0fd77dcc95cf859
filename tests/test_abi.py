"""The C-ABI boundary (include/nbody.h <-> libnbody_hip.so) -- CPU only, no compute calls.

Checks that the library builds and loads, exports every function the header declares, that
the POD layouts match the reference's #[repr(C)] structs byte for byte, the shard arithmetic,
and the error behaviour without a device (the product must fail loudly: no CPU fallback).
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from tests.helpers import ROOT

HEADER = os.path.join(ROOT, "include", "nbody.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(nb_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if n != "nb_init_fn"))


def test_library_exports_every_declared_symbol(nb):
    from wgpu_n_body_amd import _lib
    L = _lib.lib()
    decl = declared_functions()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(L, name), f"{name} declared in nbody.h but not exported"
    assert sorted(_lib.ABI_SYMBOLS) == decl, "python binding list out of sync with the header"


def test_exports_are_plain_c_symbols(nb):
    from wgpu_n_body_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True,
                         text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert set(declared_functions()) <= exported
    # nothing C++-mangled leaks out as part of the intended surface
    assert all(not s.startswith("_Z") or "nb" in s for s in exported)


def test_pod_layouts_match_the_reference(nb):
    from wgpu_n_body_amd import _lib
    # Particle: src/sims/mod.rs:9-16 -- 10 x f32 = 40 B, order pos/vel/acc/mass
    d = nb.PARTICLE_DTYPE
    assert d.itemsize == 40
    assert [d.fields[k][1] for k in ("position", "velocity", "acceleration", "mass")] == [0, 12, 24, 36]
    # SimParams: src/sims/mod.rs:51-58 -- u32 + 3 x f32 = 16 B
    assert C.sizeof(_lib.nb_sim_params) == 16
    assert [getattr(_lib.nb_sim_params, f).offset for f in ("particle_num", "g", "e", "dt")] == [0, 4, 8, 12]
    # Octant: src/sims/tree.rs:605-622 -- 52 B
    o = nb.OCTANT_DTYPE
    assert o.itemsize == 52
    assert [o.fields[k][1] for k in ("cog", "mass", "bodies", "children")] == [0, 12, 16, 20]
    assert C.sizeof(_lib.nb_add_params) == 8


def test_defaults_match_the_reference(nb):
    sp = nb.SimParams()  # SimParams::default, src/sims/mod.rs:62-71
    assert (sp.particle_num, sp.g, sp.e, sp.dt) == (10000, 0.000001, 0.0001, 0.016)
    assert nb.PARTICLES_PER_GROUP == 64  # src/sims/mod.rs:7
    hdr = open(HEADER).read()
    assert "#define NB_DEFAULT_THETA 0.75f" in hdr  # src/sims/tree.rs:42-51


def test_version_and_variants(nb):
    assert nb.version().startswith("nbody_hip") and "gfx950" in nb.version()
    # the library carries the hash of the sources it was built from: a stale .so shipped beside
    # newer sources (the .so is git-ignored and travels with the snapshot) fails here
    from wgpu_n_body_amd.build import source_hash
    assert nb.version().endswith("src:" + source_hash()), (nb.version(), source_hash())
    v = nb.naive_variants()
    assert len(v) >= 4 and len(set(v)) == len(v)


@pytest.mark.parametrize("n,world", [(65536, 1), (65536, 8), (262144, 8), (1000, 3), (1, 1),
                                     (0, 2), (255, 2), (257, 2), (4194304, 8)])
def test_shard_arithmetic(nb, n, world):
    per = nb.shard_bodies_per_rank(n, world)
    pad = nb.shard_padded_bodies(n, world)
    assert per % 256 == 0 and per > 0
    assert pad == per * world and pad >= n
    assert per * (world - 1) <= max(n, 1) + 256 * world   # not grossly over-padded
    # ranges are disjoint, ordered and cover [0, n)
    covered = 0
    for r in range(world):
        lo, hi = min(n, r * per), min(n, (r + 1) * per)
        assert lo == covered or lo == n
        covered = hi
    assert covered == n


def test_no_device_fails_loudly(nb):
    """Without a HIP device nb_sim_create must return NB_ERR_NO_DEVICE -- never fall back."""
    if nb.device_count() > 0:
        pytest.skip("this machine has a GPU; covered by the -m gpu tests")
    sp = nb.SimParams(particle_num=8)
    with pytest.raises(nb.NBodyError) as ei:
        nb.NaiveSim.new(sp, nb.AddParams.NaiveSimParams(), nb.inits.uniform_init)
    assert ei.value.code == 2 and "no CPU fallback" in str(ei.value)
    with pytest.raises(nb.NBodyError):
        nb.OfflineHeadless(nb.NaiveSim, sp, None, nb.inits.uniform_init)


def test_argument_errors_are_status_codes_not_crashes(nb):
    from wgpu_n_body_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    assert L.nb_sim_create(C.byref(h), None, None, None, None, None) == _lib.NB_ERR_INVALID
    assert b"non-null" in L.nb_last_error() or b"null" in L.nb_last_error()
    assert L.nb_sim_encode(None) == _lib.NB_ERR_INVALID
    assert L.nb_sim_wait(None) == _lib.NB_ERR_INVALID
    assert L.nb_runner_step(None) == _lib.NB_ERR_INVALID
    assert L.nb_sim_destroy(None) == _lib.NB_OK and L.nb_runner_destroy(None) == _lib.NB_OK
    sp = _lib.nb_sim_params(4, 1e-6, 1e-4, 0.016)
    bad = _lib.nb_add_params(7, 0.0)
    buf = np.zeros(4, dtype=nb.PARTICLE_DTYPE)
    rc = L.nb_sim_create_from_particles(C.byref(h), C.byref(sp), C.byref(bad), None,
                                        buf.ctypes.data, 4)
    assert rc == _lib.NB_ERR_INVALID and b"kind" in L.nb_last_error()


def test_init_fn_exceptions_do_not_cross_the_abi(nb):
    def boom(_sp):
        raise KeyError("init exploded")

    with pytest.raises(KeyError):
        nb.NaiveSim.new(nb.SimParams(particle_num=4), None, boom)

    def short(_sp):
        return np.zeros(3, dtype=nb.PARTICLE_DTYPE)

    with pytest.raises(ValueError):
        nb.NaiveSim.new(nb.SimParams(particle_num=4), None, short)


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under wgpu_n_body_amd/ may reference it."""
    pkg = os.path.join(ROOT, "wgpu_n_body_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        if "_build" in dirpath or "__pycache__" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "nbody_oracle" not in text and "from oracle" not in text \
                    and "import oracle" not in text and "libnbody_oracle" not in text, f
