"""Builds libnbody_hip.so (the C-ABI library of include/nbody.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
.so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJ = os.path.join(PKG, "_build")
LIB = os.path.join(PKG, "libnbody_hip.so")
HEADLESS = os.path.join(PKG, "headless")
ARCH = "gfx950"

# (source, extra flags)
SOURCES = [
    ("nb_naive.hip", []),
    ("nb_tree.hip", []),
    ("nb_abi.cpp", []),
    ("nb_group.cpp", []),
    # the inits are specified bit-exactly (DESIGN.md "RNG"): no FMA contraction
    ("nb_inits.cpp", ["-ffp-contract=off"]),
]
HEADERS = ["nb_common.hpp", "nb_sim.hpp", "nb_group.hpp", os.path.join(INCLUDE, "nbody.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def source_hash() -> str:
    """sha256 over every source the library is built from (sorted by name).  Compiled into the
    library (nb_version() ends with "src:<12 hex digits>") so that a stale .so shipped beside newer
    sources is caught by tests/test_abi.py and by __graft_entry__.build()."""
    import hashlib
    h = hashlib.sha256()
    files = sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC)
                    if f.endswith((".hip", ".cpp", ".hpp"))] + [os.path.join(INCLUDE, "nbody.h")])
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    hdrs.append(os.path.abspath(__file__))
    common = ["-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}", f"-I{CSRC}", "-Wall",
              "-Wno-unused-function"]
    objs = []
    # the source hash goes into nb_abi.o; a changed hash (any source edited) recompiles it
    digest = source_hash()
    stamp = os.path.join(OBJ, "source_hash.txt")
    stamp_ok = os.path.exists(stamp) and open(stamp).read().strip() == digest
    if os.environ.get("NB_ALL_VARIANTS") == "1":
        common.append("-DNB_ALL_VARIANTS")
    for src, extra in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if src == "nb_abi.cpp":
            extra = extra + [f'-DNB_SOURCE_HASH="{digest}"']
        if force or _newer(obj, [path] + hdrs) or (src == "nb_abi.cpp" and not stamp_ok):
            cmd = [hipcc, f"--offload-arch={ARCH}"] + common + extra + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
    if force or _newer(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(digest + "\n")
    # the C++ host mirror (simulator.hpp) + the headless CLI, plain g++ against the C ABI
    cli_src = os.path.join(CSRC, "headless.cpp")
    if force or _newer(HEADLESS, [cli_src, os.path.join(CSRC, "simulator.hpp"), LIB]):
        cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", f"-I{INCLUDE}", f"-I{CSRC}",
               cli_src, "-o", HEADLESS, f"-L{PKG}", "-lnbody_hip", f"-Wl,-rpath,{PKG}",
               "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
