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
    # the inits are specified bit-exactly (DESIGN.md "RNG"): no FMA contraction
    ("nb_inits.cpp", ["-ffp-contract=off"]),
]
HEADERS = ["nb_common.hpp", "nb_sim.hpp", os.path.join(INCLUDE, "nbody.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


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
    for src, extra in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _newer(obj, [path] + hdrs):
            cmd = [hipcc, f"--offload-arch={ARCH}"] + common + extra + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
    if force or _newer(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
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
