"""north_star's MFMA clause, the part that can be settled without a kernel: recasting the pairwise
r^2 as a dense fp32 contraction means r^2 = |x_i|^2 + |x_j|^2 - 2 x_i . x_j (the only form a
matrix instruction computes), every product and sum rounded to binary32 as v_mfma_f32_16x16x4f32
rounds them.  This script evaluates the all-pairs acceleration of BASELINE configs[1] (65,536
bodies, uniform_init) for a sample of bodies three ways -- binary64 (reference), binary32 with
r^2 from coordinate differences (what nb_naive.hip does), binary32 with r^2 from the expansion --
and prints the error of the two fp32 forms against binary64.  numpy on the CPU; needs the built
library only for the seeded init.      python tools/mfma_r2_error.py [n] [sample]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_n_body_amd as nb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sample = int(sys.argv[2]) if len(sys.argv) > 2 else 512
sp = nb.SimParams(particle_num=n)
f = nb.as_floats(nb.inits.uniform_init(sp, seed=0))
pos32, m32 = f[:, 0:3].copy(), f[:, 9].copy()
e32 = np.float32(sp.e)
idx = np.linspace(0, n - 1, sample).astype(np.int64)


def acc(xi, i, mode):
    if mode == "f64":
        d = pos32.astype(np.float64) - xi.astype(np.float64)
        r2 = (d * d).sum(axis=1)
        r2[i] = 1.0
        w = m32.astype(np.float64) / (r2 * r2 + float(e32) * np.sqrt(r2))
        w[i] = 0.0
        return (w[:, None] * d).sum(axis=0)
    d = pos32 - xi                                   # binary32 throughout
    if mode == "diff":
        r2 = d[:, 2] * d[:, 2] + (d[:, 1] * d[:, 1] + d[:, 0] * d[:, 0])
    else:                                            # the contraction: K = 4 column (x, y, z, |x_j|^2) . (-2 x_i, 1)
        nj = pos32[:, 0] * pos32[:, 0] + pos32[:, 1] * pos32[:, 1] + pos32[:, 2] * pos32[:, 2]
        ni = np.float32(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2])
        a = np.float32(-2.0) * xi
        c = ((pos32[:, 0] * a[0] + pos32[:, 1] * a[1]) + pos32[:, 2] * a[2]) + nj   # the MFMA accumulation chain
        r2 = np.maximum(c + ni, np.float32(0.0))
    r2 = r2.astype(np.float32)
    r2[i] = 1.0
    w = m32 / (r2 * r2 + e32 * np.sqrt(r2))
    w[i] = 0.0
    w = np.where(np.isfinite(w), w, np.float32(0.0)).astype(np.float32)
    return (w[:, None] * d).astype(np.float32).sum(axis=0, dtype=np.float32)


err = {"diff": [], "mfma": []}
r2rel = []
for i in idx:
    ref = acc(pos32[i], i, "f64")
    for mode in err:
        got = acc(pos32[i], i, mode).astype(np.float64)
        err[mode].append(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    # the nearest neighbour's r^2 both ways
    d = pos32.astype(np.float64) - pos32[i].astype(np.float64)
    r2 = (d * d).sum(axis=1)
    r2[i] = np.inf
    j = int(np.argmin(r2))
    nj = np.float32(pos32[j] @ pos32[j])
    ni = np.float32(pos32[i] @ pos32[i])
    c = np.float32(np.float32(np.float32(-2.0) * np.float32(pos32[i] @ pos32[j])) + nj) + ni
    r2rel.append(abs(float(c) - r2[j]) / r2[j])
print(f"n = {n} bodies (uniform_init seed 0), {sample} sampled bodies, softening e = {float(e32):g}")
for mode, label in (("diff", "fp32, r^2 from coordinate differences (nb_naive.hip)"),
                    ("mfma", "fp32, r^2 = |xi|^2 + |xj|^2 - 2 xi.xj (the MFMA contraction)")):
    v = np.array(err[mode])
    print(f"  {label:62s} acceleration error vs fp64: median {np.median(v):.2e}  p99 {np.percentile(v, 99):.2e}  max {v.max():.2e}")
v = np.array(r2rel)
print(f"  relative error of the contraction's r^2 for the nearest neighbour:            median {np.median(v):.2e}  p99 {np.percentile(v, 99):.2e}  max {v.max():.2e}")
print("  (tests/test_naive_gpu.py accepts 2e-5 of the largest acceleration)")
