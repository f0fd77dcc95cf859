"""Generates tests/golden/*.npz -- the committed parity fixtures.

The reference (arpan-dhatt/wgpu-n-body) holds no golden vectors, no tests and no CPU force
path, and cannot be built or run in this image (Rust + wgpu), so these fixtures are produced
by this repo's CPU oracle (oracle/nbody_oracle*.c, cross-checked bit-for-bit against the
independent numpy restatement oracle/oracle_np.py).  PARITY UNPINNED by the reference.

Each fixture: seeded initial particles (the product's nb_init_* generators), the literal-fp32
oracle state and the fp64 oracle state after 1 and 10 steps.

    python tests/golden/make_golden.py        # rewrites every fixture
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import wgpu_n_body_amd as nb  # noqa: E402
from oracle import oracle as O  # noqa: E402

G, E, DT = 0.000001, 0.0001, 0.016  # SimParams::default, src/sims/mod.rs:62-71

NAIVE_CASES = [  # (name, init, n, seed, g, dt)
    ("naive_uniform_n2", "uniform", 2, 11, G, DT),
    ("naive_uniform_n3", "uniform", 3, 12, G, DT),
    ("naive_uniform_n64", "uniform", 64, 13, G, DT),
    ("naive_uniform_n1000", "uniform", 1000, 14, G, DT),
    ("naive_spherical_n64", "spherical", 64, 15, G, DT),
    ("naive_spherical_n1024", "spherical", 1024, 1, G, DT),   # BASELINE config 1
    ("naive_disc_n64", "disc", 64, 16, 0.00001, 0.0016),      # visualize.rs:26-31 parameters
    ("naive_disc_n1024", "disc", 1024, 17, 0.00001, 0.0016),
]
TREE_CASES = [  # (name, init, n, seed, theta)
    ("tree_uniform_n64", "uniform", 64, 21, 0.5),
    ("tree_uniform_n1000", "uniform", 1000, 22, 0.5),
    ("tree_uniform_n1024_t075", "uniform", 1024, 23, 0.75),
    ("tree_spherical_n1024", "spherical", 1024, 24, 0.5),
    ("tree_disc_n1024", "disc", 1024, 25, 0.75),
]


def init_state(kind, n, seed, g):
    sp = nb.SimParams(particle_num=n, g=g)
    fn = {"uniform": nb.inits.uniform_init, "disc": nb.inits.disc_init,
          "spherical": nb.inits.spherical_init}[kind]
    return nb.as_floats(fn(sp, seed=seed)).copy()


def main():
    for name, kind, n, seed, g, dt in NAIVE_CASES:
        s0 = init_state(kind, n, seed, g)
        out = dict(init=s0, params=np.array([g, E, dt], dtype=np.float32), n=n, seed=seed)
        s32, s64 = s0.copy(), s0.astype(np.float64)
        for step in range(1, 11):
            s32 = O.naive_step_f32(s32, g, E, dt)
            s64 = O.naive_step_f64(s64, g, E, dt)
            if step in (1, 10):
                out[f"f32_step{step}"] = s32.copy()
                out[f"f64_step{step}"] = s64.copy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name)
    for name, kind, n, seed, theta in TREE_CASES:
        g, dt = (0.00001, 0.0016) if kind == "disc" else (G, DT)
        s0 = init_state(kind, n, seed, g)
        r = O.tree_step_f32(s0, g, E, dt, theta, flags=O.INTENDED)
        lit = O.tree_step_f32(s0, g, E, dt, theta, flags=O.LITERAL)
        # 3 further intended-semantics steps chained (each step re-sorts, as the reference does)
        s = r["dst"]
        for _ in range(3):
            s = O.tree_step_f32(s, g, E, dt, theta, flags=O.INTENDED)["dst"]
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), init=s0,
            params=np.array([g, E, dt, theta], dtype=np.float32), n=n, seed=seed,
            tree=r["tree"], root_width=np.float32(r["root_width"]), order=r["order"],
            sorted_src=r["sorted_src"], dst_intended=r["dst"], dst_literal=lit["dst"],
            dst_intended_step4=s,
            stats_intended=np.array(list(r["stats"].values()), dtype=np.uint64),
            stats_literal=np.array(list(lit["stats"].values()), dtype=np.uint64))
        print("wrote", name)


if __name__ == "__main__":
    main()
