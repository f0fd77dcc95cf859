"""Shared helpers for the test modules."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

G, E, DT = 0.000001, 0.0001, 0.016  # SimParams::default, src/sims/mod.rs:62-71


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def make_state(kind: str, n: int, seed: int, g: float = G) -> np.ndarray:
    """Seeded float32[n,10] initial state from the product's own (host-side) inits."""
    import wgpu_n_body_amd as nb
    sp = nb.SimParams(particle_num=n, g=g)
    fn = {"uniform": nb.inits.uniform_init, "disc": nb.inits.disc_init,
          "spherical": nb.inits.spherical_init}[kind]
    return nb.as_floats(fn(sp, seed=seed)).copy()
