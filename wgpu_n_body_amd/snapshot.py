"""Snapshots of a simulation: the reference's only on-disk-capable format is its POD pair
`SimParams` (16 B, src/sims/mod.rs:51-58) + `Particle[]` (40 B each, src/sims/mod.rs:9-16)
(SURVEY 8f F3).  File layout, little-endian:

    offset  0  magic   b"NBSNAP01"
            8  u64     step number
           16  SimParams {u32 particle_num; f32 g, e, dt}
           32  Particle[particle_num]   (10 x f32 each, field order pos/vel/acc/mass)

Restore with Simulator.write_particles(load_snapshot(path)[1]) (nb_sim_write_particles), or
construct a new simulator from the particles.
"""
from __future__ import annotations

import struct

import numpy as np

from . import PARTICLE_DTYPE, SimParams, as_particles

MAGIC = b"NBSNAP01"


def save_snapshot(path: str, sim_params: SimParams, particles, step_num: int = 0) -> None:
    p = as_particles(particles)
    if p.shape[0] != sim_params.particle_num:
        raise ValueError("particle count does not match sim_params.particle_num")
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", int(step_num)))
        f.write(struct.pack("<Ifff", int(sim_params.particle_num), float(sim_params.g),
                            float(sim_params.e), float(sim_params.dt)))
        f.write(p.tobytes())


def load_snapshot(path: str):
    """-> (SimParams, particles[PARTICLE_DTYPE], step_num)"""
    with open(path, "rb") as f:
        head = f.read(32)
        if len(head) != 32 or head[:8] != MAGIC:
            raise ValueError(f"{path}: not an n-body snapshot")
        (step_num,) = struct.unpack("<Q", head[8:16])
        n, g, e, dt = struct.unpack("<Ifff", head[16:32])
        body = f.read()
    if len(body) != n * PARTICLE_DTYPE.itemsize:
        raise ValueError(f"{path}: expected {n} particles, file holds {len(body) / 40:.1f}")
    return SimParams(n, g, e, dt), np.frombuffer(body, dtype=PARTICLE_DTYPE).copy(), int(step_num)
