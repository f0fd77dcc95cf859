"""Seeded inits (nb_init_*; reference src/inits.rs:6-83) -- CPU only.

The reference's generators use rand::thread_rng() and cannot be reproduced, so parity is on
the DISTRIBUTION; the product's own generator is specified bit-exactly (nb_inits.cpp header)
and restated here in numpy, independently, so the specification itself is pinned.
"""
import numpy as np
import pytest

from tests.helpers import bits

M64 = (1 << 64) - 1
F = np.float32


class Rng:
    """splitmix64 counter stream -> 24-bit -> [-1, 1] (both ends reachable)."""

    def __init__(self, seed):
        self.seed, self.k = seed & M64, 0

    def unif(self):
        self.k += 1
        z = (self.seed + self.k * 0x9E3779B97F4A7C15) & M64
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        z = z ^ (z >> 31)
        return F(np.float64(z >> 40) * (2.0 / 16777215.0) - 1.0)


def length3(x, y, z):
    return F(np.sqrt(F(F(x * x + y * y) + z * z)))


def ref_uniform(n, seed):
    r = Rng(seed)
    out = np.zeros((n, 10), F)
    for i in range(n):
        out[i, 0:3] = [r.unif(), r.unif(), r.unif()]
        out[i, 3:6] = [F(r.unif() * F(0.001)) for _ in range(3)]
        out[i, 9] = 1.0
    return out


def ref_spherical(n, seed):
    r = Rng(seed)
    out = np.zeros((n, 10), F)
    for i in range(n):
        x, y, z = r.unif(), r.unif(), r.unif()
        while length3(x, y, z) > F(1.0):
            x, y, z = r.unif(), r.unif(), r.unif()
        inv = F(F(1.0) / length3(x, y, z))
        out[i, 0:3] = [x, y, z]
        out[i, 3:6] = [F(F(c * inv) * F(0.4)) for c in (x, y, z)]
        out[i, 9] = F(r.unif() + F(2.0))
    return out


def ref_disc(n, seed, g):
    r = Rng(seed)
    out = np.zeros((n, 10), F)
    if n:
        out[0, 9] = 150000.0
    for i in range(1, n):
        x, y, z = r.unif(), r.unif(), F(0.0)
        ln = length3(x, y, z)
        while ln > F(1.0) or ln < F(0.25):
            x, y, z = r.unif(), r.unif(), F(r.unif() * F(0.1))
            ln = length3(x, y, z)
        x, y, z = F(x * ln), F(y * ln), F(z * ln)
        speed = F(np.sqrt(F(F(F(g) * F(1000.0)) / length3(x, y, z))))
        cx, cy, cz = y, F(-x), F(0.0)
        inv = F(F(1.0) / length3(cx, cy, cz))
        out[i, 0:3] = [x, y, z]
        out[i, 3:6] = [F(speed * F(cx * inv)), F(speed * F(cy * inv)), F(speed * F(cz * inv))]
        out[i, 9] = 1.0
    return out


@pytest.mark.parametrize("seed", [0, 1, 0xDEADBEEFCAFEF00D])
def test_uniform_matches_numpy_spec_bitwise(nb, seed):
    sp = nb.SimParams(particle_num=200)
    got = nb.as_floats(nb.inits.uniform_init(sp, seed=seed))
    assert np.array_equal(bits(got), bits(ref_uniform(200, seed)))


@pytest.mark.parametrize("seed", [0, 7])
def test_spherical_matches_numpy_spec_bitwise(nb, seed):
    sp = nb.SimParams(particle_num=150)
    got = nb.as_floats(nb.inits.spherical_init(sp, seed=seed))
    assert np.array_equal(bits(got), bits(ref_spherical(150, seed)))


@pytest.mark.parametrize("seed,g", [(0, 1e-6), (3, 1e-5)])
def test_disc_matches_numpy_spec_bitwise(nb, seed, g):
    sp = nb.SimParams(particle_num=150, g=g)
    got = nb.as_floats(nb.inits.disc_init(sp, seed=seed))
    assert np.array_equal(bits(got), bits(ref_disc(150, seed, np.float32(g))))


def test_uniform_distribution(nb):
    """inits.rs:6-27: pos ~ U[-1,1]^3, vel ~ U[-1,1]^3 * 0.001, acc 0, mass 1."""
    p = nb.as_floats(nb.inits.uniform_init(nb.SimParams(particle_num=200000), seed=2))
    assert p[:, 0:3].min() >= -1 and p[:, 0:3].max() <= 1
    assert np.abs(p[:, 0:3].mean(0)).max() < 0.01
    assert np.abs(p[:, 0:3].var(0) - 1 / 3).max() < 0.01
    assert np.abs(p[:, 3:6]).max() <= 0.001 and np.abs(p[:, 3:6].var(0) - 1e-6 / 3).max() < 1e-8
    assert not p[:, 6:9].any() and (p[:, 9] == 1).all()
    # octants equally populated
    occ = np.bincount((p[:, 0] > 0) + 2 * (p[:, 1] > 0) + 4 * (p[:, 2] > 0), minlength=8)
    assert np.abs(occ / len(p) - 0.125).max() < 0.005


def test_spherical_distribution(nb):
    """inits.rs:56-83: uniform in the unit ball, v = 0.4 * r_hat, m ~ U[1,3]."""
    p = nb.as_floats(nb.inits.spherical_init(nb.SimParams(particle_num=100000), seed=3))
    r = np.linalg.norm(p[:, 0:3].astype(np.float64), axis=1)
    assert r.max() <= 1.0 + 1e-6
    assert abs((r < 0.5).mean() - 0.125) < 0.005           # P(r < a) = a^3
    speed = np.linalg.norm(p[:, 3:6].astype(np.float64), axis=1)
    assert np.allclose(speed, 0.4, rtol=1e-5)
    cos = (p[:, 0:3] * p[:, 3:6]).sum(1) / (r * speed)
    assert np.allclose(cos, 1.0, atol=1e-5)                # outward
    assert p[:, 9].min() >= 1 and p[:, 9].max() <= 3 and abs(p[:, 9].mean() - 2) < 0.01
    assert not p[:, 6:9].any()


def test_disc_distribution(nb):
    """inits.rs:29-54: heavy centre + thin disc on circular-ish orbits about +z."""
    g = 1e-5
    p = nb.as_floats(nb.inits.disc_init(nb.SimParams(particle_num=50000, g=g), seed=4))
    assert p[0, 9] == 150000 and not p[0, 0:9].any()
    q = p[1:].astype(np.float64)
    r = np.linalg.norm(q[:, 0:3], axis=1)
    assert r.min() >= 0.25 ** 2 - 1e-6 and r.max() <= 1 + 1e-6   # len in [.25,1], then squared
    assert np.abs(q[:, 2]).max() <= 0.1                     # thin
    speed = np.linalg.norm(q[:, 3:6], axis=1)
    assert np.allclose(speed, np.sqrt(g * 1000 / r), rtol=1e-5)
    assert np.abs((q[:, 0:3] * q[:, 3:6]).sum(1)).max() < 1e-6  # v is perpendicular to pos (xy)
    assert (q[:, 5] == 0).all() and (q[:, 9] == 1).all()
    lz = q[:, 0] * q[:, 4] - q[:, 1] * q[:, 3]
    assert (lz < 0).all()                                    # pos x Z: clockwise about +z


def test_seed_changes_stream_and_zero_particles(nb):
    sp = nb.SimParams(particle_num=10)
    a = nb.as_floats(nb.inits.uniform_init(sp, seed=1))
    b = nb.as_floats(nb.inits.uniform_init(sp, seed=2))
    assert not np.array_equal(a, b)
    assert np.array_equal(a, nb.as_floats(nb.inits.uniform_init(sp, seed=1)))
    for fn in (nb.inits.uniform_init, nb.inits.disc_init, nb.inits.spherical_init):
        assert fn(nb.SimParams(particle_num=0)).shape == (0,)
