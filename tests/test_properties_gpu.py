"""Size-independent properties of the step at BASELINE.json's FULL single-GPU sizes (`-m gpu`), where the
oracle can only afford windows of bodies: exact scaling laws of the force (every power-of-two
scaling of the masses or of g commutes with every fp32 rounding of the kernels, so the new
accelerations must scale bit for bit), independence of the Barnes-Hut step from the order the
bodies are handed over in (the tree is keyed on positions), and the momentum balance.
configs[1]: 65,536 bodies all-pairs; configs[2]: 1,048,576 bodies Barnes-Hut theta 0.5."""
import numpy as np
import pytest

from tests.helpers import DT, E, G, bits

pytestmark = pytest.mark.gpu


def step_once(nb, cls, state, g=G, theta=None):
    sp = nb.SimParams(particle_num=len(state), g=g, e=E, dt=DT)
    add = nb.AddParams.TreeSimParams(theta) if theta is not None else None
    sim = cls.from_particles(sp, add, state)
    sim.encode()
    sim.cleanup()
    sim.wait()
    out = nb.as_floats(sim.dest_particle_slice()).copy()
    sim.destroy()
    return out


@pytest.mark.parametrize("which", ["naive-65536", "tree-1048576"])
def test_accelerations_scale_exactly_with_the_masses_and_with_g(gpu, which):
    nb = gpu
    kind, n = which.split("-")
    n = int(n)
    cls, theta = (nb.NaiveSim, None) if kind == "naive" else (nb.TreeSim, 0.5)
    sp = nb.SimParams(particle_num=n)
    base = nb.as_floats(nb.inits.spherical_init(sp, seed=12)).copy()
    base[:, 9] = 1.0 + (np.arange(n) % 7).astype(np.float32) * np.float32(0.25)     # a few different masses
    a = step_once(nb, cls, base, theta=theta)
    heavy = base.copy()
    heavy[:, 9] *= np.float32(2.0)
    b = step_once(nb, cls, heavy, theta=theta)
    c = step_once(nb, cls, base, g=G * 4.0, theta=theta)
    assert np.isfinite(a).all()
    assert np.array_equal(bits(a[:, 0:3]), bits(b[:, 0:3])) and np.array_equal(bits(a[:, 0:3]), bits(c[:, 0:3]))
    assert np.array_equal(bits(b[:, 6:9]), bits(a[:, 6:9] * np.float32(2.0)))       # masses x 2 -> forces x 2, every bit
    assert np.array_equal(bits(c[:, 6:9]), bits(a[:, 6:9] * np.float32(4.0)))       # g x 4 -> forces x 4, every bit
    assert np.array_equal(b[:, 9], a[:, 9] * np.float32(2.0))                       # (same body order)


def test_momentum_balance_of_the_full_size_steps(gpu):
    """sum_i m_i a_i = 0 for exact pair forces.  All-pairs evaluates every pair twice with different
    roundings (1e-6 of sum |m a|); Barnes-Hut replaces far groups by their centre of gravity, which
    breaks the symmetry by the method's own error (theta 0.5: below 1e-3)."""
    nb = gpu
    for cls, n, theta, tol in ((nb.NaiveSim, 65536, None, 2e-6), (nb.TreeSim, 1 << 20, 0.5, 1e-3)):
        sp = nb.SimParams(particle_num=n)
        s = nb.as_floats(nb.inits.uniform_init(sp, seed=13)).copy()
        s[:, 9] = 0.5 + (np.arange(n) % 5).astype(np.float32) * np.float32(0.5)
        out = step_once(nb, cls, s, theta=theta).astype(np.float64)
        ma = out[:, 9:10] * out[:, 6:9]
        assert np.linalg.norm(ma.sum(axis=0)) <= tol * np.abs(ma).sum(), cls.__name__


def test_barnes_hut_step_does_not_depend_on_the_input_order(gpu):
    """1,048,576 bodies handed over in a random order: the keys, hence the tree order, the tree and every
    group of the walk are the same, so every body ends with the same bits (bodies matched by mass tag)."""
    nb = gpu
    n = 1 << 20
    sp = nb.SimParams(particle_num=n)
    s = nb.as_floats(nb.inits.uniform_init(sp, seed=14)).copy()
    s[:, 9] = 1.0 + np.arange(n, dtype=np.float32) / np.float32(2 * n)             # distinct masses = tags
    perm = np.random.default_rng(3).permutation(n)
    a = step_once(nb, nb.TreeSim, s, theta=0.5)
    b = step_once(nb, nb.TreeSim, s[perm], theta=0.5)
    assert np.array_equal(bits(a), bits(b))            # both leave the bodies in tree order: the same array
