"""BASELINE.json's multi-GPU configurations at their FULL sizes, on this box's one GPU -- `-m gpu`.

The 8-GPU runs themselves are the driver's; here the same code paths run with all ranks on one
device (the one-process runner of the C ABI with a device id repeated; the LET protocol with the
exchanges done by device copies), which checks results, not links:
  configs[3]  262,144 bodies, all-pairs, 8 ranks   -> oracle on windows of bodies
  configs[4]  4,194,304 bodies, Barnes-Hut theta 0.5, 8 Morton domains + LET exchange
              -> every body kept, no status flag, force error of a sampled window against exact
                 all-pairs no worse than the single octree's
  and a TreeSim twice configs[4]'s size (the reference accepts up to 26.8 M particles,
  src/runners/mod.rs:18; round 1 capped the builder at 4,194,304)."""
import numpy as np
import pytest

from tests.helpers import DT, E, G, bits
from tests.test_let_gpu import LetGroup, by_tag, tagged

pytestmark = pytest.mark.gpu


def test_config3_262144_bodies_all_pairs_on_8_ranks(gpu, oracle):
    nb = gpu
    n, world = 262144, 8
    sp = nb.SimParams(particle_num=n)
    init = nb.inits.uniform_init(sp, seed=4)
    runner = nb.OfflineHeadless(nb.NaiveSim, sp, None, lambda _p: init, device_ids=[0] * world)
    runner.step()
    got = nb.as_floats(runner.read_particles())
    runner.destroy()
    assert np.isfinite(got).all()
    src = nb.as_floats(init)
    for lo in (0, 32768 - 64, n // 2, n - 128):        # inside a rank, across a rank border, the tail
        ref = oracle.naive_step_f32(src, G, E, DT, lo, lo + 128)[lo:lo + 128]
        win = got[lo:lo + 128]
        assert np.array_equal(bits(win[:, 0:3]), bits(ref[:, 0:3]))
        assert np.abs(win[:, 6:9] - ref[:, 6:9]).max() <= 2e-5 * np.abs(ref[:, 6:9]).max()
        assert np.array_equal(win[:, 9], ref[:, 9])


def test_config4_4194304_bodies_let_on_8_domains(gpu, oracle):
    nb = gpu
    n, world, theta = 4194304, 8, 0.5
    sp, p = tagged(nb, n, 5)
    grp = LetGroup(nb, sp, p, world, theta)
    grp.step()
    for s in grp.sims:
        s.wait()                                      # raises on any device status flag
        assert not s.debug_buffer("status", np.uint32).any()
    let = nb.as_floats(grp.particles())
    counts = grp.counts
    grp.destroy()
    # the same step through the C ABI alone (nb_runner_create_multi_let: rank threads inside the library,
    # records stored into the peers' import areas, counts consumed on the device): the same bits
    native = nb.OfflineHeadless(nb.TreeSim, sp, nb.AddParams.TreeSimParams(theta), lambda _p: p,
                                device_ids=[0] * world, let_migrate_every=4)
    native.step()
    assert np.array_equal(bits(nb.as_floats(native.read_particles())), bits(let))
    native.destroy()
    assert len(let) == n and np.isfinite(let).all()
    assert np.array_equal(np.sort(let[:, 9]), nb.as_floats(p)[:, 9])        # every body exactly once
    off = ~np.eye(world, dtype=bool)
    assert counts[off].max() < 0.05 * n                # a LET is a small part of a peer's octree
    # a sampled window of bodies: exact all-pairs (literal fp32 oracle) vs the single octree vs LET
    single = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(theta), p)
    single.encode()
    single.wait()
    one = nb.as_floats(single.dest_particle_slice())
    single.destroy()
    src = nb.as_floats(p)
    exact = np.concatenate([oracle.naive_step_f32(src, G, E, DT, lo, lo + 64)[lo:lo + 64]
                            for lo in (1000, n // 2 + 17, n - 5000)])       # 192 bodies, 8e8 pairs
    tag = exact[:, 9]

    def rows(state):
        order = np.argsort(state[:, 9], kind="stable")
        idx = order[np.searchsorted(state[order, 9], tag)]
        assert np.array_equal(state[idx, 9], tag)
        return state[idx]

    a_let, a_one, a_ex = rows(let)[:, 6:9], rows(one)[:, 6:9], exact[:, 6:9]
    norm = np.linalg.norm(a_ex, axis=1)
    err_one = np.median(np.linalg.norm(a_one - a_ex, axis=1) / norm)
    err_let = np.median(np.linalg.norm(a_let - a_ex, axis=1) / norm)
    assert err_one < 0.03 and err_let <= 1.25 * err_one + 1e-6, (err_let, err_one)
    assert np.array_equal(bits(rows(let)[:, 0:3]), bits(rows(one)[:, 0:3]))   # x' does not depend on the forces


def test_tree_sim_with_8388608_bodies(gpu, oracle):
    nb = gpu
    n = 8388608
    sp = nb.SimParams(particle_num=n)
    init = nb.inits.uniform_init(sp, seed=6)
    sim = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(0.75), init)
    sim.encode()
    sim.wait()                                         # status words clean, or this raises
    nodes = sim.tree_node_count()
    order = sim.debug_buffer("order", np.uint32)
    out = nb.as_floats(sim.dest_particle_slice())
    sim.destroy()
    assert 1.3 * n < nodes < 1.7 * n
    assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32))
    assert np.isfinite(out).all()
    src = nb.as_floats(init)
    assert np.array_equal(out[:, 9], src[order, 9])
    # forces of a window of sorted bodies against the oracle's own tree and per-thread walk
    tree, rw = oracle.tree_build(src)
    assert len(tree) == nodes
    assert np.array_equal(oracle.tree_dfs_order(tree, n), order)
    lo = n // 3
    want, _st = oracle.tree_walk_window(src[order], tree, rw, G, E, DT, 0.75, lo, lo + 256, order)
    got = out[lo:lo + 256]
    assert np.array_equal(bits(got[:, 0:3]), bits(want[:, 0:3]))
    a, b = got[:, 6:9].astype(np.float64), want[:, 6:9].astype(np.float64)
    err = np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-300)
    assert np.median(err) < 1e-5 and np.percentile(err, 99) < 1e-4 and err.max() < 5e-2
