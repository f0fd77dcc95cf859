"""Property-based checks (hypothesis) of the CPU-side pieces: the two tree restatements agree on
arbitrary small point sets, the shard arithmetic always tiles [0, N), and the all-pairs oracle
is equivariant under reflections.  CPU only."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import oracle_np as P
from tests.helpers import DT, E, G, bits

coords = st.floats(min_value=-4.0, max_value=4.0, allow_nan=False, width=32)


def state_from(points, masses):
    s = np.zeros((len(points), 10), np.float32)
    s[:, 0:3] = np.array(points, np.float32)
    s[:, 9] = np.array(masses, np.float32)
    return s


@settings(max_examples=40, deadline=None)
@given(st.lists(st.tuples(coords, coords, coords), min_size=2, max_size=40, unique=True),
       st.floats(min_value=0.2, max_value=1.2), st.data())
def test_tree_restatements_agree_on_arbitrary_points(oracle, points, theta, data):
    # distinct points only (coincident bodies never terminate in the reference's builder)
    pts = np.array(points, np.float32)
    if len(np.unique(pts, axis=0)) != len(pts):
        return
    masses = data.draw(st.lists(st.floats(min_value=0.5, max_value=3.0, width=32),
                                min_size=len(points), max_size=len(points)))
    s = state_from(points, masses)
    try:
        a = oracle.tree_step_f32(s, G, E, DT, theta, flags=oracle.INTENDED, max_depth=40)
    except RuntimeError:
        return  # too deep for the cap: points closer than the oracle is asked to separate
    b = P.tree_step(s, G, E, DT, theta, flags=7)
    assert len(a["tree"]) == len(b["nodes"])
    assert np.array_equal(a["order"], b["order"])
    for t, nd in zip(a["tree"], b["nodes"]):
        assert t["bodies"] == nd["bodies"] and list(t["children"]) == list(nd["children"])
    assert np.array_equal(bits(a["dst"]), bits(b["dst"]))
    # structural invariants
    leaves = a["tree"][a["tree"]["bodies"] == 1]
    assert sorted(leaves["children"][:, 0]) == list(range(len(points)))
    assert a["root_width"] == 2 * max(1.0, float(np.abs(s[:, 0:3]).max()))


@settings(max_examples=200, deadline=None)
@given(st.integers(min_value=0, max_value=5_000_000), st.integers(min_value=1, max_value=16))
def test_shard_ranges_tile_the_bodies(nb, n, world):
    per = nb.shard_bodies_per_rank(n, world)
    assert per % 256 == 0 and per >= 256
    assert nb.shard_padded_bodies(n, world) == per * world >= n
    assert per * world - n < 256 * world + per      # padding stays small
    total = 0
    for r in range(world):
        lo, hi = min(n, r * per), min(n, (r + 1) * per)
        assert lo == total or lo == n
        total = hi
    assert total == n


@settings(max_examples=25, deadline=None)
@given(st.lists(st.tuples(coords, coords, coords), min_size=2, max_size=24, unique=True),
       st.sampled_from([0, 1, 2]))
def test_all_pairs_oracle_is_reflection_equivariant(oracle, points, axis):
    s = state_from(points, [1.0] * len(points))
    s[:, 3:6] = s[:, 0:3] * np.float32(0.01)
    a = oracle.naive_step_f32(s, G, E, DT)
    m = s.copy()
    m[:, [axis, 3 + axis, 6 + axis]] *= -1
    b = oracle.naive_step_f32(m, G, E, DT)
    b[:, [axis, 3 + axis, 6 + axis]] *= -1
    assert np.array_equal(a, b, equal_nan=True)   # value equality: -0.0 == 0.0


def test_morton_domains_cut_on_cell_borders():
    """Start-up domains of the LET scheme: a permutation, near-equal counts, cuts that sit on
    octree-cell borders (8 ranks on a uniform cube = the 8 octants), splits = first key of each
    domain, and degenerate sizes do not break it."""
    import wgpu_n_body_amd as nb
    from wgpu_n_body_amd.sharded import _morton_keys, morton_domains
    sp = nb.SimParams(particle_num=400000)
    p = nb.inits.uniform_init(sp, seed=3)
    order, cuts, splits, ref = morton_domains(p, 8, with_owners=True)
    assert sorted(order.tolist()) == list(range(400000))
    assert cuts[0] == 0 and cuts[-1] == 400000 and all(b >= a for a, b in zip(cuts, cuts[1:]))
    counts = np.diff(cuts)
    assert counts.min() > 0.98 * 50000 and counts.max() < 1.02 * 50000
    pos = nb.as_floats(p)[order][:, 0:3]
    for r in range(8):
        d = pos[cuts[r]:cuts[r + 1]]
        sign = [(r >> k) & 1 for k in range(3)]             # x is the lowest key bit of a level
        for k in range(3):
            assert (d[:, k] >= 0).all() if sign[k] else (d[:, k] <= 0).all()
    keys = np.sort(_morton_keys(p))
    assert splits == [int(keys[c]) for c in cuts[1:-1]] and splits == sorted(splits)
    assert 0.99 < ref <= 1.0
    for n, w in ((7, 4), (1, 2), (2, 4), (0, 3)):
        q = nb.inits.uniform_init(nb.SimParams(particle_num=max(n, 1)), seed=1)[:n]
        if n == 0:
            continue
        o, c = morton_domains(q, w)
        assert sorted(o.tolist()) == list(range(n)) and c[0] == 0 and c[-1] == n and len(c) == w + 1
