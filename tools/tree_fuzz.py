"""Random Barnes-Hut cases against the CPU oracle (test infrastructure, like tests/): sizes, distributions,
theta, cube scale and walk shape drawn at random; the checks are tests/test_tree_gpu.py's (tree and order
bit-exact, positions bit-exact after one step, accelerations within the fp32 tolerances, visit counts).
usage: tree_fuzz.py [SECONDS [SEED]]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wgpu_n_body_amd as nb  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.helpers import DT, E, G, bits, make_state  # noqa: E402
from tests.test_tree_gpu import WALK_SHAPES, check_tree, rel_err, run_tree  # noqa: E402


def fuzz(budget=60.0, seed=1, max_cases=None, log=print):
    """Returns (cases, failures, flips)."""
    O.build()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    cases = fails = flips = 0
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        kind = ["uniform", "spherical", "disc"][rng.integers(3)]
        n = int(rng.choice([rng.integers(1, 70), rng.integers(70, 600), rng.integers(600, 6000), rng.integers(6000, 30000)]))
        theta = float(rng.choice([0.3, 0.5, 0.6, 0.75, 0.9, 1.0, 1.3]))
        g, dt = (1e-5, 0.0016) if kind == "disc" else (G, DT)
        s = make_state(kind, n, int(rng.integers(1 << 30)), g)
        scale = float(rng.choice([1.0, 1.0, 0.01, 7.5, 300.0]))
        s[:, 0:3] *= np.float32(scale)
        shape = dict(WALK_SHAPES[rng.integers(len(WALK_SHAPES))])
        if rng.integers(3) == 0:
            shape["tree_cell_scan_inline"] = int(rng.integers(3))
        if rng.integers(4) == 0:
            shape["tree_rank_sort_max"] = int(rng.choice([0, 1 << 20]))
        if rng.integers(3) == 0:
            shape["tree_walk_gathers"] = int(rng.choice([0, 2]))
        if rng.integers(4) == 0:
            shape["tree_key_descent"] = 1
        tag = f"{kind} n={n} theta={theta} scale={scale} {shape}"
        cases += 1
        try:
            ref = O.tree_step_f32(s, g, E, dt, theta, flags=O.INTENDED)
            r = run_tree(nb, s, theta, 1, g, E, dt, tuning=shape)
            if r["status"].any():
                log("status", r["status"][:4], tag)
                continue
            check_tree(r["tree"], r["root_width"], r["order"], ref["tree"], ref["root_width"], ref["order"],
                       extent=float(np.abs(s[:, 0:3]).max()))
            # check_step's tolerances, but its cap on the worst body (5e-2: one flipped acceptance test at theta <= 1)
            # widened for wide theta: a flip costs one cell's approximation error, ~theta^2
            assert np.isfinite(r["dst"]).all()
            assert np.array_equal(bits(r["dst"][:, 0:3]), bits(ref["dst"][:, 0:3]))
            assert np.array_equal(r["dst"][:, 9], ref["dst"][:, 9])
            err = rel_err(r["dst"][:, 6:9], ref["dst"][:, 6:9])
            assert np.median(err) < 1e-5, np.median(err)
            assert np.percentile(err, 99) < 1e-4, np.percentile(err, 99)
            assert (err > 1e-3).sum() <= max(2, int(2e-4 * n)), (err > 1e-3).sum()
            # the worst body: one flipped acceptance test costs it that cell's approximation error -- a few per cent
            # of the CELL's pull, which can be a large part of the body's NET acceleration where the pulls cancel
            # (the middle of a uniform cube).  Measured against the typical acceleration as well:
            worst = int(np.argmax(err))
            a_ref = np.linalg.norm(ref["dst"][:, 6:9].astype(np.float64), axis=1)
            d_abs = np.linalg.norm(r["dst"][worst, 6:9].astype(np.float64) - ref["dst"][worst, 6:9].astype(np.float64))
            cap = 5e-2 * max(1.0, (theta / 0.75) ** 2) * 2.0
            assert err[worst] < cap or d_abs < cap * np.median(a_ref), (
                worst, err[worst], d_abs / np.median(a_ref), a_ref[worst] / np.median(a_ref))
            # (an acceptance test within an ulp of theta may flip -- size^2 / theta^2 < r^2 here, the oracle's
            # literal form there -- and takes the cell's subtree in or out of the walk: reported, and bounded)
            dv = int(r["counters"][0]) - ref["stats"]["visits"]
            da = int(r["counters"][1]) - ref["stats"]["accepted"]
            if abs(dv) > max(2, 1e-5 * ref["stats"]["visits"]) or abs(da) > max(2, 1e-5 * ref["stats"]["accepted"]):
                flips += 1
                log("flip?", tag, "visits %+d of %d, accepted %+d of %d, bodies off by > 1e-3: %d" % (
                    dv, ref["stats"]["visits"], da, ref["stats"]["accepted"], (err > 1e-3).sum()))
            assert abs(dv) <= max(200, 1e-4 * ref["stats"]["visits"])
            assert abs(da) <= max(200, 1e-4 * ref["stats"]["accepted"])
        except AssertionError as ex:
            import traceback
            tb = traceback.extract_tb(ex.__traceback__)[-1]
            fails += 1
            log("FAIL", tag, f"{tb.name}:{tb.lineno}: {tb.line}", repr(ex)[:200])
        except Exception as ex:  # bodies on one 63-bit key (the reference's build would not terminate): reported, not counted
            log("error", tag, repr(ex)[:200])
    return cases, fails, flips


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cases, fails, flips = fuzz(budget, seed, log=lambda *a: print(*a, flush=True))
    print(f"done: {cases} cases, failures: {fails}, walks with a flipped acceptance test: {flips}")
    sys.exit(1 if fails else 0)
