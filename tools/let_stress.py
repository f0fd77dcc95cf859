"""Development tool: hunt for ordering bugs in the one-process LET runner (nb_runner_create_multi_let: rank
threads, events, peer stores).  Random body counts, rank counts and migration periods; after a few steps
every rank must hold, bit for bit, what the Python-hosted protocol (tests/test_let_gpu.py LetGroup:
one rank after the other, exchanges by hipMemcpy) holds.  python tools/let_stress.py [iterations]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_n_body_amd as nb  # noqa: E402
from tests.test_let_gpu import LetGroup, tagged  # noqa: E402

rng = np.random.default_rng(1)
bad = 0
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for it in range(iters):
    n = int(rng.integers(300, 70000))
    world = int(rng.integers(2, 9))
    migrate = int(rng.choice([0, 1, 2, 5]))
    theta = float(rng.choice([0.4, 0.5, 0.75]))
    steps = int(rng.integers(2, 9))
    sp, p = tagged(nb, n, 1000 + it, "spherical" if it % 4 == 3 else "uniform")
    grp = LetGroup(nb, sp, p, world, theta, migrate_every=migrate)
    native = nb.OfflineHeadless(nb.TreeSim, sp, nb.AddParams.TreeSimParams(theta), lambda _p: p,
                                device_ids=[0] * world, let_migrate_every=migrate)
    try:
        for _ in range(steps):
            grp.step()
        # some steps one by one, the rest in one call
        k = int(rng.integers(0, steps + 1))
        for _ in range(k):
            native.step()
        if steps - k:
            native.step_n(steps - k)
        a = nb.as_floats(native.read_particles()).view(np.uint32)
        b = nb.as_floats(grp.particles()).view(np.uint32)
        same = a.shape == b.shape and np.array_equal(a, b)
    except nb.NBodyError as e:
        same = False
        print(f"it {it}: {e}", flush=True)
    if not same:
        bad += 1
        print(f"it {it}: n {n} world {world} migrate {migrate} theta {theta} steps {steps}: MISMATCH", flush=True)
    native.destroy()
    grp.destroy()
print("done, failures:", bad, "of", iters)
