"""Development tool: the one-process runners of nb_runner_create_multi (all-pairs: peer stores from the finish
kernel; Barnes-Hut: replicated tree + pushed slices) with random sizes and rank counts against the
one-device runner: the tree bit for bit, all-pairs to summation-order rounding.
python tools/multi_stress.py [iterations]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_n_body_amd as nb  # noqa: E402

rng = np.random.default_rng(2)
bad = 0
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for it in range(iters):
    n = int(rng.integers(50, 120000))
    world = int(rng.integers(2, 9))
    steps = int(rng.integers(1, 7))
    tree = it % 2 == 0
    sp = nb.SimParams(particle_num=n)
    init = nb.inits.uniform_init(sp, seed=it)
    kind = nb.TreeSim if tree else nb.NaiveSim
    add = nb.AddParams.TreeSimParams(0.5) if tree else None
    multi = nb.OfflineHeadless(kind, sp, add, lambda _p: init, device_ids=[0] * world)
    one = nb.OfflineHeadless(kind, sp, add, lambda _p: init)
    k = int(rng.integers(0, steps + 1))
    for _ in range(k):
        multi.step()
    if steps - k:
        multi.step_n(steps - k)
    one.step_n(steps)
    a, b = nb.as_floats(multi.read_particles()), nb.as_floats(one.read_particles())
    multi.destroy()
    one.destroy()
    if tree:
        ok = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    else:
        scale = np.abs(b[:, 6:9]).max()
        ok = np.abs(a[:, 6:9] - b[:, 6:9]).max() <= 2e-5 * scale and np.abs(a[:, 0:3] - b[:, 0:3]).max() <= 1e-6
    if not ok:
        bad += 1
        print(f"it {it}: {'tree' if tree else 'naive'} n {n} world {world} steps {steps}: MISMATCH", flush=True)
print("done, failures:", bad, "of", iters)
