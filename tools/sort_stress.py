"""Development tool: the high-word sort + thread-per-body fix-up + adaptive digits against the full 8-pass sort.
Random sizes around the rule's borders, random distributions (uniform, disc, sphere, a dense core with escapers,
a thin slab), several steps with a wait after each (so the feedback changes the number of digits mid-run): body
order and new state must be bit for bit the full sort's on every step."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_n_body_amd as nb

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
sizes = [16385, 16640, 20000, 65535, 65536, 65537, 131071, 131072, 131073, 262144, 300001, 524288, 1 << 20]
t_end = time.time() + budget
bad = it = 0
while time.time() < t_end:
    n = int(rng.choice(sizes)) if it % 2 == 0 else int(rng.integers(16385, 400000))
    kind = ["uniform", "disc", "spherical", "core", "slab"][it % 5]
    sp = nb.SimParams(particle_num=n, g=1e-9)
    if kind in ("uniform", "disc", "spherical"):
        state = nb.as_floats(getattr(nb.inits, kind + "_init")(sp, seed=100 + it)).copy()
    else:
        state = np.zeros((n, 10), np.float32)
        state[:, 9] = 1.0
        if kind == "core":
            c = int(0.85 * n)
            state[:c, 0:3] = rng.uniform(-0.02, 0.02, (c, 3)) + 0.25
            state[c:, 0:3] = rng.uniform(-1, 1, (n - c, 3))
        else:
            state[:, 0:2] = rng.uniform(-1, 1, (n, 2))
            state[:, 2] = rng.uniform(-0.002, 0.002, n)
        state = state[rng.permutation(n)]
    sims = []
    for mode in (1, 0):
        s = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(0.75), state)
        s.set_tuning("tree_sort_mode", mode)
        sims.append(s)
    try:
        for step in range(4):
            outs = []
            for s in sims:
                s.encode(); s.wait()
                outs.append((s.debug_buffer("order", np.uint32).copy(), s.dest_particle_slice().copy()))
            if not (np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])):
                bad += 1
                print(f"MISMATCH it {it} n {n} {kind} step {step}", flush=True)
                break
    except nb.NBodyError as e:   # (colliding keys in a dense sample: both sorts report it)
        print(f"it {it} n {n} {kind}: {e}", flush=True)
    for s in sims:
        s.destroy()
    it += 1
print(f"done: {it} cases, failures: {bad}")
sys.exit(1 if bad else 0)
