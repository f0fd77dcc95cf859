"""Development tool: 300 steps of 2^20 bodies (spherical_init) through the three Barnes-Hut runners -- one device,
nb_runner_create_multi (replicated tree, 8 ranks on this GPU; must stay bit-equal to one device) and
nb_runner_create_multi_let (8 domains, migration every 4th step; every body exactly once, positions and kinetic
energy tracking the one-tree run)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_n_body_amd as nb
n = 1 << 20
sp = nb.SimParams(particle_num=n)
p = nb.inits.spherical_init(sp, seed=4).copy()
f = nb.as_floats(p); f[:, 9] = 1.0 + np.arange(n, dtype=np.float32) / np.float32(2 * n)
let = nb.OfflineHeadless(nb.TreeSim, sp, nb.AddParams.TreeSimParams(0.5), lambda _p: p, device_ids=[0] * 8, let_migrate_every=4)
one = nb.OfflineHeadless(nb.TreeSim, sp, nb.AddParams.TreeSimParams(0.5), lambda _p: p)
rep = nb.OfflineHeadless(nb.TreeSim, sp, nb.AddParams.TreeSimParams(0.5), lambda _p: p, device_ids=[0] * 8)
t0 = time.time()
for chunk in range(6):
    let.step_n(50); one.step_n(50); rep.step_n(50)
    a = nb.as_floats(let.read_particles()); b = nb.as_floats(one.read_particles()); c = nb.as_floats(rep.read_particles())
    a = a[np.argsort(a[:, 9], kind="stable")]; bb = b[np.argsort(b[:, 9], kind="stable")]
    ok_rep = np.array_equal(b.view(np.uint32), c.view(np.uint32))
    ke = lambda s: 0.5 * float((s[:, 9].astype(np.float64) * (s[:, 3:6].astype(np.float64) ** 2).sum(axis=1)).sum())
    print(f"steps {50 * (chunk + 1)}: LET bodies {len(np.unique(a[:, 9]))} finite {np.isfinite(a).all()} max |dx| vs one tree {np.abs(a[:, 0:3] - bb[:, 0:3]).max():.2e} "
          f"KE let {ke(a):.6e} one {ke(bb):.6e}; replicated == single: {ok_rep}; {time.time() - t0:.1f} s", flush=True)
