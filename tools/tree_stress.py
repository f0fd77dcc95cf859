"""Development tool: hunt for nondeterminism in the tree build.  Many fresh TreeSims, one step
each; every internal node's mass must equal its body count (unit masses) and the whole step
must be bit-reproducible."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_n_body_amd as nb

rng = np.random.default_rng(0)
bad = 0
keep = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    n = int(rng.integers(2000, 90000))
    sp = nb.SimParams(particle_num=n)
    init = nb.inits.uniform_init(sp, seed=it)
    outs = []
    for rep in range(2):
        sim = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(0.5), init)
        sim.encode(); sim.wait()
        tree, rw = sim.read_tree()
        out = sim.dest_particle_slice()
        st = sim.debug_buffer("status", np.uint32)
        keep.append(sim) if rep == 0 and it % 3 == 0 else sim.destroy()   # vary allocator reuse
        if len(keep) > 3:
            keep.pop(0).destroy()
        ok_mass = np.array_equal(tree["mass"], tree["bodies"].astype(np.float32))
        if not ok_mass or st.any():
            bad += 1
            w = np.nonzero(tree["mass"] != tree["bodies"].astype(np.float32))[0]
            print(f"it {it} rep {rep} n {n}: mass mismatch at {len(w)} nodes, first {w[:5]}, "
                  f"mass {tree['mass'][w[:5]]} bodies {tree['bodies'][w[:5]]} status {st}", flush=True)
        outs.append(out.copy())
    if not np.array_equal(outs[0], outs[1]):
        bad += 1
        print(f"it {it} n {n}: two identical runs differ", flush=True)
print("done, failures:", bad)
