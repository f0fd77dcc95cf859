"""Development tool: TreeSim (GPU) against the CPU oracle on one fixture-sized problem."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_n_body_amd as nb
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
theta = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
g, dt = (1e-5, 0.0016) if kind == "disc" else (1e-6, 0.016)
sp = nb.SimParams(particle_num=n, g=g, dt=dt)
init = getattr(nb.inits, kind + "_init")(sp, seed=22)
s0 = nb.as_floats(init).copy()
t0 = time.time()
ref = O.tree_step_f32(s0, sp.g, sp.e, sp.dt, theta, flags=O.INTENDED)
print(f"oracle step {time.time()-t0:.2f}s nodes {len(ref['tree'])} stats {ref['stats']}")
sim = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(theta), init)
sim.set_tuning("tree_count_visits", 1)
sim.encode(); sim.wait()
out = nb.as_floats(sim.dest_particle_slice())
tree, rw = sim.read_tree()
order = sim.debug_buffer("order", np.uint32)
print("status", sim.debug_buffer("status", np.uint32), "counters", sim.debug_buffer("counters", np.uint64))
print("root_width", rw, ref["root_width"], "n_nodes", len(tree), len(ref["tree"]))
print("order equal:", np.array_equal(order, ref["order"]))
m = min(len(tree), len(ref["tree"]))
print("bodies equal:", np.array_equal(tree["bodies"][:m], ref["tree"]["bodies"][:m]),
      "children equal:", np.array_equal(tree["children"][:m], ref["tree"]["children"][:m]))
if not np.array_equal(tree["children"][:m], ref["tree"]["children"][:m]):
    bad = np.nonzero((tree["children"][:m] != ref["tree"]["children"][:m]).any(1))[0]
    print(" first bad nodes", bad[:5]); 
    for b in bad[:3]: print(b, tree[b], ref["tree"][b])
print("mass rel err", np.abs(tree["mass"][:m] - ref["tree"]["mass"][:m]).max() / ref["tree"]["mass"][:m].max(),
      "cog abs err", np.abs(tree["cog"][:m] - ref["tree"]["cog"][:m]).max())
merr = np.abs(tree["mass"][:m] - ref["tree"]["mass"][:m])
worst = np.argsort(-merr)[:6]
for w in worst:
    print("  node", w, "bodies", tree["bodies"][w], "mass gpu", tree["mass"][w], "ref", ref["tree"]["mass"][w], "cog", tree["cog"][w], ref["tree"]["cog"][w])
print("pos bit-equal:", np.array_equal(out[:, 0:3].view(np.uint32), ref["dst"][:, 0:3].view(np.uint32)))
a, b = out[:, 6:9].astype(np.float64), ref["dst"][:, 6:9].astype(np.float64)
rel = np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)
print("acc rel err: median %.2e p99 %.2e max %.2e; finite %s" % (np.median(rel), np.percentile(rel, 99), rel.max(), np.isfinite(out).all()))
for k in (1, 3, 10):
    tot, ker = sim.encode_n_timed(k)
    print(f"  {k} steps: {tot/k:.3f} ms/step, walk {ker:.3f} ms")
bad = np.nonzero(rel > 1e-3)[0]
print("bad bodies:", len(bad), bad[:40], "lanes:", sorted(set(bad % 64))[:20])
