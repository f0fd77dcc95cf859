"""Per-phase cycle counts of the cells walk (needs a library built with -DNB_DIAG_PHASES, NB_LIB=...).
The probe forces s_waitcnt(0) at the phase borders, so loads no longer overlap compute inside a
wave: it measures latencies, not the production schedule.  Builder tool."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import wgpu_n_body_amd as nb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sp = nb.SimParams(particle_num=n)
init = nb.inits.uniform_init(sp, seed=3)
sim = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(0.5), init)
sim.set_tuning("tree_walk_group", G)
for _ in range(3):
    sim.encode()
sim.wait()
tot, walk = sim.encode_n_timed(5)
ph = sim.debug_buffer("phases", np.uint64).reshape(-1, 4)[: n // G].astype(np.float64)
print("walk kernel %.3f ms with the probe; per wave cycles: pop %.0f load %.0f valu %.0f scan+push %.0f (sum %.0f)" % (
    (walk,) + tuple(ph.mean(0)) + (ph.sum(1).mean(),)))
