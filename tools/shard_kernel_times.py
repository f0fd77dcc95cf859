"""Development tool: kernel time of one rank's share of the 65,536-body all-pairs step for
world = 1, 2, 4, 8 (run on one GPU; no exchange).  Shows what strong scaling can reach."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_n_body_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sp = nb.SimParams(particle_num=n)
init = nb.inits.uniform_init(sp, seed=2)
base = None
for world in (1, 2, 4, 8):
    sim = nb.NaiveSim.from_particles(sp, None, init, nb.Placement(0, 0, world))
    sim.encode_n_timed(100)
    tot, ker = sim.encode_n_timed(200)
    sim.destroy()
    base = base or ker
    print(f"world {world}: local {n//world:6d} bodies  kernel {ker*1e3:8.1f} us  total/step {tot/200*1e3:8.1f} us"
          f"  ideal {base/world*1e3:8.1f} us  kernel-efficiency {base/world/ker*100:5.1f}%", flush=True)
