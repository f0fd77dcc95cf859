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
print("two-phase (own tiles | other tiles + finish), HIP-event timed through torch-free ABI:")
import ctypes as C, time
for world in (2, 4, 8):
    sim = nb.NaiveSim.from_particles(sp, None, init, nb.Placement(0, 0, world))
    for _ in range(100):
        sim.encode_phase(0); sim.encode_phase(1)
    sim.wait()
    t0 = time.perf_counter()
    for _ in range(300):
        sim.encode_phase(0); sim.encode_phase(1)
    sim.wait()
    dt2 = (time.perf_counter() - t0) / 300
    t0 = time.perf_counter()
    for _ in range(300):
        sim.encode()
    sim.wait()
    dt1 = (time.perf_counter() - t0) / 300
    sim.destroy()
    print(f"world {world}: single-launch step {dt1*1e6:7.1f} us   two-phase step {dt2*1e6:7.1f} us", flush=True)
