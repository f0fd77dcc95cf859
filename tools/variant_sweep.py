"""Sweep the all-pairs kernel variants on the GPU: parity vs the CPU oracle at a small N,
then HIP-event timing at the benchmark N.  Development tool (run through gpurun)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_n_body_amd as nb  # noqa: E402
from oracle import oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--check-n", type=int, default=3000)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--variants", type=str, default="")
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--jsplit", type=int, default=0)
args = ap.parse_args()

names = nb.naive_variants()
sel = [int(v) for v in args.variants.split(",")] if args.variants else list(range(len(names)))

# parity at a ragged small N (exercises the masked tiles and the tail)
spc = nb.SimParams(particle_num=args.check_n)
init = nb.inits.spherical_init(spc, seed=7)
st = nb.as_floats(init)
ref32 = O.naive_run_f32(st, spc.g, spc.e, spc.dt, 2)
ref64 = O.naive_run_f64(st, spc.g, spc.e, spc.dt, 2)
scale = np.abs(ref64[:, 6:9]).max()
print(f"oracle f32 vs f64: acc err/scale {np.abs(ref32[:,6:9]-ref64[:,6:9]).max()/scale:.3e}")

sp = nb.SimParams(particle_num=args.n)
big = nb.inits.uniform_init(sp, seed=2)
results = {}
for v in sel:
    sim = nb.NaiveSim.from_particles(spc, None, init)
    sim.set_tuning("naive_variant", v)
    sim.encode(); sim.encode(); sim.wait()
    out = nb.as_floats(sim.dest_particle_slice())
    e_acc = np.abs(out[:, 6:9] - ref64[:, 6:9]).max() / scale
    e_pos = np.abs(out[:, 0:3] - ref64[:, 0:3]).max()
    e_vel = np.abs(out[:, 3:6] - ref64[:, 3:6]).max() / np.abs(ref64[:, 3:6]).max()
    pos_bits = np.array_equal(out[:, 0:3].view(np.uint32), ref32[:, 0:3].view(np.uint32))
    sim.destroy()
    results[v] = dict(name=names[v], e_acc=float(e_acc), e_pos=float(e_pos), e_vel=float(e_vel))
    print(f"[{v:2d}] {names[v]:28s} acc err/scale {e_acc:.3e}  pos abs {e_pos:.3e}  vel rel {e_vel:.3e}"
          f"  finite={np.isfinite(out).all()}", flush=True)

for r in range(args.rounds):
    for v in sel:
        sim = nb.NaiveSim.from_particles(sp, None, big)
        sim.set_tuning("naive_variant", v)
        sim.set_tuning("naive_jsplit", args.jsplit)
        sim.encode_n_timed(3)
        tot, ker = sim.encode_n_timed(args.steps)
        sim.destroy()
        pairs = args.n * (args.n - 1)
        rate = pairs / (ker * 1e-3)
        results[v].setdefault("ms", []).append(ker)
        print(f"round {r} [{v:2d}] {names[v]:28s} kernel {ker:8.4f} ms  total/step {tot/args.steps:8.4f} ms"
              f"  {rate/1e12:6.3f} Tpairs/s  {20*rate/1e12:6.2f} TFLOP/s ({20*rate/157.3e12*100:5.1f}% of 157.3)",
              flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(results, open("gpurun_out/variant_sweep.json", "w"), indent=1)
