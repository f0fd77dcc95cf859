import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import wgpu_n_body_amd as nb
from tests.helpers import bits
from tests.test_tree_gpu import run_tree
bad = 0
for n in (524288, 1000003, 2097169, 4000000):
    for init in ("uniform", "spherical", "disc"):
        sp = nb.SimParams(particle_num=n)
        state = nb.as_floats(getattr(nb.inits, init + "_init")(sp, seed=n % 1000))
        g, dt = (1e-5, 0.0016) if init == "disc" else (1e-6, 0.016)
        try:
            a = run_tree(nb, state, 0.75, steps=3, g=g, dt=dt, count=False, tuning={"tree_walk_gathers": 0})
            b = run_tree(nb, state, 0.75, steps=3, g=g, dt=dt, count=False)
        except Exception as ex:
            print(n, init, "error", repr(ex)[:120], flush=True)
            continue
        ok = (np.array_equal(a["order"], b["order"]) and np.array_equal(bits(a["dst"]), bits(b["dst"]))
              and a["tree"].tobytes() == b["tree"].tobytes() and not a["status"].any() and not b["status"].any())
        bad += not ok
        print(n, init, "equal" if ok else "DIFFERENT", flush=True)
print("done, different:", bad)
sys.exit(1 if bad else 0)
