"""Latency of NB_PHASE_LET_BUILD (tree build + LET export) of ONE rank with the GPU to itself -- what a
rank sees on its own GPU -- for the two export forms (tree_let_export_mode 1: one launch, 0: a launch
per tree level).  All ranks of the group live on this GPU; only one works at a time.
    python tools/let_export_latency.py [bodies] [world]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_n_body_amd as nb  # noqa: E402
from tests.test_let_gpu import BUILD, META, WALK, LetGroup, tagged  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sp, p = tagged(nb, n, 5)
for mode in (0, 1):
    g = LetGroup(nb, sp, p, world, 0.5, export_mode=mode)
    for _ in range(3):
        g.step()
    lat = np.zeros((5, world))
    for it in range(5):
        for s in g.sims:
            s.encode_phase(META)
        g._all_gather(0)
        for r, s in enumerate(g.sims):
            s.wait()
            t0 = time.perf_counter()
            s.encode_phase(BUILD)
            s.wait()
            lat[it, r] = time.perf_counter() - t0
        counts = g._matrix(1)
        received = g._all_to_all(counts, 2, 3, 32)
        for me, s in enumerate(g.sims):
            s.let_set_imports(received[me])
        for s in g.sims:
            s.encode_phase(WALK)
        g.steps_done += 1
    off = counts[~np.eye(world, dtype=bool)]
    print(f"n {n} world {world} export mode {mode}: BUILD latency per rank mean {lat[1:].mean() * 1e6:7.1f} us, "
          f"slowest rank {lat[1:].mean(axis=0).max() * 1e6:7.1f} us; records per pair mean {off.mean():.0f} max {off.max()}")
    g.destroy()
