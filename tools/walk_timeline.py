"""When the waves of the cells walk ran (needs a library built with -DNB_DIAG_TIMELINE or -DNB_DIAG_PHASES,
NB_LIB=...; 8 bodies per wave).  usage: walk_timeline.py N [THETA].  Builder tool."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import wgpu_n_body_amd as nb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
theta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.75
sp = nb.SimParams(particle_num=n)
init = nb.inits.uniform_init(sp, seed=n)
sim = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(theta), init)
sim.set_tuning("tree_walk_group", 8)
if len(sys.argv) > 3:
    for kv in sys.argv[3].split(","):
        k, v = kv.split("=")
        sim.set_tuning(k, int(v))
for _ in range(5):
    sim.encode()
sim.wait()
tot, walk = sim.encode_n_timed(5)
ph = sim.debug_buffer("phases", np.uint64).reshape(-1, 8)[: n // 8].astype(np.float64)
b = ph[:, 4]
t0 = ph[:, 7].min()
launch, start, end = (ph[:, 7] - t0) / 100.0, (ph[:, 5] - t0) / 100.0, (ph[:, 6] - t0) / 100.0   # us (100 MHz)
life = end - start
print("n %d theta %.2f: walk kernel %.1f us; waves %d; batches per wave mean %.1f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (
    n, theta, walk * 1e3, len(b), b.mean(), np.percentile(b, 50), np.percentile(b, 90), np.percentile(b, 99), b.max()))
print("  wave launch us: p10 %.1f p50 %.1f p90 %.1f max %.1f | prologue (launch -> first batch): mean %.1f p90 %.1f max %.1f" % (
    np.percentile(launch, 10), np.percentile(launch, 50), np.percentile(launch, 90), launch.max(), (start - launch).mean(),
    np.percentile(start - launch, 90), (start - launch).max()))
print("  wave start us: p50 %.1f p90 %.1f max %.1f | end: p50 %.1f p90 %.1f p99 %.1f max %.1f | life: mean %.1f p99 %.1f max %.1f | us per batch: mean %.2f" % (
    np.percentile(start, 50), np.percentile(start, 90), start.max(), np.percentile(end, 50), np.percentile(end, 90),
    np.percentile(end, 99), end.max(), life.mean(), np.percentile(life, 99), life.max(), (life / np.maximum(b, 1)).mean()))
# waves in flight over time
ts = np.linspace(0, end.max(), 21)
infl = [(np.logical_and(start <= t, end > t)).sum() for t in ts]
print("  in flight at " + " ".join("%.0f:%d" % (t, c) for t, c in zip(ts, infl)))
late = np.argsort(end)[-5:]
print("  last five waves: " + "; ".join("group %d batches %.0f start %.1f end %.1f" % (i, b[i], start[i], end[i]) for i in late))
if ph[:, 0].sum() > 0:
    print("  per wave cycles: pop %.0f load %.0f valu %.0f scan+push %.0f" % tuple(ph[:, :4].mean(0)))
    print("  per batch cycles: pop %.0f load %.0f valu %.0f scan+push %.0f" % tuple(ph[:, :4].sum(0) / b.sum()))
