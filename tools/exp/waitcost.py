import os, sys, time
sys.path.insert(0, os.getcwd())
import wgpu_n_body_amd as nb
for n in (8192, 32768, 131072):
    sp = nb.SimParams(particle_num=n)
    r = nb.OfflineHeadless(nb.TreeSim, sp, nb.AddParams.TreeSimParams(0.75), lambda p: nb.inits.uniform_init(p, seed=n))
    for _ in range(100): r.step()
    best_s = best_n = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(300): r.step()
        best_s = min(best_s, (time.perf_counter() - t0) / 300)
        t0 = time.perf_counter()
        r.step_n(300)
        best_n = min(best_n, (time.perf_counter() - t0) / 300)
    print("n %d: step() %.1f us, step_n(300)/300 %.1f us" % (n, best_s * 1e6, best_n * 1e6), flush=True)
    r.destroy()
