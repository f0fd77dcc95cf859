"""Shared helpers for the test modules."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

G, E, DT = 0.000001, 0.0001, 0.016  # SimParams::default, src/sims/mod.rs:62-71


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def make_state(kind: str, n: int, seed: int, g: float = G) -> np.ndarray:
    """Seeded float32[n,10] initial state from the product's own (host-side) inits."""
    import wgpu_n_body_amd as nb
    sp = nb.SimParams(particle_num=n, g=g)
    fn = {"uniform": nb.inits.uniform_init, "disc": nb.inits.disc_init,
          "spherical": nb.inits.spherical_init}[kind]
    return nb.as_floats(fn(sp, seed=seed)).copy()


def run_workers(cmds, port, tmp_path, timeout=300):
    """Start one worker process per rank (output to files, not pipes); on a time-out or a failure
    kill every sibling before reporting, so that no orphan keeps the GPU."""
    import subprocess
    world = len(cmds)
    procs, logs = [], []
    for rank, cmd in enumerate(cmds):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        log = open(os.path.join(tmp_path, f"worker{rank}.log"), "w")
        logs.append(log)
        procs.append(subprocess.Popen(cmd, env=env, stdout=log, stderr=subprocess.STDOUT))
    failed = None
    try:
        for rank, p in enumerate(procs):
            try:
                if p.wait(timeout=timeout) != 0 and failed is None:
                    failed = rank
            except subprocess.TimeoutExpired:
                failed = rank if failed is None else failed
                break
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
        for log in logs:
            log.close()
    if failed is not None:
        tails = "\n".join(f"--- rank {r} ---\n" + open(os.path.join(tmp_path, f"worker{r}.log")).read()[-3000:]
                          for r in range(world))
        raise AssertionError(f"worker {failed} failed or timed out\n{tails}")


