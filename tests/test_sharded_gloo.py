"""The N>1 path on CPU: world_size-2/3 `gloo` runs of the sharded all-pairs host logic
(body-range partition + one in-place all-gather of position slices per step), with the CPU
oracle as the local step, must reproduce the single-process oracle bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import DT, E, G, ROOT, bits, make_state


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(tmp_path, world, n, steps, kind, seed):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), str(tmp_path),
             str(n), str(steps), kind, str(seed)], env=env, stdout=subprocess.PIPE,
            stderr=subprocess.STDOUT))
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, out.decode(errors="replace")[-2000:]
    return [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]


@pytest.mark.parametrize("world,n,kind", [(2, 1000, "uniform"), (2, 513, "spherical"),
                                          (3, 700, "uniform"), (2, 100, "disc")])
def test_gloo_sharded_run_equals_single_process_oracle(tmp_path, oracle, world, n, kind):
    steps, seed = 3, 31
    ranks = run_world(tmp_path, world, n, steps, kind, seed)
    want = oracle.naive_run_f32(make_state(kind, n, seed), G, E, DT, steps)
    covered = np.zeros(n, bool)
    for r in ranks:
        assert int(r["step_num"]) == steps
        # every rank ends with every body's position and mass
        assert np.array_equal(bits(r["posm"]), bits(want[:, [0, 1, 2, 9]]))
        lo, hi = int(r["lo"]), int(r["hi"])
        assert np.array_equal(bits(r["va"][lo:hi]), bits(want[lo:hi, 3:9]))
        covered[lo:hi] = True
    assert covered.all()


def test_shard_plan_ranges(nb):
    from wgpu_n_body_amd.sharded import ShardPlan
    for n, world in [(65536, 8), (262144, 8), (1000, 3), (100, 2), (5, 4)]:
        plan = ShardPlan(n, world)
        assert plan.padded == plan.per_rank * world >= n and plan.per_rank % 256 == 0
        edges = [plan.range(r) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(a[1] == b[0] or b[0] == n for a, b in zip(edges, edges[1:]))


@pytest.mark.parametrize("world,R", [(2, 8), (3, 12), (4, 8)])
def test_let_exchange_packs_every_peers_segment_in_rank_order(tmp_path, world, R):
    """The all-to-all-v of the LET protocol (exchange_segments) with gloo on CPU tensors: rank q
    must end up with counts[r][q] records of every rank r's segment q, packed in rank order,
    and nothing else touched; the counts matrix itself is all-gathered in place row by row."""
    port = free_port()
    seg_records = 8
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "tests", "_gloo_let_worker.py"), str(tmp_path),
             str(seg_records), str(R)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode(errors="replace")[-2000:]
    from tests._gloo_let_worker import counts_for
    want = counts_for(world)
    for me in range(world):
        r = np.load(os.path.join(tmp_path, f"let_rank{me}.npz"))
        assert np.array_equal(r["counts"], want)                  # the in-place all-gather of the rows
        assert r["got"].tolist() == [0 if src == me else int(want[src, me]) for src in range(world)]
        expect = []
        for src in range(world):
            if src == me:
                continue
            for k in range(int(want[src, me])):
                expect += [src * 1000 + me * 100 + k * 10 + e for e in range(R)]
        recv = r["recv"]
        assert recv[:len(expect)].tolist() == [float(v) for v in expect]
        assert (recv[len(expect):] == -2.0).all()                 # nothing written past the packed records
