"""bench.py's contract on a GPU box -- `-m gpu`: one JSON line with the driver's keys, the same schema from
both hosts and for N = 1 and N > 1 (the N > 1 runs rehearsed with every rank on device 0: NB_BENCH_SAME_DEVICE=1,
and gloo standing in for RCCL, which refuses several ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu

DRIVER_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
               "vs_baseline", "dtype", "data", "config", "roofline"}


def run_bench(argv, env=None, launcher=None, timeout=600):
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + argv
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def check_line(d, n_gpus, steps, warmup, shared=False):
    assert DRIVER_KEYS <= set(d), DRIVER_KEYS - set(d)
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["unit"] == "pairs/s" and d["dtype"] == "f32" and d["vs_baseline"] is None and d["scaling"] == "strong"
    assert d["config"]["bodies"] == 65536 and "src:" in d["config"]["lib"] and "gfx950" in d["config"]["lib"]
    assert abs(d["value"] - 65536 * 65535 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # (ranks that share one GPU run their kernels side by side: a rank's kernels then take longer than a step)
    assert shared or r["kernel_ms"] <= d["ms_per_step"] * 1.05


def test_one_gpu_both_hosts_same_schema():
    a = run_bench(["--steps", "20", "--warmup", "3"])                                   # the driver's default shape
    b = run_bench(["--steps", "20", "--warmup", "3", "--host", "native", "--no-cpu-baseline", "--no-tree",
                   "--no-criterion"])
    check_line(a, 1, 20, 3)
    check_line(b, 1, 20, 3)
    assert 0.9 < a["value"] / b["value"] < 1.1                       # the same kernel behind both boundaries
    assert set(b) - {"ranks"} <= set(a)
    # the extras of the default N = 1 line
    assert a["cpu_baseline"]["kind"] == "port" and a["cpu_baseline"]["cores"] >= 1
    rows = a["criterion"]["rows"]
    assert [(r["group"], r["n"]) for r in rows] == [(g, n) for g in ("naive", "tree")
                                                    for n in (8192, 16384, 32768, 65536, 131072)]
    assert all(r["iterations"] >= 200 and 5 < r["us_per_step_median"] < 2e4 for r in rows)
    t = a["tree_1m_theta05"]
    assert t["bodies"] == 1 << 20 and 0.0 < t["build_hbm_frac"] < 1.0 and t["cpu_baseline"]["kind"] == "port"
    assert a["tree_4m_theta075_headless"]["bodies"] == 4000000


def test_two_ranks_both_hosts_same_schema():
    env = {"NB_BENCH_SAME_DEVICE": "1"}
    nat = run_bench(["--gpus", "2", "--steps", "10", "--warmup", "3", "--host", "native"], env=env)
    check_line(nat, 2, 10, 3, shared=True)
    assert len(nat["ranks"]["rank_kernel_ms"]) == 2 and all(x > 0 for x in nat["ranks"]["rank_kernel_ms"])
    for key in ("config3_262144_allpairs", "config4_4m_let_theta05"):
        got = nat[key]["native_host"]
        assert "error" not in got, got
        assert len(got["rank_kernel_ms"]) == 2 and got["ms_per_step"] > 0
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", "29631"]
    rccl = run_bench(["--gpus", "2", "--steps", "10", "--warmup", "3"], env=dict(env, NB_DIST_BACKEND="gloo"),
                     launcher=launcher)
    check_line(rccl, 2, 10, 3, shared=True)
    assert "error" not in rccl["native_host"], rccl["native_host"]
    assert "error" not in rccl["config3_262144_allpairs"]["rccl_host"]
    assert set(nat) - {"ranks"} <= set(rccl) | {"ranks"}
