"""The C++ host mirror (csrc/simulator.hpp: Simulator / NaiveSim / TreeSim / OfflineHeadless<T>) and
the headless CLI built on it (csrc/headless.cpp), the counterparts of the reference's
src/runners/offline_headless.rs and src/bin/headless.rs:14-35 -- driven as a subprocess, its
output format checked against headless.rs:32, its final state (--dump, the F3 snapshot layout)
against the CPU oracle.  `-m gpu`."""
import os
import re
import subprocess

import numpy as np
import pytest

from tests.helpers import DT, E, G, ROOT, bits

pytestmark = pytest.mark.gpu

CLI = os.path.join(ROOT, "wgpu_n_body_amd", "headless")


def run_cli(args, tmp_path):
    out = os.path.join(tmp_path, "final.nbsnap")
    p = subprocess.run([CLI] + [str(a) for a in args] + ["--dump", out], capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout, out


def test_cli_naive_against_oracle(gpu, oracle, tmp_path):
    from wgpu_n_body_amd.snapshot import load_snapshot
    nb = gpu
    assert os.path.exists(CLI), "the headless CLI is built by wgpu_n_body_amd.build"
    text, path = run_cli(["--sim", "naive", "--n", 1024, "--init", "spherical", "--seed", 1, "--steps", 10], tmp_path)
    lines = text.strip().splitlines()
    # src/bin/headless.rs:21,29,32,34: the four kinds of line, one duration per step
    assert lines[0] == "Initializing Simulation" and lines[1] == "Running Simulation"
    assert lines[-1] == "Finished Running"
    steps = [ln for ln in lines if ln.startswith("Step Duration: ")]
    assert len(steps) == 10 and all(re.fullmatch(r"Step Duration: \d+ \u00b5s", ln) for ln in steps)
    sp, parts, step = load_snapshot(path)
    assert step == 10 and sp.particle_num == 1024
    assert (np.float32(sp.g), np.float32(sp.e), np.float32(sp.dt)) == (np.float32(G), np.float32(E), np.float32(DT))
    init = nb.as_floats(nb.inits.spherical_init(nb.SimParams(particle_num=1024), seed=1))
    ref = oracle.naive_run_f64(init, G, E, DT, 10)
    got = nb.as_floats(parts)
    scale = np.abs(ref[:, 6:9]).max()
    assert np.abs(got[:, 6:9] - ref[:, 6:9]).max() / scale < 2e-5
    assert np.abs(got[:, 0:3] - ref[:, 0:3]).max() < 2e-6


def test_cli_tree_against_oracle(gpu, oracle, tmp_path):
    from wgpu_n_body_amd.snapshot import load_snapshot
    nb = gpu
    text, path = run_cli(["--sim", "tree", "--n", 4096, "--theta", 0.5, "--init", "uniform", "--seed", 9,
                          "--steps", 1], tmp_path)
    assert text.count("Step Duration: ") == 1
    sp, parts, step = load_snapshot(path)
    assert step == 1 and sp.particle_num == 4096
    init = nb.as_floats(nb.inits.uniform_init(nb.SimParams(particle_num=4096), seed=9))
    ref = oracle.tree_step_f32(init, G, E, DT, 0.5, flags=oracle.INTENDED)
    got = nb.as_floats(parts)
    assert np.array_equal(bits(got[:, 0:3]), bits(ref["dst"][:, 0:3]))      # sorted (DFS) order, bit-exact
    a, b = got[:, 6:9].astype(np.float64), ref["dst"][:, 6:9].astype(np.float64)
    err = np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-300)
    assert np.median(err) < 1e-5 and np.percentile(err, 99) < 1e-4 and err.max() < 5e-2


def test_cli_reports_errors_with_a_status(gpu, tmp_path):
    p = subprocess.run([CLI, "--sim", "tree", "--n", "64", "--bogus", "1"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 2 and "unknown option" in p.stderr
