"""All-pairs parity on a real MI355X, through the C ABI (libnbody_hip.so) -- `-m gpu`.

Hot path under test: nb_sim_encode on a NaiveSim (nb_naive.hip), the replacement for
src/sims/shaders/naive.wgsl:23-69 + NaiveSim::encode (src/sims/naive.rs:147-162).

Tolerances (fp32 path; the reference's own result is not bit-defined -- WGSL `distance`,
`normalize`, `/` have driver ULP slack -- and the kernel sums j in a different order):
  * positions after ONE step: bit-identical to the literal-fp32 oracle (the integrator lines
    are evaluated operation for operation; x' depends only on the body's own x, v, a);
  * stored acceleration: max |err| / max |acc| <= ACC_TOL = 2e-5 against the fp64 oracle, and
    no worse than 4x the literal-fp32 oracle's own error against fp64 (+ a 1e-6 floor);
  * after 10 steps: positions max |err| <= 2e-6 (box scale ~1), velocities
    max |err| / max |v| <= 2e-5, kinetic energy and momentum relative error <= 2e-5.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from tests.helpers import DT, E, G, GOLDEN, bits, make_state

pytestmark = pytest.mark.gpu

ACC_TOL = 2e-5
POS_TOL_10 = 2e-6
VEL_TOL_10 = 2e-5


def run_gpu(nb, state, steps, g=G, e=E, dt=DT, variant=None, cls=None, jsplit=None):
    sp = nb.SimParams(particle_num=len(state), g=g, e=e, dt=dt)
    sim = (cls or nb.NaiveSim).from_particles(sp, None, state)
    if variant is not None:
        sim.set_tuning("naive_variant", variant)
    if jsplit is not None:
        sim.set_tuning("naive_jsplit", jsplit)
    for _ in range(steps):
        sim.encode()
        sim.cleanup()
    sim.wait()
    out = nb.as_floats(sim.dest_particle_slice()).copy()
    sim.destroy()
    return out


def acc_err(a, ref64):
    return np.abs(a[:, 6:9] - ref64[:, 6:9]).max() / max(np.abs(ref64[:, 6:9]).max(), 1e-300)


def check_against_oracles(out, ref32, ref64, steps):
    assert np.isfinite(out).all()
    e_gpu, e_lit = acc_err(out, ref64), acc_err(ref32, ref64)
    assert e_gpu <= ACC_TOL, f"acc err {e_gpu:.3e}"
    assert e_gpu <= 4 * e_lit + 1e-6, f"gpu {e_gpu:.3e} vs literal fp32 {e_lit:.3e}"
    if steps == 1:
        assert np.array_equal(bits(out[:, 0:3]), bits(ref32[:, 0:3]))
    assert np.abs(out[:, 0:3] - ref64[:, 0:3]).max() <= POS_TOL_10
    vs = max(np.abs(ref64[:, 3:6]).max(), 1e-300)
    assert np.abs(out[:, 3:6] - ref64[:, 3:6]).max() / vs <= VEL_TOL_10
    assert np.array_equal(out[:, 9], ref32[:, 9])
    m = ref64[:, 9]
    ke = lambda s: (0.5 * m * (s[:, 3:6].astype(np.float64) ** 2).sum(1)).sum()
    assert abs(ke(out) - ke(ref64)) <= 2e-5 * ke(ref64) + 1e-300
    mom = lambda s: (m[:, None] * s[:, 3:6].astype(np.float64)).sum(0)
    assert np.abs(mom(out) - mom(ref64)).max() <= 2e-5 * np.abs(m[:, None] * ref64[:, 3:6]).sum()


def test_gpu_present_and_library_is_the_hip_one(gpu):
    """... and it is the library built from THESE sources: libnbody_hip.so is git-ignored and travels to
    the GPU box as a binary, so the box has to prove which one it loaded (nb_version() ends with a hash
    of the sources it was compiled from; tests/test_abi.py checks the same thing without a GPU)."""
    from wgpu_n_body_amd.build import source_hash
    assert gpu.device_count() >= 1
    assert "gfx950" in gpu.version()
    assert gpu.version().endswith("src:" + source_hash()), (gpu.version(), source_hash())


def test_kat1_two_bodies(gpu, oracle):
    s = np.zeros((2, 10), np.float32)
    s[1, 0] = 1.0
    s[:, 9] = 1.0
    out = run_gpu(gpu, s, 1)
    ref = oracle.naive_step_f32(s, G, E, DT)
    assert np.array_equal(out[:, 0:3], s[:, 0:3])
    assert out[0, 6] == pytest.approx(1.59984008e-08, rel=3e-7)   # 0x32896cdb +- 2 ulp
    assert out[0, 3] == pytest.approx(1.27987218e-10, rel=3e-7)
    assert np.array_equal(out[1, 3:9], -out[0, 3:9])
    assert np.allclose(out, ref, rtol=3e-7, atol=0)
    out2 = run_gpu(gpu, s, 2)
    assert out2[0, 0] == pytest.approx(4.09559105e-12, rel=1e-6)  # KAT-2


def test_single_body_pure_drift_and_empty(gpu, oracle):
    s = np.array([[0.1, 0.2, 0.3, 1.0, -2.0, 0.5, 0.25, 0.5, -0.75, 3.0]], np.float32)
    assert np.array_equal(bits(run_gpu(gpu, s, 1)), bits(oracle.naive_step_f32(s, G, E, DT)))
    out = run_gpu(gpu, np.zeros((0, 10), np.float32), 2)
    assert out.shape == (0, 10)


def test_self_excluded_by_index_and_coincident_is_nan(gpu, oracle):
    s = np.zeros((2, 10), np.float32)
    s[0, 3] = 1.0
    s[1, 0] = 100.0
    s[:, 9] = 1.0
    out = run_gpu(gpu, s, 1)
    assert np.isfinite(out).all() and abs(out[0, 6]) < 1e-10
    assert np.allclose(out, oracle.naive_step_f32(s, G, E, DT), rtol=1e-6, atol=1e-30)
    c = np.zeros((3, 10), np.float32)
    c[2, 0] = 1.0
    c[:, 9] = 1.0
    out = run_gpu(gpu, c, 1)
    assert np.isnan(out[0, 6:9]).any() and np.isnan(out[1, 6:9]).any()  # as naive.wgsl:39


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "naive_*.npz"))))
def test_golden_fixtures(gpu, path):
    z = np.load(path)
    g, e, dt = (float(x) for x in z["params"])
    for steps in (1, 10):
        out = run_gpu(gpu, z["init"], steps, g, e, dt)
        check_against_oracles(out, z[f"f32_step{steps}"], z[f"f64_step{steps}"], steps)


@pytest.mark.parametrize("n", [2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 1000])
def test_ragged_sizes_against_oracle(gpu, oracle, n):
    s = make_state("spherical", n, 100 + n)
    out = run_gpu(gpu, s, 2)
    ref32 = oracle.naive_run_f32(s, G, E, DT, 2)
    ref64 = oracle.naive_run_f64(s, G, E, DT, 2)
    check_against_oracles(out, ref32, ref64, 2)


def test_every_kernel_variant_agrees(gpu, oracle):
    s = make_state("uniform", 1337, 42)
    ref32 = oracle.naive_run_f32(s, G, E, DT, 3)
    ref64 = oracle.naive_run_f64(s, G, E, DT, 3)
    for v, name in enumerate(gpu.naive_variants()):
        out = run_gpu(gpu, s, 3, variant=v)
        check_against_oracles(out, ref32, ref64, 3)


@pytest.mark.parametrize("n,variant,jsplit", [(4096, 3, 1), (4096, 3, 2), (4096, 3, 4),
                                              (5000, 1, 3), (5000, 2, 4), (2000, 0, 2),
                                              (8192, None, None), (700, 3, 8)])
def test_j_split_across_workgroups(gpu, oracle, n, variant, jsplit):
    """Few bodies per launch: the j range is split over JS workgroups per i-tile and a
    second kernel adds the partial sums in fixed order.  Same step, same tolerances."""
    s = make_state("uniform", n, 300 + n)
    out = run_gpu(gpu, s, 2, variant=variant, jsplit=jsplit)
    ref32 = oracle.naive_run_f32(s, G, E, DT, 2)
    ref64 = oracle.naive_run_f64(s, G, E, DT, 2)
    check_against_oracles(out, ref32, ref64, 2)
    again = run_gpu(gpu, s, 2, variant=variant, jsplit=jsplit)
    assert np.array_equal(bits(out), bits(again))   # deterministic (fixed summation order)


@pytest.mark.parametrize("kind,g,dt", [("uniform", G, DT), ("spherical", G, DT),
                                       ("disc", 0.00001, 0.0016)])
def test_medium_n_ten_steps(gpu, oracle, kind, g, dt):
    n = 4096
    s = make_state(kind, n, 77, g)
    out = run_gpu(gpu, s, 10, g=g, dt=dt)
    ref32 = oracle.naive_run_f32(s, g, E, dt, 10)
    ref64 = oracle.naive_run_f64(s, g, E, dt, 10)
    check_against_oracles(out, ref32, ref64, 10)


def test_runner_matches_simulator_and_write_read_roundtrip(gpu):
    nb = gpu
    sp = nb.SimParams(particle_num=777)
    init = nb.inits.spherical_init(sp, seed=5)
    runner = nb.OfflineHeadless(nb.NaiveSim, sp, nb.AddParams.NaiveSimParams(),
                                lambda p: nb.inits.spherical_init(p, seed=5))
    got_sp = runner.sim_params()
    assert got_sp.particle_num == 777 and np.float32(got_sp.g) == np.float32(sp.g) \
        and np.float32(got_sp.e) == np.float32(sp.e) and np.float32(got_sp.dt) == np.float32(sp.dt)
    assert np.array_equal(runner.read_particles(), init)      # state 0 == init, bit for bit
    runner.step()
    runner.step_n(2)
    a = runner.read_particles()
    assert runner.sim.step_num() == 3
    b = run_gpu(nb, nb.as_floats(init), 3)
    assert np.array_equal(bits(nb.as_floats(a)), bits(b))       # deterministic
    # checkpoint / restore: write back an earlier state and replay
    runner.sim.write_particles(init)
    runner.step_n(3)
    assert np.array_equal(runner.read_particles(), a)
    runner.destroy()


# ---- BASELINE.json full size (65,536 bodies): size-independent properties ---------------------

N_FULL = 65536


@pytest.fixture(scope="module")
def full_state():
    return make_state("uniform", N_FULL, 2)


@pytest.fixture(scope="module")
def full_one_step(gpu, full_state):
    return run_gpu(gpu, full_state, 1)


def test_full_size_sampled_bodies_against_oracle(gpu, oracle, full_state, full_one_step):
    """Oracle on three 128-body windows of the 64k problem (each is 128 x 65536 pairs)."""
    s64 = full_state.astype(np.float64)
    for lo in (0, 32768 - 64, N_FULL - 128):
        hi = lo + 128
        r32 = oracle.naive_step_f32(full_state, G, E, DT, lo, hi)[lo:hi]
        r64 = oracle.naive_step_f64(s64, G, E, DT, lo, hi)[lo:hi]
        got = full_one_step[lo:hi]
        assert np.array_equal(bits(got[:, 0:3]), bits(r32[:, 0:3]))
        scale = np.abs(r64[:, 6:9]).max()
        e_gpu = np.abs(got[:, 6:9] - r64[:, 6:9]).max() / scale
        e_lit = np.abs(r32[:, 6:9] - r64[:, 6:9]).max() / scale
        assert e_gpu <= ACC_TOL and e_gpu <= 4 * e_lit + 1e-6


def test_full_size_mirror_symmetry_is_exact(gpu, full_state, full_one_step):
    """x -> -x maps the step onto its mirror image exactly (negation is exact in fp32)."""
    m = full_state.copy()
    m[:, [0, 3, 6]] *= -1
    out = run_gpu(gpu, m, 1)
    out[:, [0, 3, 6]] *= -1
    assert np.array_equal(bits(out), bits(full_one_step))


def test_full_size_force_is_linear_in_g_and_mass(gpu, full_state, full_one_step):
    """acc = (g dt) * sum(m_j ...): doubling g, or every mass, doubles acc bit for bit."""
    a2 = run_gpu(gpu, full_state, 1, g=2 * G)
    assert np.array_equal(bits(a2[:, 6:9]), bits(2 * full_one_step[:, 6:9]))
    assert np.array_equal(bits(a2[:, 0:3]), bits(full_one_step[:, 0:3]))
    m2 = full_state.copy()
    m2[:, 9] *= 2
    b2 = run_gpu(gpu, m2, 1)
    assert np.array_equal(bits(b2[:, 6:9]), bits(2 * full_one_step[:, 6:9]))


def test_full_size_permutation_equivariance(gpu, full_state, full_one_step):
    perm = np.random.default_rng(0).permutation(N_FULL)
    out = run_gpu(gpu, full_state[perm], 1)
    scale = np.abs(full_one_step[:, 6:9]).max()
    assert np.array_equal(bits(out[:, 0:3]), bits(full_one_step[perm, 0:3]))
    assert np.abs(out[:, 6:9] - full_one_step[perm, 6:9]).max() / scale <= ACC_TOL


def test_full_size_momentum_change_is_small(gpu, full_state, full_one_step):
    """The update is not exactly momentum conserving (new x_i vs old x_j), but
    total momentum moves by far less than the sum of the individual impulses."""
    m = full_state[:, 9:10].astype(np.float64)
    dp = (m * (full_one_step[:, 3:6].astype(np.float64) - full_state[:, 3:6])).sum(0)
    impulses = np.abs(m * (full_one_step[:, 3:6].astype(np.float64) - full_state[:, 3:6])).sum()
    assert np.abs(dp).max() < 1e-3 * impulses


# ---- body-range sharding on one GPU (the multi-GPU data path without RCCL) ----------------------

def _hip():
    for name in ("libamdhip64.so", "libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    raise RuntimeError("libamdhip64 not found")


@pytest.mark.parametrize("n,world,jsplit", [(1000, 2, 1), (5000, 3, 1), (4096, 4, 1),
                                            (4096, 2, 4)])
def test_sharded_ranks_reproduce_the_single_simulator(gpu, n, world, jsplit):
    """`world` simulators, each owning a body range, exchanging position slices by
    device-to-device copy (what the RCCL all-gather does across GPUs) == one simulator."""
    nb = gpu
    hip = _hip()
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    sp = nb.SimParams(particle_num=n)
    init = nb.inits.uniform_init(sp, seed=9)
    single = nb.NaiveSim.from_particles(sp, None, init)
    sims = [nb.NaiveSim.from_particles(sp, None, init, nb.Placement(0, r, world))
            for r in range(world)]
    for s in [single] + sims:   # same tiling everywhere => bitwise-equal sums
        s.set_tuning("naive_variant", 1)
        s.set_tuning("naive_jsplit", jsplit)
    for _ in range(3):
        single.encode()
        for s in sims:
            s.encode()
        regions = []
        for s in sims:
            s.wait()
            regions.append(s.exchange_region())
        for src_rank, (sp_ptr, off, ln, tot) in enumerate(regions):   # all-gather
            for dst_rank, (dp_ptr, _o, _l, _t) in enumerate(regions):
                if dst_rank != src_rank:
                    assert hip.hipMemcpy(dp_ptr + off, sp_ptr + off, ln, 3) == 0  # DtoD
        # (a device-to-device hipMemcpy may return before the copy has run, and the simulators' own streams do
        # not wait for the null stream: without this the next step could read a slice that is still on its way --
        # seen once in ~10 runs of the whole suite as a mismatch of the last step's accelerations)
        assert hip.hipDeviceSynchronize() == 0
    want = nb.as_floats(single.dest_particle_slice())
    per = nb.shard_bodies_per_rank(n, world)
    for r, s in enumerate(sims):
        got = nb.as_floats(s.dest_particle_slice())
        lo, hi = min(n, r * per), min(n, (r + 1) * per)
        assert np.array_equal(bits(got[:, [0, 1, 2, 9]]), bits(want[:, [0, 1, 2, 9]]))  # all pos
        assert np.array_equal(bits(got[lo:hi]), bits(want[lo:hi]))                      # own v, a
        assert not got[:lo, 3:9].any() and not got[hi:, 3:9].any()
        s.destroy()
    single.destroy()


@pytest.mark.parametrize("mode,world", [("naive", 2), ("naive-overlap", 2), ("naive-overlap", 4)])
def test_sharded_naive_sim_two_ranks_one_gpu(gpu, oracle, tmp_path, mode, world):
    """The product's multi-GPU class (ShardedNaiveSim: torch-owned position buffers, kernels on
    torch's stream, in-place all_gather_into_tensor) run as 2 processes sharing this one GPU,
    gloo standing in for RCCL.  Without overlap the step is the single simulator's, bit for bit;
    with overlap (own-tiles half enqueued beside the exchange, other tiles after it) the j sum
    is split differently, so it is checked against the oracle's tolerances instead."""
    import socket
    import subprocess
    import sys
    from tests.helpers import ROOT
    nb = gpu
    n, steps = 3000, 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    from tests.helpers import run_workers
    run_workers([[sys.executable, os.path.join(ROOT, "tests", "_gpu_shard_worker.py"), str(tmp_path), str(n),
                  str(steps), mode]] * world, port, tmp_path)
    sp = nb.SimParams(particle_num=n)
    init = nb.inits.uniform_init(sp, seed=77)
    single = nb.NaiveSim.from_particles(sp, None, init)
    single.set_tuning("naive_variant", 1)
    single.set_tuning("naive_jsplit", 1)
    for _ in range(steps):
        single.encode()
    want = nb.as_floats(single.dest_particle_slice())
    single.destroy()
    ranks = [np.load(os.path.join(tmp_path, f"gpu_rank{r}.npz")) for r in range(world)]
    if mode == "naive":
        for z in ranks:
            lo, hi = int(z["lo"]), int(z["hi"])
            assert np.array_equal(bits(z["state"][:, [0, 1, 2, 9]]), bits(want[:, [0, 1, 2, 9]]))
            assert np.array_equal(bits(z["state"][lo:hi]), bits(want[lo:hi]))
    else:
        merged = np.zeros_like(want)
        for z in ranks:
            lo, hi = int(z["lo"]), int(z["hi"])
            merged[lo:hi] = z["state"][lo:hi]
            # every rank holds every position, identical across ranks
            assert np.array_equal(bits(z["state"][:, [0, 1, 2, 9]]),
                                  bits(ranks[0]["state"][:, [0, 1, 2, 9]]))
        s0 = nb.as_floats(init)
        check_against_oracles(merged, oracle.naive_run_f32(s0, G, E, DT, steps),
                              oracle.naive_run_f64(s0, G, E, DT, steps), steps)


def test_two_phase_step_on_one_rank_of_many(gpu, oracle):
    """nb_sim_encode_phase: phase 0 (own tiles) + phase 1 (other tiles + integrate) == one step,
    within the oracle's tolerances, for a rank that owns the middle third of the bodies."""
    nb = gpu
    n = 3000
    s = make_state("spherical", n, 88)
    sp = nb.SimParams(particle_num=n)
    sim = nb.NaiveSim.from_particles(sp, None, s, nb.Placement(0, 1, 3))
    sim.encode_phase(0)
    sim.encode_phase(1)
    sim.wait()
    assert sim.step_num() == 1
    got = nb.as_floats(sim.dest_particle_slice())
    per = nb.shard_bodies_per_rank(n, 3)
    lo, hi = per, min(n, 2 * per)
    ref32 = oracle.naive_step_f32(s, G, E, DT, lo, hi)
    ref64 = oracle.naive_step_f64(s.astype(np.float64), G, E, DT, lo, hi)
    check_against_oracles(got[lo:hi], ref32[lo:hi], ref64[lo:hi], 1)
    # encode() after a phase 0 completes that step; a second phase 0 in a row is an error
    sim.encode_phase(0)
    with pytest.raises(nb.NBodyError):
        sim.encode_phase(0)
    sim.encode()
    assert sim.step_num() == 2
    sim.destroy()


def test_random_sizes_variants_and_splits_against_oracle(gpu, oracle):
    """Fuzz: random body counts (ragged tiles, tails, tiny N), every kernel variant and forced
    j-splits, all three inits -- each checked against the fp32 and fp64 oracles."""
    nb = gpu
    rng = np.random.default_rng(2024)
    nvar = len(nb.naive_variants())
    for it in range(36):
        n = int(rng.choice([rng.integers(1, 70), rng.integers(70, 700), rng.integers(700, 6000)]))
        kind = ["uniform", "spherical", "disc"][it % 3]
        g, dt = (0.00001, 0.0016) if kind == "disc" else (G, DT)
        variant = int(rng.integers(0, nvar))
        jsplit = int(rng.choice([0, 1, 2, 3, 5, 8]))
        s = make_state(kind, n, 4000 + it, g)
        out = run_gpu(nb, s, 2, g=g, dt=dt, variant=variant, jsplit=jsplit)
        ref32 = oracle.naive_run_f32(s, g, E, dt, 2)
        ref64 = oracle.naive_run_f64(s, g, E, dt, 2)
        try:
            check_against_oracles(out, ref32, ref64, 2)
        except AssertionError as ex:
            raise AssertionError(f"n={n} kind={kind} variant={variant} jsplit={jsplit}: {ex}") from ex
