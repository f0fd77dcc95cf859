"""The one-process several-GPU runners (nb_runner_create_multi, nb_runner_create_multi_let) on DISTINCT
devices -- `-m gpu`, skipped on a box with one GPU.

Everywhere else in the suite the ranks of these runners share device 0 (`device_ids=[0] * world`: the same
code minus the links), which never runs hipDeviceEnablePeerAccess, a store over xGMI, the visibility of such a
store behind a cross-device hipStreamWaitEvent, or hipMemcpyPeerAsync between two devices.  The first box
with two GPUs runs these tests by itself: all three schemes on device_ids = 0 .. world-1, each against the
SAME runner with every rank on device 0 -- whose parity with the oracle / the single simulator / the
Python-hosted protocol the rest of the suite holds -- bit for bit, and the all-pairs one against the oracle
directly.  Until then the distinct-device path is unexecuted (DESIGN.md section 5a says so).

The two tests at the end need no second GPU: the create-time peer-store check and the per-rank timing marks
run with ranks sharing a device too.
"""
import numpy as np
import pytest

from tests.helpers import DT, E, G, bits

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ids(gpu):
    n = gpu.device_count()
    if n < 2:
        pytest.skip("needs >= 2 HIP devices (this box has one): the distinct-device path stays unexecuted here")
    return list(range(min(n, 8)))


def test_all_pairs_on_distinct_devices(gpu, oracle, ids):
    nb = gpu
    n = 65536                                                  # BASELINE configs[1], the headline size
    sp = nb.SimParams(particle_num=n)
    init = nb.inits.uniform_init(sp, seed=2)
    multi = nb.OfflineHeadless(nb.NaiveSim, sp, None, lambda _p: init, device_ids=ids)
    same = nb.OfflineHeadless(nb.NaiveSim, sp, None, lambda _p: init, device_ids=[0] * len(ids))
    multi.step(); same.step()
    a1, b1 = nb.as_floats(multi.read_particles()), nb.as_floats(same.read_particles())
    multi.step_n(4); same.step_n(4)
    a5, b5 = nb.as_floats(multi.read_particles()), nb.as_floats(same.read_particles())
    multi.destroy(); same.destroy()
    # the same kernels on the same slices in the same order: every bit, after 1 and after 5 steps
    assert np.array_equal(bits(a1), bits(b1)) and np.array_equal(bits(a5), bits(b5))
    s0 = nb.as_floats(init)
    for lo in (0, n // 2 - 64, n - 128):                       # windows across rank borders, literal-fp32 oracle
        ref = oracle.naive_step_f32(s0, G, E, DT, lo, lo + 128)[lo:lo + 128]
        assert np.array_equal(bits(a1[lo:lo + 128, 0:3]), bits(ref[:, 0:3]))
        assert np.abs(a1[lo:lo + 128, 6:9] - ref[:, 6:9]).max() <= 2e-5 * np.abs(ref[:, 6:9]).max()


def test_replicated_tree_on_distinct_devices(gpu, ids):
    nb = gpu
    n = 1 << 17
    sp = nb.SimParams(particle_num=n)
    init = nb.inits.uniform_init(sp, seed=17)
    add = nb.AddParams.TreeSimParams(0.5)
    multi = nb.OfflineHeadless(nb.TreeSim, sp, add, lambda _p: init, device_ids=ids)
    one = nb.OfflineHeadless(nb.TreeSim, sp, add, lambda _p: init)
    multi.step(); one.step()
    multi.step_n(5); one.step_n(5)
    a, b = nb.as_floats(multi.read_particles()), nb.as_floats(one.read_particles())
    multi.destroy(); one.destroy()
    assert np.isfinite(b).all() and np.array_equal(bits(a), bits(b))   # the single TreeSim, bit for bit


@pytest.mark.parametrize("migrate", [0, 3])
def test_let_runner_on_distinct_devices(gpu, ids, migrate):
    nb = gpu
    n = 200000
    sp = nb.SimParams(particle_num=n)
    init = nb.inits.uniform_init(sp, seed=23)
    add = nb.AddParams.TreeSimParams(0.5)
    runs = []
    for dev in (ids, [0] * len(ids)):
        r = nb.OfflineHeadless(nb.TreeSim, sp, add, lambda _p: init, device_ids=dev, let_migrate_every=migrate)
        r.step_n(7)
        runs.append(nb.as_floats(r.read_particles()))
        r.destroy()
    assert np.isfinite(runs[0]).all() and np.array_equal(bits(runs[0]), bits(runs[1]))


def test_create_time_peer_store_check_and_rank_times(gpu):
    """Ranks sharing device 0: DeviceGroup::create runs its peer-store rehearsal (a kernel of every rank stores a
    word into every peer's table, the peers read them behind the events) and the per-rank timing marks split a
    batch of steps into the rank's own kernels and its waits for the peers."""
    nb = gpu
    world, n = 4, 32768
    sp = nb.SimParams(particle_num=n)
    init = nb.inits.uniform_init(sp, seed=5)
    r = nb.OfflineHeadless(nb.NaiveSim, sp, None, lambda _p: init, device_ids=[0] * world)
    r.step_n(3)                                      # unprofiled: nothing recorded
    k0, w0 = r.rank_times(world)
    assert k0 == [0.0] * world and w0 == [0.0] * world
    r.set_profiling(True)
    r.step_n(10)
    k, w = r.rank_times(world)
    r.set_profiling(False)
    got = nb.as_floats(r.read_particles())
    r.destroy()
    assert np.isfinite(got).all()
    assert all(0.01 < x < 100.0 for x in k), k       # 10 steps of 8,192 x 32,768 pairs per rank: ~0.5 ms of kernels
    assert all(0.0 <= x < 100.0 for x in w), w


def test_rank_times_of_the_tree_runners(gpu):
    nb = gpu
    world, n = 3, 60000
    sp = nb.SimParams(particle_num=n)
    init = nb.inits.uniform_init(sp, seed=6)
    add = nb.AddParams.TreeSimParams(0.5)
    for kw in ({}, {"let_migrate_every": 2}):
        r = nb.OfflineHeadless(nb.TreeSim, sp, add, lambda _p: init, device_ids=[0] * world, **kw)
        r.set_profiling(True)
        r.step_n(5)
        k, w = r.rank_times(world)
        one = nb.OfflineHeadless(nb.TreeSim, sp, add, lambda _p: init)   # a one-device runner: kernel_ms[0] only
        one.set_profiling(True)
        one.step_n(5)
        k1, w1 = one.rank_times(1)
        r.destroy(); one.destroy()
        assert all(0.01 < x < 1000.0 for x in k) and all(0.0 <= x < 1000.0 for x in w), (kw, k, w)
        assert 0.01 < k1[0] < 1000.0 and w1[0] == 0.0
