"""Multi-GPU Barnes-Hut with locally essential trees (LET): the device side of the protocol
(nb_sim_encode_phase(NB_PHASE_LET_*), nb_sim_let_set_imports) driven here by `world` simulators
sharing one GPU, the all-gathers and the all-to-all done with device-to-device copies -- what
RCCL does across GPUs -- plus the product class LetTreeSim run as separate processes.

What a body feels under LET is the sum of per-domain Barnes-Hut walks.  That is not the single
octree's approximation (cells are cut along domain borders), so against the single TreeSim the
comparison is at the level of the method's own error; the exact statements tested are:
  * world = 1: the protocol is the single TreeSim, bit for bit;
  * pruning is decision-exact: exporting whole octrees instead of LETs changes no bit;
  * theta -> 0: all-pairs (oracle), to fp32 summation tolerance;
  * the force error against all-pairs is that of the single tree."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.helpers import ROOT, bits, run_workers

pytestmark = pytest.mark.gpu

META, BUILD, WALK, MIGRATE, WALK_OWN = 2, 3, 4, 5, 6
REC = 32  # bytes per exported record


def _hip():
    for name in ("libamdhip64.so", "libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so"):
        try:
            lib = C.CDLL(name)
            lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            return lib
        except OSError:
            continue
    raise RuntimeError("libamdhip64 not found")


def tagged(nb, n, seed, init="uniform"):
    """Seeded particles whose masses are distinct, so bodies can be matched after the sims have
    re-ordered them (every tree step leaves the bodies in tree order, like the reference)."""
    sp = nb.SimParams(particle_num=n)
    p = getattr(nb.inits, init + "_init")(sp, seed=seed).copy()
    f = nb.as_floats(p)
    f[:, 9] = 1.0 + np.arange(n, dtype=np.float32) / np.float32(2 * n)
    return sp, p


def by_tag(nb, particles):
    f = nb.as_floats(particles)
    return f[np.argsort(f[:, 9], kind="stable")]


class LetGroup:
    """`world` TreeSims on one GPU running the LET protocol, exchanges by hipMemcpy."""

    def __init__(self, nb, sp, particles, world, theta, prune=True, cap=None, migrate_every=0,
                 own_first=False, fixed_stride=None, export_mode=None):
        self.export_mode = export_mode     # tree_let_export_mode: 1 one launch (default), 0 a launch per level
        self.nb, self.world, self.hip = nb, world, _hip()
        self.fixed_stride = fixed_stride   # records per peer moved blindly; counts stay on the device
        self.migrate_every, self.steps_done, self.own_first = migrate_every, 0, own_first
        self.sp, self.theta, self.prune, self.cap = sp, theta, prune, cap
        self.sims = []
        self.counts = None
        self.migrated = 0
        self._adopt(particles)

    def _adopt(self, particles):
        from wgpu_n_body_amd.sharded import morton_domains
        nb, world, sp, theta, prune, cap = self.nb, self.world, self.sp, self.theta, self.prune, self.cap
        for s in self.sims:
            s.destroy()
        order, cuts, splits, ref_bound = morton_domains(particles, world, with_owners=True)
        counts = [cuts[r + 1] - cuts[r] for r in range(world)]
        capacity = int(1.25 * max(counts)) + 4096
        self.mig_cap = max(1024, capacity // 8)
        self.sims = []
        for r in range(world):
            mine = particles[order[cuts[r]:cuts[r + 1]]]
            padded = np.zeros(capacity, dtype=mine.dtype)
            padded[:len(mine)] = mine
            spl = nb.SimParams(particle_num=capacity, g=sp.g, e=sp.e, dt=sp.dt)
            s = nb.TreeSim.from_particles(spl, nb.AddParams.TreeSimParams(theta), padded)
            s.set_tuning("tree_let_world", world)
            s.set_tuning("tree_let_rank", r)
            s.set_tuning("tree_let_active", len(mine))
            s.set_tuning("tree_let_prune", 1 if prune else 0)
            if self.export_mode is not None:
                s.set_tuning("tree_let_export_mode", self.export_mode)
            s.set_tuning("tree_let_cap", cap or (2 * capacity + 64))
            s.let_set_owners(splits, ref_bound, self.mig_cap)
            self.sims.append(s)

    def rebalance(self):
        """what LetTreeSim.rebalance does: all bodies, rank by rank, re-cut into new domains"""
        self._adopt(self.particles().copy())

    def _copy(self, dst, src, nbytes):
        if nbytes:
            assert self.hip.hipMemcpy(dst, src, nbytes, 3) == 0  # device to device
            # (it may return before the copy has run, and the simulators' streams do not wait for the null stream)
            assert self.hip.hipDeviceSynchronize() == 0

    def _all_gather(self, k):
        regs = []
        for s in self.sims:
            s.wait()
            regs.append(s.exchange_region(k))
        for src, (sp_, off, ln, _t) in enumerate(regs):
            for dst, (dp_, _o, _l, _t2) in enumerate(regs):
                if dst != src:
                    self._copy(dp_ + off, sp_ + off, ln)

    def _matrix(self, k):
        """all-gather region k, then the counts matrix as rank 0's host reads it"""
        W = self.world
        self._all_gather(k)
        ptr, _o, _l, tot = self.sims[0].exchange_region(k)
        host = np.zeros(W * W, dtype=np.uint32)
        assert self.hip.hipMemcpy(host.ctypes.data, ptr, tot, 2) == 0
        return host.reshape(W, W).astype(np.int64)

    def _all_to_all(self, counts, k_send, k_recv, rec_bytes):
        """segment `me` of every peer's region k_send -> region k_recv of `me`, packed in rank order"""
        W = self.world
        received = []
        for me, s in enumerate(self.sims):
            rptr = s.exchange_region(k_recv)[0]
            offs, recv = 0, []
            for r in range(W):
                c = 0 if r == me else int(counts[r, me])
                recv.append(c)
                if c:
                    sptr, _o, seg, _t = self.sims[r].exchange_region(k_send)
                    self._copy(rptr + offs * rec_bytes, sptr + me * seg, c * rec_bytes)
                offs += c
            received.append(recv)
        return received

    def migrate(self):
        for s in self.sims:
            s.encode_phase(MIGRATE)
        counts = self._matrix(4)
        received = self._all_to_all(counts, 5, 6, 48)
        for me, s in enumerate(self.sims):
            s.let_set_arrivals(int(counts[me, me]), received[me])
        self.migrated += int(counts.sum() - np.trace(counts))
        return counts

    def step(self):
        if self.migrate_every and self.steps_done and self.steps_done % self.migrate_every == 0:
            self.migrate()
        for s in self.sims:
            s.encode_phase(META)
        self._all_gather(0)
        for s in self.sims:
            s.encode_phase(BUILD)
            if self.own_first:
                s.encode_phase(WALK_OWN)      # while the exchange below is "in flight"
        if self.fixed_stride:
            # nb_sim_let_set_import_stride: all-gather the counts (device to device, never read on
            # the host), move a fixed number of records per peer
            self._all_gather(1)
            F = self.fixed_stride
            for me, s in enumerate(self.sims):
                rptr = s.exchange_region(3)[0]
                for r in range(self.world):
                    if r == me:
                        continue
                    j = r if r < me else r - 1
                    sptr, _o, seg, _t = self.sims[r].exchange_region(2)
                    self._copy(rptr + j * F * REC, sptr + me * seg, F * REC)
                s.let_set_import_stride(F)
        else:
            counts = self._matrix(1)
            self.counts = counts
            received = self._all_to_all(counts, 2, 3, REC)
            for me, s in enumerate(self.sims):
                s.let_set_imports(received[me])
        for s in self.sims:
            s.encode_phase(WALK)
        self.steps_done += 1

    def particles(self):
        return np.concatenate([s.dest_particle_slice() for s in self.sims])

    def destroy(self):
        for s in self.sims:
            s.destroy()


def test_let_protocol_with_one_rank_is_the_single_tree_sim(gpu):
    nb = gpu
    sp, p = tagged(nb, 5000, 21)
    single = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(0.5), p)
    grp = LetGroup(nb, sp, p, 1, 0.5)
    for _ in range(3):
        single.encode()
        grp.step()
    a, b = by_tag(nb, single.dest_particle_slice()), by_tag(nb, grp.particles())
    assert np.array_equal(bits(a), bits(b))
    single.destroy()
    grp.destroy()


@pytest.mark.parametrize("n,world,theta,init", [(6000, 2, 0.5, "uniform"), (9000, 3, 0.75, "uniform"),
                                                 (20000, 4, 0.5, "uniform"), (8000, 3, 0.6, "disc")])
def test_let_pruning_changes_no_bit(gpu, n, world, theta, init):
    """Exporting each rank's WHOLE octree to every peer and exporting only the locally essential
    part must give identical bodies: the pruning never alters a decision of the walk."""
    nb = gpu
    sp, p = tagged(nb, n, 22, init)
    pruned = LetGroup(nb, sp, p, world, theta, prune=True)
    whole = LetGroup(nb, sp, p, world, theta, prune=False)
    for _ in range(3):
        pruned.step()
        whole.step()
    a, b = by_tag(nb, pruned.particles()), by_tag(nb, whole.particles())
    assert np.isfinite(a).all()
    assert np.array_equal(bits(a), bits(b))
    off = ~np.eye(world, dtype=bool)
    assert (pruned.counts[off] <= whole.counts[off]).all()
    # ... and it does prune (least for the thin disc, whose Morton domains interleave)
    assert pruned.counts[off].sum() < (0.8 if init == "disc" else 0.7) * whole.counts[off].sum()
    pruned.destroy()
    whole.destroy()


@pytest.mark.parametrize("n,world,own_first", [(6000, 2, False), (9000, 3, True), (20000, 4, False), (3000, 1, False)])
def test_let_fixed_stride_imports_change_no_bit(gpu, n, world, own_first):
    """The hand-over without a host round trip (nb_sim_let_set_import_stride: a fixed number of
    records per peer, the real counts read on the device) walks the same trees in the same order
    as the packed hand-over (nb_sim_let_set_imports): identical bodies; too small a stride is
    reported, not walked."""
    nb = gpu
    sp, p = tagged(nb, n, 27)
    packed = LetGroup(nb, sp, p, world, 0.5, own_first=own_first, migrate_every=2)
    packed.step()
    need = int(packed.counts.max()) if world > 1 else 1
    fixed = LetGroup(nb, sp, p, world, 0.5, own_first=own_first, migrate_every=2, fixed_stride=need + need // 4 + 64)
    fixed.step()
    for _ in range(3):
        packed.step()
        fixed.step()
    a, b = by_tag(nb, packed.particles()), by_tag(nb, fixed.particles())
    assert np.isfinite(a).all()
    assert np.array_equal(bits(a), bits(b))
    packed.destroy()
    fixed.destroy()
    if world > 1:
        small = LetGroup(nb, sp, p, world, 0.5, fixed_stride=max(8, need // 4))
        with pytest.raises(nb.NBodyError) as ex:
            small.step()
            for s in small.sims:
                s.wait()
        assert "tree_let_cap" in str(ex.value) or "records" in str(ex.value)
        small.destroy()


def test_let_with_theta_to_zero_is_all_pairs(gpu, oracle):
    nb = gpu
    n, world = 3000, 3
    sp, p = tagged(nb, n, 23)
    grp = LetGroup(nb, sp, p, world, 1e-4)
    grp.step()
    got = by_tag(nb, grp.particles())
    want = oracle.naive_step_f64(nb.as_floats(p), sp.g, sp.e, sp.dt)
    want = want[np.argsort(want[:, 9], kind="stable")]
    scale = np.abs(want[:, 6:9]).max()
    assert np.abs(got[:, 6:9] - want[:, 6:9]).max() <= 2e-5 * scale
    assert np.abs(got[:, 0:3] - want[:, 0:3]).max() <= 1e-6
    grp.destroy()


def test_let_force_error_is_that_of_the_single_tree(gpu, oracle):
    """Against exact all-pairs forces the sum of per-domain walks is as good as one octree."""
    nb = gpu
    n, theta = 16384, 0.5
    sp, p = tagged(nb, n, 24)
    exact = oracle.naive_step_f64(nb.as_floats(p), sp.g, sp.e, sp.dt)
    exact = exact[np.argsort(exact[:, 9], kind="stable")][:, 6:9]
    single = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(theta), p)
    single.encode()
    one = by_tag(nb, single.dest_particle_slice())[:, 6:9]
    single.destroy()
    norm = np.linalg.norm(exact, axis=1)
    err_one = np.median(np.linalg.norm(one - exact, axis=1) / norm)
    for world in (2, 4, 8):
        grp = LetGroup(nb, sp, p, world, theta)
        grp.step()
        let = by_tag(nb, grp.particles())[:, 6:9]
        grp.destroy()
        err_let = np.median(np.linalg.norm(let - exact, axis=1) / norm)
        assert err_let <= 1.25 * err_one + 1e-6, (world, err_let, err_one)
        assert np.median(np.linalg.norm(let - one, axis=1) / norm) <= 2.5 * err_one


def test_let_export_capacity_is_checked(gpu):
    nb = gpu
    sp, p = tagged(nb, 4000, 25)
    grp = LetGroup(nb, sp, p, 2, 0.3, cap=64)     # far too small for a neighbour's LET
    with pytest.raises(nb.NBodyError) as ex:      # reported by the first wait after the export
        grp.step()
        for s in grp.sims:
            s.dest_particle_slice()
    assert "tree_let_cap" in str(ex.value)
    grp.destroy()


def test_let_phases_must_run_in_order(gpu):
    nb = gpu
    sp, p = tagged(nb, 512, 26)
    s = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(0.5), p)
    with pytest.raises(nb.NBodyError):
        s.encode_phase(META)                      # LET not configured
    s.set_tuning("tree_let_world", 2)
    s.set_tuning("tree_let_rank", 0)
    s.set_tuning("tree_let_cap", 4096)
    with pytest.raises(nb.NBodyError):
        s.encode_phase(BUILD)                     # META first
    with pytest.raises(nb.NBodyError):
        s.encode()                                # the plain step is not available in LET mode
    s.encode_phase(META)
    s.encode_phase(BUILD)
    with pytest.raises(nb.NBodyError):
        s.encode_phase(WALK)                      # imports not declared
    s.let_set_imports([0, 0])
    s.encode_phase(WALK)
    s.wait()
    s.destroy()


def test_let_own_tree_first_is_the_same_step(gpu):
    """NB_PHASE_LET_WALK_OWN before the exchange + imports afterwards == one walk over all trees."""
    nb = gpu
    sp, p = tagged(nb, 12000, 33)
    whole = LetGroup(nb, sp, p, 3, 0.5, migrate_every=1)
    split = LetGroup(nb, sp, p, 3, 0.5, migrate_every=1, own_first=True)
    for _ in range(4):
        whole.step()
        split.step()
    assert np.array_equal(bits(by_tag(nb, whole.particles())), bits(by_tag(nb, split.particles())))
    s = split.sims[0]
    s.encode_phase(META)
    with pytest.raises(nb.NBodyError):
        s.encode_phase(WALK_OWN)              # only after the build
    whole.destroy()
    split.destroy()


@pytest.mark.parametrize("n,world", [(5, 4), (2, 4), (1, 3), (64, 8)])
def test_let_with_almost_empty_ranks(gpu, oracle, n, world):
    """Fewer bodies than ranks: some domains hold one body or none; theta -> 0 must still be all-pairs."""
    nb = gpu
    sp, p = moving(nb, n, 34, 0.5)
    grp = LetGroup(nb, sp, p, world, 1e-4, migrate_every=1)
    for _ in range(3):
        grp.step()
    got = by_tag(nb, grp.particles())
    want = oracle.naive_run_f64(nb.as_floats(p), sp.g, sp.e, sp.dt, 3)
    want = want[np.argsort(want[:, 9], kind="stable")]
    assert len(got) == n
    assert np.abs(got[:, 0:3] - want[:, 0:3]).max() <= 2e-6
    assert np.abs(got[:, 6:9] - want[:, 6:9]).max() <= 2e-5 * max(np.abs(want[:, 6:9]).max(), 1e-30)
    grp.destroy()


def moving(nb, n, seed, speed):
    """tagged() bodies with random velocities, so that some cross domain borders in a few steps"""
    sp, p = tagged(nb, n, seed)
    f = nb.as_floats(p)
    rng = np.random.default_rng(seed)
    f[:, 3:6] = rng.uniform(-speed, speed, size=(n, 3)).astype(np.float32)
    return sp, p


def test_let_migration_rehomes_bodies_and_stays_exact(gpu, oracle):
    """With theta -> 0 the LET step is all-pairs whatever the domains are, so a run in which
    bodies change owner every step must still track the fp64 all-pairs oracle."""
    nb = gpu
    n, world, steps = 3000, 4, 5
    sp, p = moving(nb, n, 31, 1.5)
    grp = LetGroup(nb, sp, p, world, 1e-4, migrate_every=1)
    for _ in range(steps):
        grp.step()
    assert grp.migrated > 20                         # bodies did change owner
    got = by_tag(nb, grp.particles())
    assert len(got) == n and np.array_equal(got[:, 9], np.sort(nb.as_floats(p)[:, 9]))   # nobody lost
    want = oracle.naive_run_f64(nb.as_floats(p), sp.g, sp.e, sp.dt, steps)
    want = want[np.argsort(want[:, 9], kind="stable")]
    assert np.abs(got[:, 0:3] - want[:, 0:3]).max() <= 2e-6
    assert np.abs(got[:, 6:9] - want[:, 6:9]).max() <= 2e-5 * np.abs(want[:, 6:9]).max()
    # after a migration every body sits in the key range of the rank that holds it: a second one
    # right behind it moves nobody
    grp.migrate()
    counts = grp.migrate()
    assert counts.sum() == n and (counts - np.diag(np.diag(counts))).sum() == 0
    grp.destroy()


def test_let_migration_keeps_the_waves_coherent(gpu):
    """Leavers sort to the ends of their old rank's tree order and make a few waves walk far more
    cells than the rest; handing them over keeps the longest walk of any wave bounded."""
    nb = gpu
    n, world, steps = 60000, 4, 6
    sp, p = moving(nb, n, 32, 0.3)
    # (walk mode 0: one wave walks for 64 bodies, where a handful of strays costs the whole wave)
    longest = {}
    for mode in (0, 1):
        for every in (0, 1):
            grp = LetGroup(nb, sp, p, world, 0.5, migrate_every=every)
            for s in grp.sims:
                s.set_tuning("tree_count_visits", 1)
                s.set_tuning("tree_walk_mode", mode)
                s.set_tuning("tree_walk_bpw", 64)     # full waves, as on a large problem
            for _ in range(steps):
                grp.step()
            longest[mode, every] = max(int(s.debug_buffer("counters", np.uint64)[5]) for s in grp.sims)
            grp.destroy()
    assert longest[0, 1] < 0.5 * longest[0, 0], longest
    # groups of 8 bodies (mode 1, the default) suffer less from strays, but still gain
    assert longest[1, 1] < 0.8 * longest[1, 0], longest


@pytest.mark.parametrize("world,mode", [(2, "let"), (3, "let"), (2, "let-overlap"), (3, "let-rebalance"),
                                        (3, "let-async")])
def test_let_tree_sim_processes_share_one_gpu(gpu, tmp_path, world, mode):
    """The product class (LetTreeSim: torch.distributed for the three exchanges) as `world`
    processes on this one GPU, gloo standing in for RCCL == the in-process emulation above."""
    import socket
    import subprocess
    import sys
    nb = gpu
    n, steps, theta = 6000, (9 if mode == "let-async" else 3), 0.5
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    run_workers([[sys.executable, os.path.join(ROOT, "tests", "_gpu_shard_worker.py"), str(tmp_path), str(n),
                  str(steps), mode]] * world, port, tmp_path)
    sp, p0 = tagged(nb, n, 27)
    if mode == "let-async":
        # steps 0 and 1 learn the counts synchronously; from then on only the steps that migrate
        # (every 4th) read anything of the current step on the host
        for r in range(world):
            syncs = np.load(os.path.join(tmp_path, f"syncs{r}.npy"))
            assert syncs[0] >= 1 and syncs[1] >= 1
            for k in range(2, steps):
                assert (syncs[k] == 0) == (k % 4 != 0), (r, k, syncs)
    grp = LetGroup(nb, sp, p0, world, theta, migrate_every=(4 if mode == "let-async" else 1))
    for k in range(steps):
        if mode == "let-rebalance" and k == 2:
            grp.rebalance()
        grp.step()
    want = by_tag(nb, grp.particles())
    grp.destroy()
    got = np.concatenate([np.load(os.path.join(tmp_path, f"rank{r}.npy")) for r in range(world)])
    got = got[np.argsort(got[:, 9], kind="stable")]
    assert np.array_equal(bits(got), bits(want))


def test_rccl_code_paths_on_one_rank(gpu, tmp_path):
    """RCCL refuses several ranks on one device, so on this one-GPU box the product's RCCL branches
    run as a 1-rank NCCL group with the collectives forced: all_gather_into_tensor on views of
    library memory (all-pairs, replicated tree, LET regions), dist.all_to_all on zero-length device
    views, and the LET step without host reads.  Every sharded class == its single simulator."""
    import json
    import socket
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    run_workers([[sys.executable, os.path.join(ROOT, "tests", "_gpu_shard_worker.py"), str(tmp_path), "5000", "3",
                  "rccl-world1"]], port, tmp_path)
    rep = json.load(open(os.path.join(tmp_path, "rccl_world1.json")))
    assert rep["naive"] and rep["tree"] and rep["let"], rep
    assert rep["let_async_used"]
    # the first two steps and the migrating ones read counts on the host; the others do not
    assert rep["let_syncs_per_step"][3] == 0 and rep["let_syncs_per_step"][5] == 0, rep


def test_energy_and_momentum_after_many_steps_track_all_pairs(gpu, oracle):
    """North-star check 'positions / energies after N steps': 20 steps of the single octree and of
    the 4-domain LET run against 20 steps of the fp64 all-pairs oracle, theta = 0.5: kinetic energy
    and total momentum within the method's error, positions within 1e-5 of the box size."""
    nb = gpu
    n, steps, theta = 4096, 20, 0.5
    sp, p = moving(nb, n, 35, 0.05)
    want = oracle.naive_run_f64(nb.as_floats(p), sp.g, sp.e, sp.dt, steps)
    want = want[np.argsort(want[:, 9], kind="stable")]
    single = nb.TreeSim.from_particles(sp, nb.AddParams.TreeSimParams(theta), p)
    grp = LetGroup(nb, sp, p, 4, theta, migrate_every=1)
    for _ in range(steps):
        single.encode()
        grp.step()
    m = want[:, 9]
    ke_want = 0.5 * (m * (want[:, 3:6] ** 2).sum(axis=1)).sum()
    mom_want = (m[:, None] * want[:, 3:6]).sum(axis=0)
    for name, got in (("single tree", by_tag(nb, single.dest_particle_slice())),
                      ("LET x4", by_tag(nb, grp.particles()))):
        got = got.astype(np.float64)
        ke = 0.5 * (got[:, 9] * (got[:, 3:6] ** 2).sum(axis=1)).sum()
        mom = (got[:, 9:10] * got[:, 3:6]).sum(axis=0)
        assert abs(ke - ke_want) <= 1e-5 * ke_want, (name, ke, ke_want)
        assert np.abs(mom - mom_want).max() <= 1e-5 * np.abs(m[:, None] * want[:, 3:6]).sum(), name
        assert np.abs(got[:, 0:3] - want[:, 0:3]).max() <= 2e-5, name
    single.destroy()
    grp.destroy()


@pytest.mark.parametrize("n,world,theta,init,prune", [(20000, 4, 0.5, "uniform", True), (60000, 8, 0.75, "disc", True),
                                                       (9000, 3, 0.5, "spherical", False), (700, 5, 0.5, "uniform", True)])
def test_let_export_in_one_launch_changes_no_bit(gpu, n, world, theta, init, prune):
    """let_export_kernel (one launch: a workgroup per peer and root child walks its subtree breadth-first)
    against let_export_level_kernel (a launch per tree level): the segments are laid out differently
    (the order of the allocations), the records a peer walks are the same -- identical bodies, and
    identical export counts up to the unused reserved slots of the root's children and grandchildren."""
    nb = gpu
    sp, p = tagged(nb, n, 23, init)
    one = LetGroup(nb, sp, p, world, theta, prune=prune, export_mode=1)
    per_level = LetGroup(nb, sp, p, world, theta, prune=prune, export_mode=0)
    for _ in range(3):
        one.step()
        per_level.step()
    a, b = by_tag(nb, one.particles()), by_tag(nb, per_level.particles())
    ca, cb = one.counts, per_level.counts
    one.destroy()
    per_level.destroy()
    assert np.isfinite(a).all() and np.array_equal(bits(a), bits(b))
    off = ~np.eye(world, dtype=bool)
    assert ((ca - cb)[off] >= 0).all() and ((ca - cb)[off] <= 72).all()


# ---- the same protocol hosted inside the library: nb_runner_create_multi_let ---------------------------

@pytest.mark.parametrize("n,world,theta,init,migrate", [(8000, 3, 0.5, "uniform", 0), (20000, 4, 0.5, "uniform", 2),
                                                         (30000, 8, 0.75, "uniform", 1), (9000, 3, 0.6, "disc", 0), (3000, 1, 0.5, "uniform", 2),
                                                         (5000, 2, 0.6, "uniform", 3)])
def test_native_let_runner_is_the_python_hosted_protocol_bit_for_bit(gpu, n, world, theta, init, migrate):
    """nb_runner_create_multi_let (C++ rank threads, bounds / counts / records stored into the peers through
    peer access, counts consumed on the device) against LetGroup above (the protocol driven from
    Python, exchanges by hipMemcpy, counts read on the host): the same domains, the same kernels,
    so every rank must hold the same bodies with the same bits in the same order, migration steps
    included.  The `world` ranks share this box's one GPU (the code path of `world` GPUs)."""
    nb = gpu
    sp, p = tagged(nb, n, 31 + world, init)
    if init == "disc":   # the disc workload of visualize.rs (its velocities are orbits for this g)
        sp = nb.SimParams(particle_num=n, g=0.00001, dt=0.0016)
    steps = 7
    grp = LetGroup(nb, sp, p, world, theta, migrate_every=migrate)
    native = nb.OfflineHeadless(nb.TreeSim, sp, nb.AddParams.TreeSimParams(theta), lambda _p: p,
                                device_ids=[0] * world, let_migrate_every=migrate)
    native.step()
    grp.step()
    native.step_n(steps - 1)
    for _ in range(steps - 1):
        grp.step()
    assert native.step_num() == steps
    a, b = nb.as_floats(native.read_particles()), nb.as_floats(grp.particles())
    native.destroy()
    moved = grp.migrated
    grp.destroy()
    assert np.isfinite(b).all() and len(a) == n
    assert np.array_equal(bits(a), bits(b))
    assert len(np.unique(a[:, 9])) == n                     # every body exactly once
    if migrate == 1:
        assert moved > 0                                    # the migration path did run


def test_native_let_runner_against_the_single_tree(gpu):
    """... and against ONE TreeSim over all bodies: per-domain walks instead of one walk -- the same
    physics within the walk's own error (as test_let_gpu's oracle comparisons), bodies matched by tag."""
    nb = gpu
    n, world = 40000, 4
    sp, p = tagged(nb, n, 77)
    one = nb.OfflineHeadless(nb.TreeSim, sp, nb.AddParams.TreeSimParams(0.5), lambda _p: p)
    let = nb.OfflineHeadless(nb.TreeSim, sp, nb.AddParams.TreeSimParams(0.5), lambda _p: p, device_ids=[0] * world,
                             let_migrate_every=2)
    one.step_n(5)
    let.step_n(5)
    a, b = by_tag(nb, let.read_particles()), by_tag(nb, one.read_particles())
    one.destroy()
    let.destroy()
    assert np.abs(a[:, 0:3] - b[:, 0:3]).max() < 1e-6
    err = np.linalg.norm(a[:, 6:9].astype(np.float64) - b[:, 6:9], axis=1) / np.linalg.norm(b[:, 6:9].astype(np.float64), axis=1)
    assert np.median(err) < 2e-2 and np.percentile(err, 99) < 0.2


def test_native_let_runner_reports_a_migration_that_overflows_its_segments(gpu):
    """A thin disc centred on the root's z = 0 border: a third of a rank's bodies change their top-level
    octant -- and with it their Morton domain -- within two steps, more than a migration segment
    (capacity / 8 bodies per destination) holds.  The step must fail with a message, not move a cut-off
    list (LetTreeSim raises the same way; the remedy is a rebalance)."""
    nb = gpu
    sp, p = tagged(nb, 9000, 34, "disc")
    r = nb.OfflineHeadless(nb.TreeSim, nb.SimParams(particle_num=9000, g=0.00001, dt=0.0016),
                           nb.AddParams.TreeSimParams(0.6), lambda _p: p, device_ids=[0, 0, 0], let_migrate_every=2)
    r.step_n(2)
    with pytest.raises(nb.NBodyError) as ei:
        r.step()
    assert "leavers" in str(ei.value)
    r.destroy()


def test_native_let_runner_argument_errors(gpu):
    nb = gpu
    sp = nb.SimParams(particle_num=64)
    with pytest.raises(nb.NBodyError):   # all-pairs has no LET scheme
        nb.OfflineHeadless(nb.NaiveSim, sp, None, lambda q: nb.inits.uniform_init(q), device_ids=[0, 0],
                           let_migrate_every=1)
