"""Host-side cost of one sharded all-pairs step (the floor the N=8 strong-scaling point sits on).

Runs ShardedNaiveSim on ONE GPU under the RCCL backend with world_size 1 and the collective
forced on, with so few bodies that the kernels are shorter than the host work: the measured
time per step is then the Python + ctypes + torch.distributed enqueue cost, which has to stay
below the per-rank kernel time (164 us at 8 ranks x 8192 bodies) for the
pipelined loop to keep the GPU busy.

    python tools/host_overhead.py [--bodies 2048] [--steps 3000]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=3000)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import wgpu_n_body_amd as nb
    from wgpu_n_body_amd.sharded import ShardedNaiveSim

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))

    sp = nb.SimParams(particle_num=args.bodies, g=1e-6, e=1e-4, dt=0.016)
    init = nb.inits.uniform_init(sp, seed=2)
    for label, overlap, force in (("one launch, no collective", False, False),
                                  ("two-phase, no collective", True, False),
                                  ("one launch + all-gather", False, True),
                                  ("two-phase + all-gather (the N>1 loop)", True, True)):
        sim = ShardedNaiveSim(sp, init, 0, 1, 0, overlap=overlap)
        sim.force_exchange = force
        for _ in range(200):
            sim.encode()
        sim.wait()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sim.encode()
        t_enq = time.perf_counter() - t0
        sim.wait()
        t_all = time.perf_counter() - t0
        print(f"{label:40s} enqueue {t_enq / args.steps * 1e6:7.1f} us/step   "
              f"enqueue+drain {t_all / args.steps * 1e6:7.1f} us/step", flush=True)
        sim.destroy()

    # The LET step (Barnes-Hut, one process per GPU) on one rank with its collectives forced on: the
    # host cost of the protocol with the export counts read on the host every step, and without
    # (fixed-stride imports, counts consumed on the device) -- RCCL all-gathers of regions 0 and 1
    # and an all-to-all of zero-length device views either way.
    from wgpu_n_body_amd.sharded import LetTreeSim
    LetTreeSim.force_exchange = True
    spt = nb.SimParams(particle_num=4096, g=1e-6, e=1e-4, dt=0.016)
    initt = nb.inits.uniform_init(spt, seed=2)
    for label, asyn in (("LET step, counts read on the host", False), ("LET step, no host read", True)):
        let = LetTreeSim(spt, 0.5, initt, 0, 1, 0, migrate_every=0, async_exchange=asyn)
        for _ in range(50):
            let.encode()
        let.wait()
        steps = 500
        before = let.host_syncs
        t0 = time.perf_counter()
        for _ in range(steps):
            let.encode()
        t_enq = time.perf_counter() - t0
        let.wait()
        t_all = time.perf_counter() - t0
        print(f"{label:40s} enqueue {t_enq / steps * 1e6:7.1f} us/step   enqueue+drain {t_all / steps * 1e6:7.1f} us/step"
              f"   host reads per step {(let.host_syncs - before) / steps:.2f}", flush=True)
        let.destroy()
    dist.destroy_process_group()

    # The one-process runner of the C ABI (nb_runner_create_multi): 8 ranks on this one GPU, 2,048
    # bodies -- kernels far shorter than the host work, so this is the library's own cost per step
    # (8 host threads: two launches, 7 event waits, one record each, one host barrier)
    sp8 = nb.SimParams(particle_num=2048, g=1e-6, e=1e-4, dt=0.016)
    runner = nb.OfflineHeadless(nb.NaiveSim, sp8, None, lambda p: nb.inits.uniform_init(p, seed=2), device_ids=[0] * 8)
    runner.step_n(200)
    t0 = time.perf_counter()
    runner.step_n(2000)
    t = time.perf_counter() - t0
    print(f"{'nb_runner_create_multi, 8 ranks, step_n':40s} {t / 2000 * 1e6:7.1f} us/step", flush=True)
    runner.destroy()

    # Barnes-Hut through the same runner (replicated tree, partitioned walk, peer copies): `world` ranks
    # on this ONE GPU -- the builds are done `world` times on the same device, so the step time here
    # is an upper bound of the host protocol's cost, not a scaling figure
    for n, world in ((8192, 1), (8192, 8), (1 << 20, 1), (1 << 20, 2), (1 << 20, 8)):
        spn = nb.SimParams(particle_num=n, g=1e-6, e=1e-4, dt=0.016)
        runner = nb.OfflineHeadless(nb.TreeSim, spn, nb.AddParams.TreeSimParams(0.5),
                                    lambda p: nb.inits.uniform_init(p, seed=3), device_ids=[0] * world)
        runner.step_n(30)
        t0 = time.perf_counter()
        runner.step_n(100)
        t = time.perf_counter() - t0
        print(f"{'nb_runner_create_multi tree, n=%d, %d rank(s)' % (n, world):52s} {t / 100 * 1e6:8.1f} us/step", flush=True)
        runner.destroy()

    # ... and with the build sharded too (nb_runner_create_multi_let: Morton domains + LET exchange hosted
    # in the library; migration every 8th step).  Same caveat: all ranks share this one GPU.
    for n, world in ((8192, 8), (1 << 20, 2), (1 << 20, 8), (1 << 22, 8)):
        spn = nb.SimParams(particle_num=n, g=1e-6, e=1e-4, dt=0.016)
        runner = nb.OfflineHeadless(nb.TreeSim, spn, nb.AddParams.TreeSimParams(0.5),
                                    lambda p: nb.inits.uniform_init(p, seed=3), device_ids=[0] * world,
                                    let_migrate_every=8)
        runner.step_n(24)
        t0 = time.perf_counter()
        runner.step_n(64)
        t = time.perf_counter() - t0
        print(f"{'nb_runner_create_multi_let, n=%d, %d ranks' % (n, world):52s} {t / 64 * 1e6:8.1f} us/step", flush=True)
        runner.destroy()


if __name__ == "__main__":
    main()
