"""Barnes-Hut step on several GPUs (BASELINE.json configs[4]: 4,194,304 bodies, theta 0.5, 8 GPUs),
either scheme of SURVEY 8(e): `--scheme let` (default; Morton-range domains, local octrees, LET
exchange, body migration: LetTreeSim) or `--scheme replicated` (replicated tree + partitioned
walk: ShardedTreeSim).  Launch like bench.py:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P tools/bench_tree_multi.py --bodies 4194304 --steps 20 --warmup 5

Rank 0 prints one JSON line (ms/step = max over ranks, bodies/s over the whole job).
NB_DIST_BACKEND=gloo NB_BENCH_SAME_DEVICE=1 rehearse it with several ranks on one GPU."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--bodies", type=int, default=4194304)
ap.add_argument("--theta", type=float, default=0.5)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--no-overlap", action="store_true", help="replicated scheme: no pipelining")
ap.add_argument("--let-overlap", action="store_true", help="LET scheme: own-tree walk beside the exchange")
ap.add_argument("--scheme", choices=("let", "replicated"), default="let")
ap.add_argument("--migrate-every", type=int, default=1)
args = ap.parse_args()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import wgpu_n_body_amd as nb  # noqa: E402
from wgpu_n_body_amd.sharded import LetTreeSim, ShardedTreeSim  # noqa: E402

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local_rank = 0 if os.environ.get("NB_BENCH_SAME_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
backend = os.environ.get("NB_DIST_BACKEND", "nccl")
torch.cuda.set_device(local_rank)
if world > 1:
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)


def sync_all():
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()


sp = nb.SimParams(particle_num=args.bodies)
init = nb.inits.uniform_init(sp, seed=5)
if args.scheme == "let":
    sim = LetTreeSim(sp, args.theta, init, rank, world, local_rank, migrate_every=args.migrate_every,
                     overlap=args.let_overlap)
else:
    sim = ShardedTreeSim(sp, args.theta, init, rank, world, local_rank, overlap=not args.no_overlap)
for _ in range(args.warmup):
    sim.encode()
sim.wait()
sync_all()
t0 = time.perf_counter()
for _ in range(args.steps):
    sim.encode()
sim.wait()
sync_all()
wall = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
if world > 1:
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)
wall = float(wall.item())
state = nb.as_floats(sim.read_particles())
assert np.isfinite(state).all()
if rank == 0:
    print(json.dumps({"metric": "Barnes-Hut step, " + ("Morton-range domains + LET exchange" if args.scheme == "let"
                                                       else "replicated tree + partitioned walk"),
                      "bodies": args.bodies, "theta": args.theta, "n_gpus": world,
                      "ms_per_step": wall / args.steps * 1e3,
                      "bodies_per_s": args.bodies * args.steps / wall,
                      "overlap": (args.let_overlap if args.scheme == "let" else not args.no_overlap), "steps": args.steps, "warmup": args.warmup}))
sim.destroy()
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
