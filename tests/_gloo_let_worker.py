"""Worker for tests/test_sharded_gloo.py::test_let_exchange_*: one rank of a CPU (gloo) run of the
LET protocol's host-side exchange (wgpu_n_body_amd.sharded.exchange_segments + the in-place
all-gather of the counts rows), on CPU tensors.  argv: out_dir seg records_per_elem"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from wgpu_n_body_amd.sharded import exchange_segments  # noqa: E402


def counts_for(world):
    """a fixed, ragged counts matrix with zeros in it (diagonal unused)"""
    c = np.zeros((world, world), dtype=np.int64)
    for r in range(world):
        for q in range(world):
            if r != q:
                c[r, q] = (3 * r + 5 * q) % 7
    return c


def main():
    out_dir, seg_records, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    me, world = dist.get_rank(), dist.get_world_size()
    want = counts_for(world)
    # every rank knows only its own row; the matrix is all-gathered in place like region 1/4
    full = torch.zeros(world * world, dtype=torch.int32)
    full[me * world:(me + 1) * world] = torch.from_numpy(want[me].astype(np.int32))
    dist.all_gather_into_tensor(full, full[me * world:(me + 1) * world].clone())
    counts = full.numpy().astype(np.int64).reshape(world, world)
    seg = seg_records * R
    send = torch.full((world * seg,), -1.0, dtype=torch.float32)
    for q in range(world):
        for k in range(int(counts[me, q])):
            for e in range(R):   # value encodes (source, destination, record, element)
                send[q * seg + k * R + e] = me * 1000 + q * 100 + k * 10 + e
    recv = torch.full((world * seg,), -2.0, dtype=torch.float32)
    got = exchange_segments(send, seg, recv, counts, me, world, R)
    np.savez(os.path.join(out_dir, f"let_rank{me}.npz"), recv=recv.numpy(), got=np.array(got), counts=counts)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
