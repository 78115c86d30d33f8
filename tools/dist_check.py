#!/usr/bin/env python3
"""One rank of the multi-GPU gradient check (tests/test_dist_gpu.py::test_ranks_on_separate_gpus...): launched by
torch.distributed.run with N ranks on N GPUs.  Every rank takes its contiguous shard of a 32-molecule batch, runs forward +
backward with the context's RCCL communicator (gf_dist_*: the per-level gradient segments are all-reduced inside
gf_smp_backward), and rank 0 saves the resulting gradient -- which must equal the single-GPU gradient of the whole batch.
usage (under the launcher): dist_check.py out.npy"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import graphflow_amd as gf  # noqa: E402
from graphflow_amd import dist as gd  # noqa: E402
from graphflow_amd.smp import SMPOmega  # noqa: E402
from inputs import smp_params, synthetic_molecule  # noqa: E402


def main():
    world, rank, local = gd.env_world()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist = gd.init(backend="nccl", device=dev)
    ctx = gf.Context(dev.index)
    uid = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(ctx.dist_unique_id()), dtype=torch.uint8))
    dist.broadcast(uid, src=0)
    ctx.dist_init(bytes(uid.cpu().numpy().tobytes()), rank, world)
    assert (ctx.dist_rank, ctx.dist_world) == (rank, world)
    L, C, F, D, cap = 3, 64, 5, 3, 29
    mols, tg = [], []
    for seed in range(32):
        a, f, t = synthetic_molecule(4100 + seed)
        mols.append((a, f))
        tg.append(t)
    lo, hi = gd.shard(len(mols), rank, world)
    net = SMPOmega(L, C, F, D, cap, True, ctx=ctx)
    net.prepare(mols[lo:hi])
    p = torch.as_tensor(smp_params(C, F, D, L, 8).astype(np.float32)).to(dev)
    t = torch.as_tensor(np.array(tg[lo:hi], dtype=np.float32)).to(dev)
    g = torch.full((net.n_params,), float("nan"), device=dev)
    for _ in range(2):
        net.forward(p, t)
        net.backward(p, g)
    torch.cuda.synchronize()
    ref = g.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(ref, g), "rank %d holds a different sum from rank 0" % rank
    if rank == 0:
        np.save(sys.argv[1], g.cpu().numpy())
    dist.barrier()
    net.close()
    ctx.dist_finalize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
