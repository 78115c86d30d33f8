#!/usr/bin/env python3
"""bench.py -- one JSON line per run (driver contract).

Workload (N=1 and per rank for N>1, weak scaling): BASELINE.json configs[1] ("cfg2") --
RisiContraction_18 forward+backward, N=32 vertices, 64 channels, batch 256 graphs, fp32, device-resident.
A "step" is one forward + one backward pass over the rank's batch.  Graphs are independent, so ranks shard the
batch with no data-path collective; torch.distributed (RCCL) is only used for the barrier and the max-over-ranks time.

roofline: per-kernel HIP-event durations come from the library's own launch timers (gf_ctx_set_timing) on the stream
the kernels run on; `achieved` = algorithmic bytes of the dominant kernel's call / its mean duration.
cpu_baseline: the REAL reference (oracle/_ref/libgf_ref.so, kind "reference") when shipped, else our loop-nest port,
one graph of the same shape on one host core (about 20 s).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s achievable with a float4 copy)


def algorithmic_bytes(K, N, C, accumulate=False):
    """fwd 4(N^3 C + N^2 + K N^2 C); bwd 4(K N^2 C + N^2 + N^3 C) (+4 N^3 C when accumulating)."""
    fwd = 4 * (N ** 3 * C + N * N + K * N * N * C)
    bwd = 4 * (K * N * N * C + N * N + N ** 3 * C) + (4 * N ** 3 * C if accumulate else 0)
    return fwd, bwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="graphs per GPU")
    ap.add_argument("--N", type=int, default=32)
    ap.add_argument("--C", type=int, default=64)
    ap.add_argument("--K", type=int, default=18)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import graphflow_amd as gf

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)

    B, N, C, K = args.batch, args.N, args.C, args.K
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    P = torch.rand((B, N, N, N, C), device=dev, generator=gen) * 2 - 1          # U(-1,1)
    U = (torch.rand((B, N, N), device=dev, generator=gen) < 0.5).float().triu(1)
    A = (U + U.transpose(1, 2) + torch.eye(N, device=dev)).contiguous()          # symmetric ER(0.5) 0/1 + unit diagonal
    G = torch.rand((B, N, N, K, C), device=dev, generator=gen)                   # U(0,1)
    Out = torch.empty((B, N, N, K, C), device=dev)
    dP = torch.empty((B, N, N, N, C), device=dev)

    ctx = gf.Context(dev.index)
    ctx.reserve(gf.contract_workspace_bytes(K, N, C, B))

    def step():
        gf.contract_forward(P, A, K, out=Out, ctx=ctx)
        gf.contract_backward(G, A, K, dP=dP, accumulate=False, ctx=ctx)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fence()
    timers = ctx.timings()
    ctx.set_timing(False)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        graphs_per_s = world * B * args.steps / elapsed
        fwd_b, bwd_b = algorithmic_bytes(K, N, C)
        # dominant kernel = the one with the largest total device time
        per_kernel = {k: v[0] / max(v[1], 1) for k, v in timers.items()}
        dom = max(timers, key=lambda k: timers[k][0]) if timers else None
        # algorithmic bytes are defined per CALL (fwd or bwd); a call's duration = its two kernels
        fwd_ms = sum(v for k, v in per_kernel.items() if "fwd" in k)
        bwd_ms = sum(v for k, v in per_kernel.items() if "bwd" in k)
        call_ms, call_bytes = (fwd_ms, fwd_b) if (dom and "fwd" in dom) else (bwd_ms, bwd_b)
        achieved = (call_bytes * B / (call_ms * 1e-3)) / 1e9 if call_ms > 0 else 0.0
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "kernel": dom, "kernel_ms": {k: round(v, 4) for k, v in per_kernel.items()},
                    "note": "achieved = algorithmic bytes of one %s call over the batch / (its slab+rows kernel time)"
                            % ("forward" if (dom and "fwd" in dom) else "backward"),
                    "step_GBps": round((fwd_b + bwd_b) * B / (ms_per_step * 1e-3) / 1e9, 1)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import pyoracle
            from inputs import cfg_graph
            Pc, Ac, Gc = cfg_graph(N, C, 1000, K=18)
            secs, kind, _, _ = pyoracle.time_r18_fwd_bwd(Pc, Ac, Gc)
            cpu = {"value": round(1.0 / secs, 5), "unit": "graphs/s", "cores": 1, "kind": kind,
                   "sample": "1 graph, RisiContraction_18 fwd+bwd, N=%d C=%d fp64, %.1f s" % (N, C, secs)}
        line = {
            "metric": "RisiContraction_18 graphs/sec fwd+bwd (second-order CCN contraction step)",
            "value": round(graphs_per_s, 1), "unit": "graphs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg2: RisiContraction_%d fwd+bwd, N=%d, C=%d, batch=%d graphs/GPU, device-resident"
                                   % (K, N, C, B), "parallelism": "graph-sharded x%d, no collective" % world},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
