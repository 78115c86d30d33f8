#!/usr/bin/env python3
"""bench.py -- one JSON line per run (driver contract).

Default workload = BASELINE.json configs[2] ("cfg3"), the configuration the headline metric is quoted on:
SMP_omega (second-order CCN), 3 levels, 64 channels, nFeatures 5, nDepth 5, receptive-field cap 29, a batch of 1024
synthetic QM9-size molecules per GPU (nV ~ U{3..29}), fp32, device-resident.  A "step" = forward + backward over the
rank's batch (+ the single all-reduce of the flat parameter-gradient buffer when N > 1).  Host graph preparation
(receptive fields, index tables) is input preparation: done once before the timed region and reported as prep_s.
`--workload cfg2` runs BASELINE configs[1]: RisiContraction_18 fwd+bwd, N=32, C=64, batch 256 (the north_star target line).

Multi-GPU: molecules / graphs are independent, so ranks shard them (weak scaling: the per-GPU batch is fixed);
the only collective is the gradient all-reduce (RCCL).  Timing: barrier + synchronize on both sides, max over ranks.

roofline: per-kernel HIP-event durations from the library's own launch timers (gf_ctx_set_timing, same stream as the
kernels), for the kernel with the largest total device time; bytes / flops are the algorithmic figures of DESIGN.md.
cpu_baseline: the REAL reference build (oracle/_ref/libgf_ref.so, kind "reference") when shipped, else our port, on one
host core, on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s achievable with a float4 copy)
MFMA_F32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: dense fp32-input MFMA peak


def copy_ceiling_gbps(torch, dev):
    """Measured device-to-device copy rate on this box (read + write bytes / time): the practical HBM ceiling that
    SURVEY.md 8(d) asks to be reported beside the 8 TB/s spec figure."""
    n = 1 << 28   # 1 GiB of fp32 each way
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty(n, dtype=torch.float32, device=dev)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 5 * 2 * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9


def contraction_bytes(K, N, C):
    """fwd 4(N^3 C + N^2 + K N^2 C); bwd 4(K N^2 C + N^2 + N^3 C) (write-only dP)."""
    fwd = 4 * (N ** 3 * C + N * N + K * N * N * C)
    return fwd, fwd


def run_cfg2(args, torch, gf, dev, world, rank):
    K = 50 if args.workload == "cfg5" else 18
    if args.workload == "cfg5":   # BASELINE configs[4]: RisiContraction_50, N=24, 32 channels
        args.N, args.C = 24, 32
    B, N, C = args.batch or 256, args.N, args.C
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

    def finish(timers, ms_per_step):
        fwd_b, bwd_b = contraction_bytes(K, N, C)
        per = {k: v[0] / max(v[1], 1) for k, v in timers.items()}
        dom = max(timers, key=lambda k: timers[k][0])
        is_bwd = ("bwd" in dom) or ("backward" in dom)
        fwd_ms = sum(v for k, v in per.items() if not (("bwd" in k) or ("backward" in k)))
        bwd_ms = sum(v for k, v in per.items() if ("bwd" in k) or ("backward" in k))
        call_ms, call_b, which = (bwd_ms, bwd_b, "backward") if is_bwd else (fwd_ms, fwd_b, "forward")
        ach = call_b * B / (call_ms * 1e-3) / 1e9
        traffic = None   # HBM bytes per launch of the dominant kernel from the committed PMC passes (same shape only)
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_cfg2_hbm_bytes.json")
        if (B, N, C, K) == (256, 32, 64, 18) and os.path.exists(pmc):
            with open(pmc) as fh:
                t = json.load(fh).get(dom)
            if t:
                traffic = round(t["fetch"] + t["write"])
        roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": dom,
                "kernel_ms": {k: round(v, 4) for k, v in per.items()},
                "note": "achieved = algorithmic bytes of one %s call over the batch / (the device time of that call's kernels)" % which,
                "step_GBps": round((fwd_b + bwd_b) * B / (ms_per_step * 1e-3) / 1e9, 1)}
        return roof

    def cpu():
        from oracle import pyoracle
        from inputs import cfg_graph
        if K != 18:
            return None   # the reference's RisiContraction_50 loops take ~8 s per graph at this shape; not re-timed here
        Pc, Ac, Gc = cfg_graph(N, C, 1000, K=18)
        secs, kind, _, _ = pyoracle.time_r18_fwd_bwd(Pc, Ac, Gc)
        out = {"value": round(1.0 / secs, 5), "unit": "graphs/s", "cores": 1, "kind": kind,
               "sample": "1 graph, RisiContraction_18 fwd+bwd, N=%d C=%d fp64, %.1f s" % (N, C, secs)}
        # north_star's multi-thread class, RisiContraction_18_thread (six std::threads by case group, no adjacency gate): its
        # FORWARD only -- the backward races.  Slower than the single-thread op above (two of its six jobs carry
        # the N^5 cases), so `value` stays the stronger baseline; reported beside it.
        ref = pyoracle.reference()
        if ref is not None:
            t0 = time.perf_counter()
            ref.r18_thread_forward(Pc, Ac)
            out["thread_variant"] = {"class": "RisiContraction_18_thread", "threads": 6, "forward_s_per_graph": round(time.perf_counter() - t0, 2)}
        return out

    meta = {"metric": "RisiContraction_%d graphs/sec fwd+bwd (second-order CCN contraction step)" % K, "unit": "graphs/s",
            "units_per_step": B,
            "config": {"workload": "%s: RisiContraction_%d fwd+bwd, N=%d, C=%d, batch=%d graphs/GPU, device-resident"
                                   % (args.workload, K, N, C, B),
                       "parallelism": "graph-sharded x%d, no collective" % world}}
    return ctx, step, finish, cpu, meta, None


def run_cfg3(args, torch, gf, dev, world, rank, dist):
    import numpy as np
    from graphflow_amd.smp import SMPOmega
    from graphflow_amd import dist as gd
    from inputs import smp_params, synthetic_molecule
    B = args.batch or 1024
    L, C, F, D, cap = 3, args.C, 5, 5, 29
    mols, tg = [], []
    for i in range(B):
        adj, feat, t = synthetic_molecule(rank * 1000003 + i)   # seed = molecule index, disjoint across ranks
        mols.append((adj, feat))
        tg.append(t)
    ctx = gf.Context(dev.index)
    net = SMPOmega(L, C, F, D, cap, True, ctx=ctx)
    t0 = time.perf_counter()
    net.prepare(mols)
    prep_first_s = time.perf_counter() - t0   # includes the one-time device allocations (pooled afterwards)
    t0 = time.perf_counter()
    net.prepare(mols)
    prep_s = time.perf_counter() - t0         # steady state: what a training loop pays per new batch
    params = torch.as_tensor(smp_params(C, F, D, L, 1).astype(np.float32)).to(dev)
    targets = torch.as_tensor(np.array(tg, dtype=np.float32)).to(dev)
    grads = torch.empty(net.n_params, device=dev)
    sizes = [net.level_sizes(l) for l in range(L + 1)]   # (nodes, rows = sum s^2, ppos = sum s^3)

    def step():
        net.forward(params, targets)
        net.backward(params, grads)
        gd.allreduce_sum_(grads, dist)   # the one exchange of the path (SMP_omega.h:784-786)

    # algorithmic work per step and per kernel: per level l >= 1 with R = sum s^2 (rows), S = sum s^3
    # (positions), Rp = rows of the level below; bytes are fp32 HBM bytes that MUST move, flops are MFMA flops.
    fused = not args.unfused
    net.set_fused(fused)
    kb = {}   # kernel name -> algorithmic bytes per step
    kf = {}   # kernel name -> flops per step

    def add(d, k, v):
        d[k] = d.get(k, 0) + v

    for l in range(1, L + 1):
        _, R, S = sizes[l]
        _, Rp, _ = sizes[l - 1]
        unit = 2 * R * C * C                                     # one C x C block product over all rows
        if fused:
            add(kb, "smpf_tables_fwd", 4 * (Rp * C + 4 * R * C))        # gather f_{l-1} (cached), write 4 tables
            add(kb, "smpf_combine_fwd", 4 * (3 * R * C + R * C))        # O = [O_loc | Z | Z'] in, f_l out
            add(kb, "smpf_combine_bwd", 4 * (2 * R * C + 3 * R * C))
            if os.environ.get("GF_SMP_BWD_GATHER", "1") != "0":           # dP evaluated inside the consumer gather
                add(kb, "smpf_bwd_gather", 4 * (4 * R * C + Rp * C))    # table gradients in (once), df_{l-1} out
            else:
                add(kb, "smpf_tables_bwd", 4 * (4 * R * C + S * C))
                add(kb, "smp_promote_bwd", 4 * (S * C + Rp * C))
            # the eight row block products of a level, per direction.  C = 64: dedicated kernels (smp_level_c64.hip); other C,
            # or with GF_SMP_ROWPANEL / GF_SMP_WGRAD = 0: grouped launches of the generic GEMM (which also runs the small
            # per-(node,x) / per-node / compact products under these names, not counted here)
            c64 = C == 64
            names = (("smpf_products_fwd" if c64 and os.environ.get("GF_SMP_ROWPANEL", "1") != "0" else "gemm_nn"),
                     ("smpf_products_bwd" if c64 and os.environ.get("GF_SMP_ROWPANEL", "1") != "0" else "gemm_nt"),
                     ("smpf_wgrad" if c64 and os.environ.get("GF_SMP_WGRAD", "1") != "0" else "gemm_tn"))
            for k in names:
                add(kf, k, 8 * unit)
                add(kb, k, 4 * (4 * R * C + 3 * R * C))                  # T (4C) and O / dO (3C) per row, each once
        else:
            add(kb, "smp_promote_fwd", 4 * (Rp * C + S * C))
            add(kb, "r18_fwd_slab", 4 * (S * C + 10 * R * C))
            add(kb, "r18_fwd_rows", 4 * (8 * R * C))
            add(kb, "r18_bwd_rows", 4 * (8 * R * C))
            add(kb, "r18_bwd_slab", 4 * (10 * R * C + S * C))
            add(kb, "smp_promote_bwd", 4 * (S * C + Rp * C))
            for k in ("gemm_nn", "gemm_nt", "gemm_tn"):
                add(kf, k, 18 * unit)
                add(kb, k, 4 * (18 * R * C + R * C))
    step_bytes = sum(kb.values())
    step_flops = sum(kf.values())
    work = {"levels": [{"nodes": n, "rows_sum_s2": r, "ppos_sum_s3": s} for (n, r, s) in sizes],
            "algorithmic_GB_per_step": round(step_bytes / 1e9, 2), "gemm_GFLOP_per_step": round(step_flops / 1e9, 1)}

    def finish(timers, ms_per_step):
        if not timers:   # GF_BENCH_NOTIMING=1 (diagnostic: step time without the per-launch events)
            return {"note": "per-kernel timing disabled"}
        tot = {k: v[0] / args.steps for k, v in timers.items()}   # ms per step per kernel name
        dom = max(tot, key=tot.get)
        if dom in kf:
            ach = kf[dom] / (tot[dom] * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_F32_PEAK_TF, 4), "traffic": None,
                    "note": "fp32-input MFMA (v_mfma_f32_32x32x2_f32): the eight block products of the level projection in this direction, summed over levels"}
        else:
            ach = kb.get(dom, 0) / (tot[dom] * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "note": "algorithmic bytes of this kernel over all levels / its device time per step"}
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_cfg3_hbm_bytes.json")   # committed PMC passes, this exact workload only
        if fused and (B, C) == (1024, 64) and os.path.exists(pmc):
            with open(pmc) as fh:
                t = json.load(fh).get(dom)
            if t:
                roof["traffic"] = round(t["fetch"] + t["write"])   # HBM bytes of all launches of the kernel in one step
        # MFMA utilisation of the C = 64 product kernels from the committed PMC pass (SQ_VALU_MFMA_BUSY_CYCLES over four SIMDs x
        # SQ_BUSY_CU_CYCLES), same workload only
        busy = os.path.join(ROOT, "profiles", "r01_cfg3_mfma_busy.txt")
        real = {"smpf_products_fwd": "smp_rowpanel_c64<true>", "smpf_products_bwd": "smp_rowpanel_c64<false>", "smpf_wgrad": "smp_wgrad_c64"}
        if fused and (B, C) == (1024, 64) and dom in real and os.path.exists(busy):
            for line in open(busy):
                if line.startswith(real[dom]) and "MFMA busy" in line:
                    roof["mfma_busy_pmc"] = float(line.rsplit("MFMA busy", 1)[1])
        roof["kernel"] = dom
        roof["kernel_ms_per_step"] = {k: round(v, 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}
        roof["step_GBps"] = round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1)
        roof["step_gemm_TFLOPs"] = round(step_flops / (ms_per_step * 1e-3) / 1e12, 2)
        return roof

    def cpu():
        from oracle import pyoracle
        # bounded sample: the batch's first molecules until about 20 s of reference work (cost grows steeply with nV)
        sample, est = [], 0.0
        for (a, f), t in zip(mols, tg):
            sample.append(((a, f), t))
            est += 3.5e-4 * len(a) ** 3   # rough: 8.3 s at 29 atoms
            if est > 20.0 or len(sample) >= 16:
                break
        secs = pyoracle.time_reference_smp_omega([m for m, _ in sample], [t for _, t in sample], L, C, D, cap)
        if secs is None:
            return None
        return {"value": round(len(sample) / secs, 4), "unit": "molecules/s", "cores": 1, "kind": "reference",
                "sample": "first %d molecules of the batch (nV %s), SMP_omega complete_computation_graph+forward+backward, fp64, %.1f s"
                          % (len(sample), [len(m[0]) for m, _ in sample], secs)}

    meta = {"metric": "CCN-2D (SMP_omega) molecules/sec fwd+bwd", "unit": "molecules/s", "units_per_step": B,
            "config": {"workload": "cfg3: SMP_omega 3 levels, C=%d, F=5, D=5, cap=29, batch=%d synthetic QM9-size molecules/GPU, device-resident, %s levels"
                                   % (C, B, "fused" if fused else "op-by-op"),
                       "parallelism": "molecule-sharded x%d, one RCCL all-reduce of %d gradient floats per step" % (world, net.n_params)
                       if world > 1 else "single GPU", "prep_s": round(prep_s, 3), "prep_first_s": round(prep_first_s, 3), "work": work}}
    return ctx, step, finish, cpu, meta, net


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "cfg5"])
    ap.add_argument("--batch", type=int, default=0, help="graphs / molecules per GPU (default 256 for cfg2, 1024 for cfg3)")
    ap.add_argument("--N", type=int, default=32)
    ap.add_argument("--C", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="cfg3: op-by-op level pipeline instead of the fused level kernels")
    args = ap.parse_args()

    import torch
    import graphflow_amd as gf
    from graphflow_amd import dist as gd

    world, rank, local = gd.env_world()
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)
    dist = gd.init(backend="nccl", device=dev) if (world > 1 or os.environ.get("GF_FORCE_DIST")) else None

    if args.workload in ("cfg2", "cfg5"):
        ctx, step, finish, cpu, meta, keep = run_cfg2(args, torch, gf, dev, world, rank)
    else:
        ctx, step, finish, cpu, meta, keep = run_cfg3(args, torch, gf, dev, world, rank, dist)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # Per-kernel table: an identical pass of K steps with every launch bracketed by HIP events, OUTSIDE the timed region
    # (the events cost about 5 % of a 72-launch step: they keep neighbouring kernels from overlapping).  The timed region
    # then times only the dominant kernel live -- its average duration is what roofline.achieved is computed from.
    notiming = bool(os.environ.get("GF_BENCH_NOTIMING"))
    timers = {}
    if not notiming:
        ctx.set_timing_filter(None)
        ctx.set_timing(True)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        timers = ctx.timings()
        ctx.set_timing(False)
        dominant = max(timers, key=lambda k: timers[k][0])
        ctx.set_timing_filter(dominant)
        fence()
        ctx.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fence()
    if not notiming:
        live = ctx.timings()
        ctx.set_timing(False)
        ctx.set_timing_filter(None)
        timers[dominant] = live[dominant]
    elapsed = gd.max_over_ranks(elapsed, dist, dev)

    ceiling = copy_ceiling_gbps(torch, dev) if rank == 0 else None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * meta["units_per_step"] * args.steps / elapsed
        line = {"metric": meta["metric"], "value": round(value, 1), "unit": meta["unit"], "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": meta["config"], "roofline": finish(timers, ms_per_step),
                "cpu_baseline": (cpu() if (world == 1 and not args.no_cpu_baseline) else None)}
        line["roofline"]["hbm_copy_measured_GBps"] = round(ceiling, 1)
        if hasattr(keep, "device_bytes"):   # HBM held by the batch's buffers after the steps ran (lazy buffers included)
            line["config"]["device_GB"] = round(keep.device_bytes()[0] / 1e9, 2)
        line["roofline"]["timing"] = ("HIP events on the kernels' stream: the dominant kernel live inside the timed region, "
                                      "the other kernels in an identical pass of the same steps just before it")
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
