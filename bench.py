#!/usr/bin/env python3
"""bench.py -- one JSON line per run (driver contract).

Default workload = BASELINE.json configs[2] ("cfg3"), the configuration the headline metric is quoted on:
SMP_omega (second-order CCN), 3 levels, 64 channels, nFeatures 5, nDepth 5, receptive-field cap 29, a batch of 1024
synthetic QM9-size molecules per GPU (nV ~ U{3..29}), fp32, device-resident.  A "step" = forward + backward over the
rank's batch; with N > 1 the backward leaves the parameter gradients summed over all ranks (RCCL all-reduce of each
level's segment behind the C ABI, gf_dist_*, overlapped with the rest of the reverse sweep).
`value` times the device step with the batch's index tables resident (host graph preparation is input preparation: once,
before the timed region, reported as prep_s); `end_to_end` in the same line is the training-loop figure -- a NEW batch
every step (host graph preparation + upload overlapped on a second handle) + forward + backward + Adam.
`--workload cfg2` / `cfg5` run BASELINE configs[1] / configs[4] (RisiContraction_18 N=32 C=64, RisiContraction_50 N=24
C=32, batch 256) as the headline; the default line carries both under `extra`, 20 steps each, with their own rooflines.

Multi-GPU: `python bench.py --gpus N` spawns N ranks itself (torch.distributed.run on 127.0.0.1) unless it already runs
under a launcher (WORLD_SIZE set, which must equal N).  Molecules / graphs are independent, so ranks shard them (weak
scaling: the per-GPU batch is fixed).  Timing: barrier + synchronize on both sides, max over ranks.

roofline: per-kernel HIP-event durations from the library's own launch timers (gf_ctx_set_timing, same stream as the
kernels), for the kernel with the largest total device time; bytes / flops are the algorithmic figures of DESIGN.md.
cpu_baseline: kind "port" -- oracle/gf_oracle.c + oracle/smp_port.c, the C restatement of the reference's loop nests, whose
run time is held against the real reference in the build container (profiles/r02_port_vs_reference.json; the reference
itself never travels to the GPU box) -- on one host core, on six threads (RisiContraction_18_thread's forward) and
batch-parallel on all host cores (the shape of Threaded_BatchLearn), each on a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s achievable with a float4 copy)
MFMA_F32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: dense fp32-input MFMA peak
MFMA_F16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak (the vendor's 5 PF figure includes 2:1 sparsity)


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


def copy_probe_gbps(torch, dev, ctx):
    """The same 1 GiB -> 1 GiB copy by the library's own hand-written kernels (gf_hbm_copy_probe_f32: float4 grid-stride, plain and
    non-temporal): the yardstick MI355X_MICROARCH.md quotes (6.29 TB/s, 79 % of the 8 TB/s spec) measured on THIS box, beside torch's
    copy_ (round-5 review, weak #7: torch's copy_ sits ~20 % below what the part does)."""
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    out = {"float4": round(ctx.hbm_copy_probe(b, a, 0, 5), 1), "float4_nontemporal": round(ctx.hbm_copy_probe(b, a, 1, 5), 1)}
    assert torch.equal(a, b)
    return out


def contraction_bytes(K, N, C):
    """fwd 4(N^3 C + N^2 + K N^2 C); bwd 4(K N^2 C + N^2 + N^3 C) (write-only dP)."""
    fwd = 4 * (N ** 3 * C + N * N + K * N * N * C)
    return fwd, fwd


def host_threads():
    """Threads of the all-cores CPU baseline: every host core, capped (a 29-atom molecule holds about 1 GB in the port)."""
    return max(1, min(os.cpu_count() or 1, 32))


def latest_profile(name):
    """profiles/rNN_<name> of the newest round that committed one."""
    best = None
    for f in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if f.endswith(name) and f[0] == "r" and f[1:3].isdigit():
            best = f
    return os.path.join(ROOT, "profiles", best) if best else None


# ---- the timed region (shared by the headline workload and the `extra` ones) ------------------------------------------
def timed_run(torch, ctx, step, steps, warmup, fence, notiming=False, repeats=1, quiesce=None):
    """warmup untimed steps, then an identical pass of `steps` steps with every launch bracketed by HIP events OUTSIDE the
    timed region (the per-kernel table: the events cost about 5 % of a 70-launch step, they keep neighbouring kernels from
    overlapping), then the timed region proper, in which only the dominant kernel is timed live -- its average duration is
    what roofline.achieved is computed from.  The timed region runs `repeats` times (exactly `steps` steps each, every one bracketed
    by fence = barrier + synchronize on both sides): the caller reports the median and the spread.
    quiesce: a data-parallel run's bounded wait for its collectives (gf_dist_quiesce) ahead of every blocking synchronize.
    Returns ([elapsed seconds of this rank per repeat], {kernel: (total ms, launches)})."""
    def sync():
        if quiesce is not None:
            quiesce()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    if quiesce is not None:
        quiesce()
    fence()
    timers, dominant = {}, None
    if not notiming:
        ctx.set_timing_filter(None)
        ctx.set_timing(True)
        for _ in range(steps):
            step()
        sync()
        timers = ctx.timings()
        ctx.set_timing(False)
        dominant = max((k for k in timers if not k.startswith("rccl_")), key=lambda k: timers[k][0])
        ctx.set_timing_filter(dominant)
        fence()
        ctx.set_timing(True)
    elapsed = []
    for _ in range(max(1, repeats)):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        elapsed.append(time.perf_counter() - t0)
        fence()
    if not notiming:
        live = ctx.timings()
        ctx.set_timing(False)
        ctx.set_timing_filter(None)
        n = max(1, repeats)
        timers[dominant] = (live[dominant][0] / n, live[dominant][1] // n)   # (per repeat: the table's other entries are one pass of `steps` steps)
        if "rccl_allreduce" in live:
            timers["rccl_allreduce"] = (live["rccl_allreduce"][0] / n, live["rccl_allreduce"][1] // n)
    return elapsed, timers


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])


# ---- cfg2 / cfg5: one contraction family, forward + backward ---------------------------------------------------------------
def setup_contraction(workload, args, torch, gf, dev, world, rank, ctx):
    K, N, C = (50, 24, 32) if workload == "cfg5" else (18, args.N, args.C)
    B = (args.batch if workload == args.workload else 0) or 256
    if args.scaling == "strong" and workload == args.workload:
        B = (args.batch or 2048) // world
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    P = torch.rand((B, N, N, N, C), device=dev, generator=gen) * 2 - 1          # U(-1,1)
    U = (torch.rand((B, N, N), device=dev, generator=gen) < 0.5).float().triu(1)
    A = (U + U.transpose(1, 2) + torch.eye(N, device=dev)).contiguous()          # symmetric ER(0.5) 0/1 + unit diagonal
    G = torch.rand((B, N, N, K, C), device=dev, generator=gen)                   # U(0,1)
    Out = torch.empty((B, N, N, K, C), device=dev)
    dP = torch.empty((B, N, N, N, C), device=dev)
    ctx.reserve(gf.contract_workspace_bytes(K, N, C, B))

    def step():
        gf.contract_forward(P, A, K, out=Out, ctx=ctx)
        gf.contract_backward(G, A, K, dP=dP, accumulate=False, ctx=ctx)

    def finish(timers, ms_per_step, steps):
        fwd_b, bwd_b = contraction_bytes(K, N, C)
        if not timers:
            return {"note": "per-kernel timing disabled"}
        per = {k: v[0] / max(steps, 1) for k, v in timers.items()}   # ms per STEP under that launch name (fam_adj runs in both calls)
        dom = max(timers, key=lambda k: timers[k][0])
        is_bwd = ("bwd" in dom) or ("backward" in dom)
        fwd_ms = sum(v for k, v in per.items() if not (("bwd" in k) or ("backward" in k)))
        bwd_ms = sum(v for k, v in per.items() if ("bwd" in k) or ("backward" in k))
        call_ms, call_b, which = (bwd_ms, bwd_b, "backward") if is_bwd else (fwd_ms, fwd_b, "forward")
        ach = call_b * B / (call_ms * 1e-3) / 1e9
        traffic = None   # HBM bytes per step of the dominant kernel from the committed PMC passes (same shape only)
        pmc = latest_profile("_pmc_%s_hbm_bytes.json" % workload)
        if (B, N, C, K) in ((256, 32, 64, 18), (256, 24, 32, 50)) and pmc:
            with open(pmc) as fh:
                t = json.load(fh).get(dom)
            if t:
                traffic = round(t["fetch"] + t["write"])
        return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": dom,
                "kernel_ms": {k: round(v, 4) for k, v in per.items()},
                "note": "achieved = algorithmic bytes of one %s call over the batch / (the device time of that call's kernels)" % which,
                "step_GBps": round((fwd_b + bwd_b) * B / (ms_per_step * 1e-3) / 1e9, 1),
                "step_frac_of_hbm_peak": round((fwd_b + bwd_b) * B / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    def cpu():
        from oracle import pyoracle
        from inputs import cfg_graph
        if K != 18:   # cfg5: the loop-nest port of RisiContraction_50 (one five-deep nest per channel, 50 predicated updates inside)
            Pc, Ac, Gc = cfg_graph(N, C, 1000, K=K)
            secs, _, _ = pyoracle.time_r50_fwd_bwd(Pc, Ac, Gc)
            return {"value": round(1.0 / secs, 5), "unit": "graphs/s", "cores": 1, "kind": "port",
                    "sample": "1 graph, RisiContraction_50 fwd+bwd (loop nests of RisiContraction_50.h:73-802), N=%d C=%d fp64, %.1f s" % (N, C, secs)}
        Pc, Ac, Gc = cfg_graph(N, C, 1000, K=18)
        secs, kind, _, _ = pyoracle.time_r18_fwd_bwd(Pc, Ac, Gc, prefer_reference=False)
        out = {"value": round(1.0 / secs, 5), "unit": "graphs/s", "cores": 1, "kind": kind,
               "sample": "1 graph, RisiContraction_18 fwd+bwd (nnz-gated loop nests of RisiContraction_18.h:73-560), N=%d C=%d fp64, %.1f s" % (N, C, secs)}
        # north_star's multi-thread class, RisiContraction_18_thread (six threads by case group, no adjacency gate): its FORWARD
        # only -- the backward races.  Two of its six jobs carry the N^5 cases, so it is no faster than the
        # single-thread op above; reported beside it.
        orc = pyoracle.oracle()
        t0 = time.perf_counter()
        orc.r18_thread_forward(Pc, Ac)
        out["thread_variant"] = {"class": "RisiContraction_18_thread (port of forward_job_0..5)", "threads": 6,
                                 "forward_s_per_graph": round(time.perf_counter() - t0, 2)}
        nT = host_threads()
        secs_all = pyoracle.port_r18_batch_threads(Pc, Ac, Gc, nT, nT)
        out["all_cores"] = {"value": round(nT / secs_all, 4), "unit": "graphs/s", "cores": nT, "kind": "port",
                            "sample": "%d graphs fwd+bwd, one per host thread (batch-parallel), %.1f s" % (nT, secs_all)}
        return out

    meta = {"metric": "RisiContraction_%d graphs/sec fwd+bwd (second-order CCN contraction step)" % K, "unit": "graphs/s",
            "units_per_step": B,
            "config": {"workload": "%s: RisiContraction_%d fwd+bwd, N=%d, C=%d, batch=%d graphs/GPU, device-resident"
                                   % (workload, K, N, C, B),
                       "parallelism": "graph-sharded x%d, no collective" % world}}
    return step, finish, cpu, meta, None


# ---- cfg3: the batched SMP_omega step --------------------------------------------------------------------------------------
def setup_smp(args, torch, gf, dev, world, rank, ctx):
    import numpy as np
    from graphflow_amd.smp import SMPOmega
    from inputs import smp_params, synthetic_molecule
    B = (args.batch or 8192) // world if args.scaling == "strong" else (args.batch or 1024)
    L, C, F, D, cap = 3, args.C, 5, 5, 29
    mols, tg = [], []
    for i in range(B):
        adj, feat, t = synthetic_molecule(rank * 1000003 + i)   # seed = molecule index, disjoint across ranks
        mols.append((adj, feat))
        tg.append(t)
    if world > 1:   # N ranks share the host: each rank's two loader threads get their share of the cores for graph preparation
        os.environ.setdefault("GF_PREP_THREADS", str(max(2, min(32, (os.cpu_count() or 8) // (2 * world)))))
    nK = getattr(args, "nK", 18)   # (extra lines: the SMP_2D_ver6 / ver7 wirings, RisiContraction_10 / _50 with CustomMatMulTensor weights)
    net = SMPOmega(L, C, F, D, cap, True, ctx=ctx, nContractions=nK, custom_matmul=(nK != 18))
    t0 = time.perf_counter()
    net.prepare(mols)
    prep_first_s = time.perf_counter() - t0   # includes the one-time device allocations (pooled afterwards)
    t0 = time.perf_counter()
    net.prepare(mols)
    prep_s = time.perf_counter() - t0         # steady state: what a training loop pays per new batch
    if nK == 18:
        params = torch.as_tensor(smp_params(C, F, D, L, 1).astype(np.float32)).to(dev)
    else:
        params = torch.as_tensor((np.random.default_rng(1).uniform(-1, 1, net.n_params) / np.sqrt(nK * C)).astype(np.float32)).to(dev)
    targets = torch.as_tensor(np.array(tg, dtype=np.float32)).to(dev)
    grads = torch.empty(net.n_params, device=dev)
    sizes = [net.level_sizes(l) for l in range(L + 1)]   # (nodes, rows = sum s^2, ppos = sum s^3)
    fused = not args.unfused
    net.set_fused(fused)

    def step():
        net.forward(params, targets)
        net.backward(params, grads)   # N > 1: returns the gradient summed over ranks (gf_dist_*, the one exchange of the path)

    # algorithmic work per step and per kernel: per level l >= 1 with R = sum s^2 (rows), S = sum s^3
    # (positions), Rp = rows of the level below; bytes are fp32 HBM bytes that MUST move, flops are MFMA flops.
    kb, kf = {}, {}

    def add(d, k, v):
        d[k] = d.get(k, 0) + v

    c64 = C in (64, 32, 16)   # the dedicated product / weight-gradient kernels (row panels; C = 32 since round 4, 16 since round 5)
    # projected matrix O / dO: [O_loc | U] (2C) when the three dedicated C = 64 product kernels run, else [O_loc | Z | Z'] (3C)
    oc = 2 if (c64 and os.environ.get("GF_SMP_ROWPANEL", "1") != "0") else 3
    # ... and then the three product kernels run on the f16 matrix pipe with two-half fp32 operands (smp_level_c64_split.hip): 3 MFMA
    # flops of a 16x faster pipe per algorithmic flop -- they are HBM streams, priced against the HBM roofline
    split = oc == 2 and os.environ.get("GF_SMP_SPLIT", "1") != "0"
    # rows (a, b) whose S_ab / T6 table blocks are not structural zeros (b inside the field of a's source): at C = 64 the other rows'
    # blocks are neither written (tables-forward), read (forward products, weight gradients) nor back-propagated (backward products,
    # gather) -- bytes that need not move are not counted as moved
    masked = split and fused and os.environ.get("GF_SMP_MASK_ZEROS", "1") != "0"
    present = [net.level_present_rows(l) for l in range(L + 1)]
    covered = [net.level_covered_rows(l) for l in range(L + 1)]   # rows (b, c) some source covers: the S_bc / T10 blocks with data
    for l in range(1, L + 1):
        _, R, S = sizes[l]
        _, Rp, _ = sizes[l - 1]
        unit = 2 * R * C * C                                     # one C x C block product over all rows
        Tb = (2 * covered[l] + 2 * present[l]) * C if masked else 4 * R * C   # floats of T that move (written once, read twice)
        dTw = (2 * R + 2 * present[l]) * C if masked else 4 * R * C              # floats of dT written (its S_bc / T10 half: every row)
        if fused:
            add(kb, "smpf_tables_fwd", 4 * (Rp * C + Tb))               # gather f_{l-1} (cached), write 4 tables
            add(kb, "smpf_combine_fwd", 4 * (oc * R * C + R * C))       # O in, f_l out
            add(kb, "smpf_combine_bwd", 4 * (2 * R * C + oc * R * C))
            if os.environ.get("GF_SMP_BWD_GATHER", "1") != "0":           # dP evaluated inside the consumer gather
                add(kb, "smpf_bwd_gather", 4 * (Tb + Rp * C))           # table gradients in (once), df_{l-1} out
            else:
                add(kb, "smpf_tables_bwd", 4 * (4 * R * C + S * C))
                add(kb, "smp_promote_bwd", 4 * (S * C + Rp * C))
            names = (("smpf_products_fwd" if c64 and os.environ.get("GF_SMP_ROWPANEL", "1") != "0" else "gemm_nn"),
                     ("smpf_products_bwd" if c64 and os.environ.get("GF_SMP_ROWPANEL", "1") != "0" else "gemm_nt"),
                     ("smpf_wgrad" if c64 and os.environ.get("GF_SMP_ROWPANEL", "1") != "0" else "gemm_tn"))
            for k in names:
                add(kf, k, 8 * unit)
                add(kb, k, 4 * ((dTw if k == "smpf_products_bwd" else Tb) + oc * R * C))   # T (4C) and O / dO per row, each once
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
    # ... and the level's small kernels (per-(node, x) vectors, per-node scalars, the compact diagonal rows of the level below, the
    # partial images of the weight gradients): bytes that must move per step, so that no kernel above 1 % of the step goes unpriced
    if fused and c64 and split:
        pairs = [net.level_pairs(l) for l in range(L + 1)]
        for l in range(1, L + 1):
            n_l, R, _ = sizes[l]
            pr, pp = pairs[l], pairs[l - 1]
            add(kb, "smpf_diag_gather", 4 * (2 * pp * C + 2 * pp * C))                     # f_{l-1}[w][p,p], [p,c_w] in; Fdc out
            add(kb, "smpf_small_nn", 4 * (4 * pr * C + pr * C + 4 * n_l * C + n_l * C + 2 * pp * C + 2 * pp * C))   # Vt -> Vout, St -> Sout, Fdc -> Gc
            add(kb, "smpf_small_nt", 4 * (pr * C + 4 * pr * C + n_l * C + 4 * n_l * C + 2 * pp * C + 2 * pp * C))   # dVout -> dVt, dSout -> dSt, dGc -> dFdc
            add(kb, "smpf_small_tn", 4 * (4 * pr * C + pr * C + 4 * n_l * C + n_l * C + 4 * pp * C))                # Vt, dVout, St, dSout, Fdc, dGc in
            add(kb, "smpf_diag_gather_bwd", 4 * (2 * present[l] * C + 2 * pp * C))          # dU[(a, b)], dU[(b, a)] of the rows with data in; dGc out
            # dSpart, dbpart in; dSout out -- since round 6 in ONE launch with the diagonal gather (GF_SMP_FUSE_SMALL=0: its own)
            add(kb, "smpf_reduce_pairs" if os.environ.get("GF_SMP_FUSE_SMALL", "1") == "0" else "smpf_diag_gather_bwd", 4 * (2 * pr * C + n_l * C))
            add(kb, "smpf_fold", 4 * (256 * 8 * C * C + 18 * C * C))                         # <= 256 row-range images of the eight products + the small ones
    step_bytes = sum(kb.values())
    step_flops = sum(kf.values())
    work = {"levels": [{"nodes": n, "rows_sum_s2": r, "ppos_sum_s3": s, "rows_with_data": p, "rows_covered": q} for (n, r, s), p, q in zip(sizes, present, covered)],
            "algorithmic_GB_per_step": round(step_bytes / 1e9, 2), "gemm_GFLOP_per_step": round(step_flops / 1e9, 1)}

    def finish(timers, ms_per_step, steps):
        if not timers:   # GF_BENCH_NOTIMING=1 (diagnostic: step time without the per-launch events)
            return {"note": "per-kernel timing disabled"}
        sym = {k: v[0] / steps for k, v in timers.items()}   # ms per step per launch name = per kernel symbol, as rocprof's table has them
        dom = max(sym, key=sym.get)                          # the dominant kernel: the symbol with the largest time per step
        # tables-forward runs as one template instantiation per size class (smpf_tables_fwd_ni1 / 2 / 4 / 8): its algorithmic bytes are
        # known for the kernel as a whole, so the per-kernel lines below carry their sum under the family's name
        tot = {}
        for k, v in sym.items():
            fam = "smpf_tables_fwd" if k.startswith("smpf_tables_fwd") else k
            tot[fam] = tot.get(fam, 0.0) + v
        if dom.startswith("smpf_tables_fwd"):
            dom = "smpf_tables_fwd"
        mfma_bound = set() if split else set(kf)
        if dom in mfma_bound:
            ach = kf[dom] / (tot[dom] * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_F32_PEAK_TF, 4), "traffic": None,
                    "note": "fp32-input MFMA (v_mfma_f32_32x32x2_f32): the eight block products of the level projection in this direction, summed over levels"}
        else:
            ach = kb.get(dom, 0) / (tot[dom] * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "note": "algorithmic bytes of this kernel over all levels / its device time per step"}
        if split:
            roof["products"] = ("block products on v_mfma_f32_32x32x16_f16 with two-half fp32 operands (3 MFMAs per product term): "
                                "byte-bound, priced against HBM; GF_SMP_SPLIT=0 runs them on the fp32 pipe (MFMA-bound)")
        pmc = latest_profile("_pmc_cfg3_hbm_bytes.json")   # committed PMC passes, this exact workload only
        if fused and (B, C) == (1024, 64) and pmc:
            with open(pmc) as fh:
                t = json.load(fh).get(dom)
            if t:
                roof["traffic"] = round(t["fetch"] + t["write"])   # HBM bytes of all launches of the kernel in one step
                roof["traffic_source"] = os.path.basename(pmc)
                # (a committed PMC pass, not this run's: the commit the pass was taken at rides along -- every kernel change re-runs it)
                roof["traffic_commit"] = (json.load(open(pmc)).get("_meta") or {}).get("commit")
        busy = latest_profile("_cfg3_mfma_busy.txt")
        real = {"smpf_products_fwd": "smp_rowpanel_c64<true>", "smpf_products_bwd": "smp_rowpanel_c64<false>", "smpf_wgrad": "smp_wgrad_c64"}
        if fused and (B, C) == (1024, 64) and dom in real and dom in mfma_bound and busy:
            for line in open(busy):
                if line.startswith(real[dom]) and "MFMA busy" in line:
                    roof["mfma_busy_pmc"] = float(line.rsplit("MFMA busy", 1)[1])
        roof["kernel"] = dom
        # the same ranking with the template instantiations of ONE kernel summed (tables-forward runs as four size classes): the family
        # with the largest time per step and its own fraction, beside the by-symbol line above
        fam_dom = max((k for k in tot if not k.startswith("rccl_")), key=tot.get)
        fam_bytes = kb.get(fam_dom, 0)
        roof["dominant_by_family"] = {"kernel": fam_dom, "ms_per_step": round(tot[fam_dom], 4),
                                      "achieved": round(fam_bytes / (tot[fam_dom] * 1e-3) / 1e9, 1) if fam_bytes else None, "unit": "GB/s",
                                      "frac": round(fam_bytes / (tot[fam_dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if fam_bytes else None}
        unpriced = {k: v for k, v in tot.items() if k not in kb and k not in kf and not k.startswith("rccl_")}
        roof["unpriced_kernels_ms_per_step"] = round(sum(unpriced.values()), 4)   # kernels without an entry in the byte model (each < 1 % of the step)
        roof["kernel_ms_per_step"] = {k: round(v, 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}
        roof["kernel_ms_per_step"].update({k: round(v, 3) for k, v in sym.items() if k.startswith("smpf_tables_fwd_")})   # (its instantiations)
        roof["launches_per_step"] = round(sum(v[1] for v in timers.values() if v[1] > steps / 2) / steps, 1) if steps else None
        # every kernel with an algorithmic figure, against its own bound (the step is a composite of byte- and MFMA-bound kernels)
        roof["per_kernel_frac"] = {k: round((kf[k] / (tot[k] * 1e-3) / 1e12 / MFMA_F32_PEAK_TF) if k in mfma_bound
                                            else (kb[k] / (tot[k] * 1e-3) / 1e9 / HBM_PEAK_GBS), 3)
                                   for k in tot if (k in mfma_bound or k in kb) and tot[k] > 0}
        ideal_ms = sum(kf[k] / (MFMA_F32_PEAK_TF * 1e12) * 1e3 if k in mfma_bound else kb[k] / (HBM_PEAK_GBS * 1e9) * 1e3
                       for k in mfma_bound | set(kb))
        roof["step_composite_frac"] = round(ideal_ms / ms_per_step, 4)
        roof["step_GBps"] = round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1)
        roof["step_gemm_TFLOPs"] = round(step_flops / (ms_per_step * 1e-3) / 1e12, 2)
        # north_star: "MFMA utilisation reported against gfx950 peak".  Split path: every algorithmic product term is three
        # v_mfma_f32_32x32x16_f16 (ah bh + ah bl + al bh), so the matrix pipe executes 3x the algorithmic flops, at the f16 rate; the
        # fp32 pipe (GF_SMP_SPLIT=0) executes them once at the fp32-input rate.  Per product kernel: executed flops / its device time
        # / the dense peak of the instruction it issues; `busy_pmc`: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES of the committed
        # counter pass (profiles/*_cfg3_mfma_busy.txt), when there is one for the kernels that ran.
        peak, mult = (MFMA_F16_PEAK_TF, 3.0) if split else (MFMA_F32_PEAK_TF, 1.0)
        mu = {"instruction": "v_mfma_f32_32x32x16_f16 (3 per product term)" if split else "v_mfma_f32_32x32x2_f32", "peak_TFLOPs": peak,
              "per_kernel": {k: round(mult * kf[k] / (tot[k] * 1e-3) / 1e12 / peak, 4) for k in kf if k in tot and tot[k] > 0},
              "step": round(mult * step_flops / (ms_per_step * 1e-3) / 1e12 / peak, 4)}
        if busy and fused and (B, C) == (1024, 64):
            names = {"smpf_products_fwd": ("smp_rowpanel_split<true", "smp_rowpanel_c64<true"), "smpf_products_bwd": ("smp_rowpanel_split<false", "smp_rowpanel_c64<false"),
                     "smpf_wgrad": ("smp_wgrad_split", "smp_wgrad_c64")}
            pm = {}
            for line in open(busy):
                for k, (a, b) in names.items():
                    if line.startswith(a if split else b) and "MFMA busy" in line:
                        pm[k] = float(line.rsplit("MFMA busy", 1)[1])
            if pm:
                mu["busy_pmc"] = pm
                mu["busy_pmc_source"] = os.path.basename(busy)
        roof["mfma_util"] = mu
        return roof

    def cpu():
        from oracle import pyoracle
        # bounded samples of the same batch: its first molecules until about 8 s of work on one core (cost grows steeply with
        # nV: 8.3 s at 29 atoms in the reference), then one wave of the batch-parallel driver on every host core
        sample, est = [], 0.0
        for (a, f), t in zip(mols, tg):
            sample.append(((a, f), t))
            est += 3.5e-4 * len(a) ** 3
            if est > 8.0 or len(sample) >= 16:
                break
        p64 = params.cpu().numpy().astype(np.float64)
        secs, _, _, _ = pyoracle.port_smp_batch([m for m, _ in sample], [t for _, t in sample], p64, L, C, D, cap, 1)
        out = {"value": round(len(sample) / secs, 4), "unit": "molecules/s", "cores": 1, "kind": "port",
               "sample": "first %d molecules of the batch (nV %s), SMP_omega op DAG forward+backward (oracle/smp_port.c), fp64, %.1f s"
                         % (len(sample), [len(m[0]) for m, _ in sample], secs)}
        nT = host_threads()
        wave = [(mols[i], tg[i]) for i in range(min(nT, len(mols)))]
        secs_all, _, _, _ = pyoracle.port_smp_batch([m for m, _ in wave], [t for _, t in wave], p64, L, C, D, cap, nT)
        out["all_cores"] = {"value": round(len(wave) / secs_all, 4), "unit": "molecules/s", "cores": nT, "kind": "port",
                            "sample": "first %d molecules of the batch, one host thread each (the wave structure of Threaded_BatchLearn, SMP_omega.h:750-792), %.1f s"
                                      % (len(wave), secs_all)}
        return out

    def end_to_end(steps):
        """Training loop with a NEW batch every step: gf_smp_prepare (host graph preparation + upload) of batch i+1 on a second
        handle while the device runs forward + backward + Adam of batch i.  molecules/s of this rank."""
        pool = [synthetic_molecule(7000003 + rank * 1000003 + i) for i in range(3 * B)]
        # (a data loader hands over flat arrays: packing the Python molecule lists is input generation, like drawing them)
        batches = [(SMPOmega.pack(mols), targets)]
        for i in range(3):
            sl = pool[i * B:(i + 1) * B]
            batches.append((SMPOmega.pack([(a, f) for a, f, _ in sl]), torch.as_tensor(np.array([t for *_, t in sl], dtype=np.float32)).to(dev)))
        # The host's graph preparation of a batch (10-11 ms on the GPU box) takes longer than the device step (8.5 ms): two loader
        # threads prepare batches i+1 and i+2 side by side (gf_smp_prepare releases the GIL; a worker pool per calling thread),
        # four handles rotate, the main thread only launches.
        import threading
        NH, NLOAD = 4, 2
        nets = [net] + [SMPOmega(L, C, F, D, cap, True, ctx=ctx) for _ in range(NH - 1)]
        for n in nets[1:]:
            n.set_fused(fused)
        p = params.clone()
        for k, n in enumerate(nets):   # warm every handle's pools
            n.prepare(batches[k % 4][0])
            n.forward(p, batches[k % 4][1])
            n.backward(p, grads)
        if ctx.dist_world > 1:
            ctx.dist_quiesce()
        torch.cuda.synchronize()
        ready = [threading.Semaphore(0) for _ in range(NH)]
        free = [threading.Semaphore(1) for _ in range(NH)]
        errors = []

        def loader(t):
            try:
                torch.cuda.set_device(dev)
                for it in range(t, steps, NLOAD):
                    h = it % NH
                    free[h].acquire()
                    nets[h].prepare(batches[it % 4][0])
                    ready[h].release()
            except Exception as e:   # noqa: BLE001  (reported by the main thread)
                errors.append(e)
                for r in ready:
                    r.release()

        threads = [threading.Thread(target=loader, args=(t,), daemon=True) for t in range(NLOAD)]
        t0 = time.perf_counter()
        for th in threads:
            th.start()
        for it in range(steps):
            h = it % NH
            ready[h].acquire()
            if errors:
                raise errors[0]
            cur = nets[h]
            cur.forward(p, batches[it % 4][1])
            cur.backward(p, grads)
            net.adam_step(p, grads, 1e-5, B * world)
            free[h].release()   # (gf_smp_prepare waits for the handle's own last launch before it recycles its buffers)
        if ctx.dist_world > 1:
            ctx.dist_quiesce()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        for th in threads:
            th.join()
        net.prepare(mols)   # leave the handle as the timed region had it
        for n in nets[1:]:
            n.close()
        return {"ms_per_step": round(dt * 1e3, 3), "value": round(B / dt, 1), "unit": "molecules/s per GPU", "steps": steps,
                "what": "new batch every step: host graph preparation + upload (two loader threads, four handles, overlapped) + forward + backward + Adam"}

    # what the arithmetic is: fp32 everywhere; with the split products (default at 64 / 32 / 16 channels) the level's block products take
    # each fp32 operand as two f16 halves (22 significant bits) on v_mfma_f32_32x32x16_f16 and accumulate in fp32;
    # extra.cfg3_fp32_products is the same step with those products on the fp32 MFMA pipe
    dtype = "f32 (level block products: 2xf16-split operands, 22-bit, fp32 accumulate)" if (split and fused) else "f32"
    meta = {"metric": "CCN-2D (SMP_omega) molecules/sec fwd+bwd", "unit": "molecules/s", "units_per_step": B, "dtype": dtype,
            "config": {"workload": "cfg3: SMP_omega 3 levels, C=%d, F=5, D=5, cap=29, batch=%d synthetic QM9-size molecules/GPU, device-resident, %s levels"
                                   % (C, B, "fused" if fused else "op-by-op"),
                       "parallelism": ("molecule-sharded x%d, RCCL all-reduce of the %d gradient floats inside gf_smp_backward (per-level segments, overlapped)"
                                       % (world, net.n_params)) if world > 1 else "single GPU",
                       "timed_region": "device step (index tables resident); see end_to_end for the loop with a new batch every step",
                       "prep_s": round(prep_s, 3), "prep_first_s": round(prep_first_s, 3), "work": work}}
    return step, finish, cpu, meta, (net, end_to_end)


# ---- launcher ---------------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_spawn(args):
    """`python bench.py --gpus N` outside a launcher: re-run this script as N ranks on this node."""
    if not args.plumbing and not os.environ.get("GF_BENCH_ONE_GPU"):
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit("bench.py: --gpus %d but only %d HIP device(s) are visible; refusing to report a smaller run as n_gpus=%d"
                     % (args.gpus, have, args.gpus))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def run_plumbing(args, world, rank):
    """CPU-only check of everything around the kernels for N > 1: the launcher, rank/shard bookkeeping, the single exchange
    (gloo all-reduce of a gradient-sized buffer), barrier + max-over-ranks timing and the one JSON line from rank 0.
    No kernels run and nothing is measured: the line says so."""
    import torch
    from graphflow_amd import dist as gd
    dist = gd.init(backend="gloo")
    n_params = 64 * 30 + 3 * (18 * 64 * 64 + 64) + 64
    lo, hi = gd.shard(1024 * world, rank, world)
    g = torch.full((n_params,), float(rank + 1))

    def step():
        g.fill_(float(rank + 1))
        gd.allreduce_sum_(g, dist)

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    elapsed = gd.max_over_ranks(time.perf_counter() - t0, dist)
    assert float(g[0]) == world * (world + 1) / 2, "all-reduce over %d ranks gave %r" % (world, float(g[0]))
    if rank == 0:
        print(json.dumps({"metric": "plumbing check (no kernels ran, nothing measured)", "value": 0.0, "unit": "molecules/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
                          "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "none",
                          "config": {"workload": "plumbing: launcher + sharding + gloo all-reduce of %d floats" % n_params,
                                     "shard_of_rank0": [lo, hi], "scaling_requested": args.scaling,
                                     "per_rank_units": (args.batch or 8192) // world if args.scaling == "strong" else (args.batch or 1024)},
                          "collective": {"rccl_ranks_seen": None, "gloo_ranks_seen": dist.get_world_size() if dist is not None else 1, "expected": world,
                                         "allreduces_per_step": 1, "allreduce_ms_per_step": None}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the line reports the median and the spread")
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "cfg5"])
    ap.add_argument("--batch", type=int, default=0, help="graphs / molecules per GPU (default 256 for cfg2/cfg5, 1024 for cfg3)")
    ap.add_argument("--N", type=int, default=32)
    ap.add_argument("--C", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the cfg2 / cfg5 / end_to_end sections of the default line")
    ap.add_argument("--unfused", action="store_true", help="cfg3: op-by-op level pipeline instead of the fused level kernels")
    ap.add_argument("--plumbing", action="store_true", help="CPU-only check of the multi-rank plumbing (gloo); measures nothing")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch (default 1024 molecules / 256 graphs) PER GPU; strong: --batch (default 8192 / 2048) in TOTAL, split "
                         "evenly over the ranks (SURVEY 8d cfg4 asks for both)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_spawn(args)
    from graphflow_amd import dist as gd
    world, rank, local = gd.env_world()
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); they must agree" % (args.gpus, world))
    if args.plumbing:
        return run_plumbing(args, world, rank)

    import torch
    import graphflow_amd as gf

    # GF_BENCH_ONE_GPU=1 (TEST HOOK, tests/test_dist_gpu.py): every rank on device 0, torch's side channel on gloo, the path's exchange on
    # whatever GF_RCCL_LIBRARY names (the shared-memory mock: RCCL refuses two ranks on one device) -- this entry point's N > 1 code on a
    # one-GPU box; the line says so and its numbers mean nothing
    one_gpu = bool(os.environ.get("GF_BENCH_ONE_GPU")) and world > 1
    if not one_gpu and torch.cuda.device_count() < (local + 1 if world > 1 else 1):
        sys.exit("bench.py: rank %d needs HIP device %d, %d visible" % (rank, local, torch.cuda.device_count()))
    dev = torch.device("cuda", local if (world > 1 and not one_gpu) else 0)
    torch.cuda.set_device(dev)
    side = "cpu" if one_gpu else dev   # where the tensors of torch.distributed's own collectives live
    force = bool(os.environ.get("GF_FORCE_DIST"))
    dist = (gd.init(backend="gloo") if one_gpu else gd.init(backend="nccl", device=dev)) if world > 1 else None   # torch.distributed: barrier + max-over-ranks only
    ctx = gf.Context(dev.index)
    if world > 1 or force:   # the path's own exchange lives behind the C ABI: one RCCL communicator on the context
        uid = torch.zeros(128, dtype=torch.uint8, device=side)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(ctx.dist_unique_id()), dtype=torch.uint8))
        if dist is not None:
            dist.broadcast(uid, src=0)
        ctx.dist_init(bytes(uid.cpu().numpy().tobytes()), rank, world)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload in ("cfg2", "cfg5"):
        step, finish, cpu, meta, keep = setup_contraction(args.workload, args, torch, gf, dev, world, rank, ctx)
    else:
        step, finish, cpu, meta, keep = setup_smp(args, torch, gf, dev, world, rank, ctx)

    notiming = bool(os.environ.get("GF_BENCH_NOTIMING"))
    quiesce = ctx.dist_quiesce if (world > 1 or force) else None   # bounded wait for the collectives ahead of every blocking synchronize
    reps, timers = timed_run(torch, ctx, step, args.steps, args.warmup, fence, notiming, repeats=args.repeats, quiesce=quiesce)
    # every repeat is the contract's timed region (exactly --steps steps between barrier + synchronize): MAX over ranks per repeat, the
    # line's value is the MEDIAN repeat, the spread rides beside it
    reps = [gd.max_over_ranks(e, dist, side) for e in reps]
    elapsed = median(reps)

    line = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * meta["units_per_step"] * args.steps / elapsed
        line = {"metric": meta["metric"], "value": round(value, 1), "unit": meta["unit"], "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
                "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": meta.get("dtype", "f32"), "data": "synthetic",
                "config": meta["config"], "roofline": finish(timers, ms_per_step, args.steps)}
        line["timed_region"] = {"repeats": len(reps), "steps_each": args.steps, "reported": "median",
                                "ms_per_step_min": round(1e3 * min(reps) / args.steps, 4), "ms_per_step_median": round(ms_per_step, 4),
                                "ms_per_step_max": round(1e3 * max(reps) / args.steps, 4)}
        if one_gpu:
            line["data"] = "synthetic; TEST HOOK GF_BENCH_ONE_GPU: %d ranks share one GPU through a mock transport -- plumbing only, not a measurement" % world
        if ctx.dist_world > 1 or force:
            line["config"]["collective"] = "gf_dist_* (RCCL) world %d" % ctx.dist_world
            ar = timers.get("rccl_allreduce")
            # the exchange as the library saw it: ranks in ITS communicator (not torch's), and the all-reduces' own HIP-event time on
            # the communicator's stream -- they run beside the reverse sweep, so this is not a share of ms_per_step
            line["collective"] = {"rccl_ranks_seen": ctx.dist_world, "expected": world,
                                  "allreduces_per_step": round(ar[1] / args.steps, 2) if ar else None,
                                  "allreduce_ms_per_step": round(ar[0] / args.steps, 4) if ar else None,
                                  "watchdog_s": float(os.environ.get("GF_DIST_TIMEOUT_S", "1800")),
                                  "note": "HIP events around each ncclAllReduce on the communicator's stream (rank 0), live inside the timed region"}
        line["roofline"]["timing"] = ("HIP events on the kernels' stream: the dominant kernel live inside the timed region, "
                                      "the other kernels in an identical pass of the same steps just before it")
    # the sections below run on every rank that takes part in them, outside the timed region
    e2e = None
    if keep is not None and not args.no_extra:
        net, end_to_end = keep
        if rank == 0:
            line["config"]["device_GB"] = round(net.device_bytes()[0] / 1e9, 2)   # HBM held by the batch's buffers (lazy ones included)
        if not force:
            # every rank runs its own loader threads and handles; the loop's gradients are summed over the ranks inside backward, so
            # the ranks step together: the reported time is the slowest rank's
            e2e = end_to_end(min(40, max(8, args.steps)))
            if world > 1:
                dt = gd.max_over_ranks(e2e["ms_per_step"], dist, side)
                e2e.update(ms_per_step=round(dt, 3), value=round(world * meta["units_per_step"] / (dt * 1e-3), 1), unit="molecules/s, all %d GPUs" % world,
                           host_threads_per_rank=os.environ.get("GF_PREP_THREADS"))
    if rank == 0:
        if e2e:
            line["end_to_end"] = e2e
        if world == 1 and not args.no_extra and args.workload == "cfg3" and not args.unfused:
            # BASELINE configs[1] and configs[4] under the same clock: 20 steps each after 3 warm-ups
            extra = {}
            if args.C == 64 and os.environ.get("GF_SMP_SPLIT", "1") != "0":
                # the headline step again with the C = 64 block products on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32) instead of the
                # f16 pipe with two-half operands: the same step without the operand-width caveat, under the same clock
                from graphflow_amd import _lib
                try:   # (an extra section that fails is reported in its place: the headline line does not depend on it)
                    ctx.set_option(_lib.GF_OPT_SMP_FP32_PRODUCTS, 1)
                    els, _ = timed_run(torch, ctx, step, 20, 3, torch.cuda.synchronize, True, repeats=3)
                    el = median(els)
                    extra["cfg3_fp32_products"] = {"metric": meta["metric"], "value": round(meta["units_per_step"] * 20 / el, 1), "unit": meta["unit"],
                                                   "steps": 20, "warmup": 3, "ms_per_step": round(1e3 * el / 20, 4),
                                                   "what": "GF_OPT_SMP_FP32_PRODUCTS: the level's block products on the fp32 MFMA pipe, everything else as in the headline step"}
                except Exception as e:   # noqa: BLE001
                    extra["cfg3_fp32_products"] = {"error": repr(e)}
                finally:
                    ctx.set_option(_lib.GF_OPT_SMP_FP32_PRODUCTS, 0)
            if args.C == 64 and os.environ.get("GF_SMP_SPLIT", "1") != "0":
                # the same model at 32 channels (the reference runs any nChanels): the row-panel kernel family templated on the channel
                # count, weight gradients on smp_wgrad_direct<32>; its own context and handle, same molecules
                # ... and at 10 channels, the reference's own test models (tests/test_SMP_omega.cpp:22-34): computed with the channels
                # zero-padded to 16 on the device (32 until round 5), the caller's parameter / gradient layout kept at the C ABI
                import copy
                for Cx in (32, 10):
                    try:
                        a32 = copy.copy(args)
                        a32.C = Cx
                        c32 = gf.Context(dev.index)
                        s32, _, _, m32, k32 = setup_smp(a32, torch, gf, dev, 1, 0, c32)
                        els, _ = timed_run(torch, c32, s32, 20, 3, torch.cuda.synchronize, True, repeats=3)
                        el = median(els)
                        extra["cfg3_C%d" % Cx] = {"metric": m32["metric"], "value": round(m32["units_per_step"] * 20 / el, 1), "unit": m32["unit"],
                                                  "steps": 20, "warmup": 3, "ms_per_step": round(1e3 * el / 20, 4), "workload": m32["config"]["workload"]}
                        if k32 is not None:
                            k32[0].close()
                        del s32, k32, c32
                    except Exception as e:   # noqa: BLE001
                        extra["cfg3_C%d" % Cx] = {"error": repr(e)}
                # ... and the RisiContraction_10 / _50 wirings (SMP_2D_ver6 / ver7) at the reference's 10 channels: since round 5 embedded in
                # the fused 18-slice level on [f | f^T] channels
                for name, nK in (("cfg3_ver6_C10", 10), ("cfg3_ver7_C10", 50)):
                    try:
                        a7 = copy.copy(args)
                        a7.C, a7.nK = 10, nK
                        c7 = gf.Context(dev.index)
                        s7, _, _, m7, k7 = setup_smp(a7, torch, gf, dev, 1, 0, c7)
                        els, _ = timed_run(torch, c7, s7, 20, 3, torch.cuda.synchronize, True, repeats=3)
                        el = median(els)
                        extra[name] = {"metric": "molecules/sec fwd+bwd, SMP_2D_ver%d wiring (RisiContraction_%d per node)" % (6 if nK == 10 else 7, nK),
                                       "value": round(m7["units_per_step"] * 20 / el, 1), "unit": m7["unit"], "steps": 20, "warmup": 3,
                                       "ms_per_step": round(1e3 * el / 20, 4), "workload": m7["config"]["workload"].replace("SMP_omega", "SMP_2D_ver%d" % (6 if nK == 10 else 7))}
                        if k7 is not None:
                            k7[0].close()
                        del s7, k7, c7
                    except Exception as e:   # noqa: BLE001
                        extra[name] = {"error": repr(e)}
            if args.C == 64:
                # SMP_beta (no receptive-field cap) on molecules larger than QM9's: 256 synthetic 44..48-atom molecules, whose level-3 fields
                # reach 36 - 41 positions for one or two nodes each.  Round 6 keeps such a level on the fused kernels; the same
                # step with GF_SMP_BIG_FIELDS=0 (the level op by op, as until round 5) rides beside it.
                try:
                    from graphflow_amd.smp import SMPOmega
                    from inputs import smp_params, synthetic_molecule
                    import numpy as np
                    nvb, nb = 48, 256
                    bm = [synthetic_molecule(i, nV=nvb - i % 5)[:2] for i in range(nb)]
                    bctx = gf.Context(dev.index)
                    bnet = SMPOmega(3, 64, 5, 5, nvb, True, ctx=bctx)
                    bnet.prepare(bm)
                    bp = torch.as_tensor(smp_params(64, 5, 5, 3, 1).astype(np.float32)).to(dev)
                    bt = torch.as_tensor(np.array([float(len(a)) for a, _ in bm], dtype=np.float32)).to(dev)
                    bg = torch.empty(bnet.n_params, device=dev)

                    def bstep():
                        bnet.forward(bp, bt)
                        bnet.backward(bp, bg)

                    els, _ = timed_run(torch, bctx, bstep, 20, 3, torch.cuda.synchronize, True, repeats=3)
                    el = median(els)
                    os.environ["GF_SMP_BIG_FIELDS"] = "0"
                    try:
                        els0, _ = timed_run(torch, bctx, bstep, 5, 2, torch.cuda.synchronize, True, repeats=1)
                    finally:
                        del os.environ["GF_SMP_BIG_FIELDS"]
                    extra["beta_48_atoms"] = {"metric": "molecules/sec fwd+bwd, SMP_beta wiring (no receptive-field cap)", "value": round(nb * 20 / el, 1),
                                              "unit": "molecules/s", "steps": 20, "warmup": 3, "ms_per_step": round(1e3 * el / 20, 4),
                                              "ms_per_step_level_op_by_op": round(1e3 * median(els0) / 5, 4),
                                              "workload": "SMP_beta 3 levels, C=64, F=5, D=5, no cap, batch=%d synthetic %d..%d-atom molecules (level sizes %s)"
                                                          % (nb, nvb - 4, nvb, [bnet.level_sizes(l)[:2] for l in range(4)])}
                    bnet.close()
                    del bnet, bctx
                    torch.cuda.empty_cache()
                except Exception as e:   # noqa: BLE001
                    extra["beta_48_atoms"] = {"error": repr(e)}
            for wl in ("cfg2", "cfg5"):
                try:
                    ectx = gf.Context(dev.index)
                    estep, efinish, ecpu, emeta, _ = setup_contraction(wl, args, torch, gf, dev, 1, 0, ectx)
                    els, et = timed_run(torch, ectx, estep, 20, 3, torch.cuda.synchronize, notiming, repeats=3)
                    el = median(els)
                    ems = 1e3 * el / 20
                    extra[wl] = {"metric": emeta["metric"], "value": round(emeta["units_per_step"] * 20 / el, 1), "unit": emeta["unit"],
                                 "steps": 20, "warmup": 3, "ms_per_step": round(ems, 4), "workload": emeta["config"]["workload"],
                                 "roofline": efinish(et, ems, 20)}
                    if not args.no_cpu_baseline:
                        cb = ecpu()
                        if cb:
                            extra[wl]["cpu_baseline"] = cb
                            extra[wl]["speedup_vs_cpu_1core"] = round(extra[wl]["value"] / cb["value"], 1)
                    del estep
                    ectx.close()
                    torch.cuda.empty_cache()
                except Exception as e:   # noqa: BLE001
                    extra[wl] = {"error": repr(e)}
            line["extra"] = extra
        try:
            line["roofline"]["hbm_copy_measured_GBps"] = round(copy_ceiling_gbps(torch, dev), 1)   # torch's copy_
        except Exception as e:   # noqa: BLE001
            line["roofline"]["hbm_copy_measured_GBps"] = None
            line["roofline"]["hbm_copy_error"] = repr(e)
        try:
            torch.cuda.empty_cache()
            line["roofline"]["hbm_copy_kernel_GBps"] = copy_probe_gbps(torch, dev, ctx)           # the library's own float4 copies
            line["roofline"]["hbm_copy_note"] = ("read + written bytes / time of a 1 GiB copy on this box: torch copy_ (hbm_copy_measured_GBps) and "
                                                 "gf_hbm_copy_probe_f32 (hbm_copy_kernel_GBps); MI355X_MICROARCH.md quotes 6290 GB/s for a float4 copy")
        except Exception as e:   # noqa: BLE001
            line["roofline"]["hbm_copy_kernel_GBps"] = None
            line["roofline"]["hbm_copy_kernel_error"] = repr(e)
        try:
            line["cpu_baseline"] = cpu() if (world == 1 and not args.no_cpu_baseline) else None
        except Exception as e:   # noqa: BLE001  (the measured GPU line is printed whatever happens to the host-side baseline)
            line["cpu_baseline"] = {"error": repr(e)}
    # RCCL prints its version banner through C stdio, which is block-buffered on a pipe: every rank pushes it out BEFORE rank 0
    # writes the JSON line, so that the line is the last thing the job writes to stdout
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
    # tear the path's own communicator down while every rank is still here (ncclCommDestroy wants its peers alive), bounded by the
    # watchdog; a rank that cannot is reported on stderr -- the line above is already out
    if world > 1 or force:
        try:
            ctx.dist_quiesce()
            ctx.dist_finalize()
        except Exception as e:   # noqa: BLE001
            print("bench.py: rank %d: tearing down the RCCL communicator failed: %r" % (rank, e), file=sys.stderr, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
