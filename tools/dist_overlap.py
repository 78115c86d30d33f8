"""Does the gradient exchange of a data-parallel step hide behind the reverse sweep?  (round-3 review, next-round item 9)

Two modes.

  python tools/dist_overlap.py run [--steps K]
      cfg3 (1024 molecules, 3 levels, C = 64) forward + backward K times WITHOUT and WITH a communicator on the context (one rank:
      RCCL refuses two ranks on one device; the collectives still launch their kernels on the communicator's stream, behind the
      events gf_smp_backward records, and are joined before it returns).  Prints ms per step of both and the penalty.  Run it under
      rocprofv3 --kernel-trace --output-format csv -d DIR to get the timeline for the second mode.

  python tools/dist_overlap.py report DIR
      From the kernel trace of a `run`: for every collective kernel of the LAST step, which kernels of the sweep were executing
      while it ran (same device clock), and what fraction of its duration was covered by them.
"""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def run(steps):
    import numpy as np
    import torch
    import graphflow_amd as gf
    from graphflow_amd.smp import SMPOmega
    from inputs import smp_params, synthetic_molecule
    L, C, F, D, cap, B = 3, 64, 5, 5, 29, 1024
    mols, tg = [], []
    for seed in range(B):
        a, f, t = synthetic_molecule(seed)
        mols.append((a, f))
        tg.append(t)
    p = torch.as_tensor(smp_params(C, F, D, L, 1).astype(np.float32)).cuda()
    t = torch.as_tensor(np.array(tg, dtype=np.float32)).cuda()
    out = {}
    for name in ("plain", "communicator"):
        ctx = gf.Context(0)
        if name == "communicator":
            ctx.dist_init(ctx.dist_unique_id(), 0, 1)
        net = SMPOmega(L, C, F, D, cap, True, ctx=ctx)
        net.prepare(mols)
        g = torch.empty(net.n_params, device="cuda")
        for _ in range(5):
            net.forward(p, t)
            net.backward(p, g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net.forward(p, t)
            net.backward(p, g)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / steps * 1e3
        out[name + "_grad"] = g.clone()
        net.close()
        ctx.close()
    same = bool(torch.equal(out["plain_grad"], out["communicator_grad"]))
    print("cfg3 step, %d steps: %.3f ms without a communicator, %.3f ms with the four segment all-reduces on the communicator's stream "
          "(penalty %+.2f %%); gradients bit-identical: %s" % (steps, out["plain"], out["communicator"],
                                                              100 * (out["communicator"] / out["plain"] - 1), same))
    assert same


def report(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", ""),
                         r.get("Queue_Id", r.get("Stream_Id", "?"))))
    rows.sort()
    def is_coll(n):
        n = n.lower()
        return ("nccl" in n or "rccl" in n or "allreduce" in n) and "rocclr" not in n
    coll = [i for i, r in enumerate(rows) if is_coll(r[2])]
    if not coll:
        names = sorted({r[2].split("(")[0] for r in rows})
        print("# %d kernel dispatches traced, %d distinct kernels, none of them a collective: RCCL executes a ONE-rank in-place all-reduce as a\n"
              "# no-op (no kernel is launched), so on one GPU the trace can only show that the segment exchanges' stream choreography (an event\n"
              "# after each level's fold, the collective on the communicator's stream, a join before gf_smp_backward returns) costs nothing:\n"
              "# see the step times above.  Kernels on the timeline of the last step run back to back on one queue." % (len(rows), len(names)))
        return
    # the collectives of the last step: those after the last but one readout_dW ... simply the last four
    last = coll[-4:]
    print("# collective kernels of the last traced step and what ran beside them (rocprofv3 --kernel-trace, device timestamps)")
    for i in last:
        s, e, n, q = rows[i]
        beside = []
        covered = 0
        for j, (s2, e2, n2, q2) in enumerate(rows):
            if j == i or e2 <= s or s2 >= e or j in coll:
                continue
            ov = min(e, e2) - max(s, s2)
            covered += ov
            beside.append("%s (%.1f us of its %.1f us)" % (n2.split("(")[0].split("::")[-1][:40], ov / 1e3, (e2 - s2) / 1e3))
        print("%-44s queue %s  %8.1f us   overlapped by the sweep for %.0f %% of its duration" %
              (n.split("(")[0][-44:], q, (e - s) / 1e3, min(100.0, 100.0 * covered / max(e - s, 1))))
        for b in beside[:6]:
            print("      beside: " + b)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 50
        run(steps)
