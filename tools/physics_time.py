"""Forward + backward of the `_physics` / `_pairgraphs` models (gf_smp_model_*) on the cfg3 batch: time per step with the towers computed at
their padded width (fused level kernels, default) and at their own halving widths (GF_SMP_PAD_CHANNELS=0, op-by-op levels).
usage: python tools/physics_time.py [towers] [C] [batch] [nKept]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from inputs import synthetic_molecule
from graphflow_amd.smp import SMPModel

towers = int(sys.argv[1]) if len(sys.argv) > 1 else 1
Cn = int(sys.argv[2]) if len(sys.argv) > 2 else 32
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
nKept = int(sys.argv[4]) if len(sys.argv) > 4 else 0
L, cap, F = 3, 29, 5
mols = [synthetic_molecule(i)[:2] for i in range(B)]
tg = torch.tensor(np.array([synthetic_molecule(i)[2] for i in range(B)], dtype=np.float32), device="cuda")
for mode in ("1", "0"):
    os.environ["GF_SMP_PAD_CHANNELS"] = mode
    net = SMPModel(L, Cn, cap, [F] * towers, nKept=nKept)
    p = torch.tensor(np.random.default_rng(1).uniform(-0.2, 0.2, net.n_params).astype(np.float32), device="cuda")
    g = torch.empty(net.n_params, device="cuda")
    net.prepare(mols, mols if towers == 2 else None)
    net.set_mode(True)
    for it in range(3):
        net.forward(p, tg); net.backward(p, g)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for it in range(n):
        net.forward(p, tg); net.backward(p, g)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print("towers %d C %d batch %d nKept %d GF_SMP_PAD_CHANNELS=%s: %.2f ms per forward + backward (%.0f samples/s), |g| max %.3e" %
          (towers, Cn, B, nKept, mode, ms, B / ms * 1e3, float(g.abs().max())), flush=True)
    net.close()
if os.environ.get("GF_PHYS_KERNELS"):   # per-kernel times of one step of the last configuration's twin (default padding)
    os.environ["GF_SMP_PAD_CHANNELS"] = "1"
    net = SMPModel(L, Cn, cap, [F] * towers, nKept=nKept)
    p = torch.tensor(np.random.default_rng(1).uniform(-0.2, 0.2, net.n_params).astype(np.float32), device="cuda")
    g = torch.empty(net.n_params, device="cuda")
    net.prepare(mols, mols if towers == 2 else None)
    net.set_mode(True)
    net.forward(p, tg); net.backward(p, g)
    net.ctx.set_timing(True)
    net.forward(p, tg); net.backward(p, g)
    torch.cuda.synchronize()
    for k, v in sorted(net.ctx.timings().items(), key=lambda kv: -kv[1][0])[:14]:
        print("  %-28s %8.3f ms  %d launches" % (k, v[0], v[1]))
