"""SMP_omega forward + backward on molecules LARGER than the fused levels' receptive-field cap (32): per-kernel times of one step.
usage: python tools/big_fields_time.py [nV] [cap] [C] [batch]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from inputs import synthetic_molecule, smp_params
import graphflow_amd as gf
from graphflow_amd.smp import SMPOmega

nV = int(sys.argv[1]) if len(sys.argv) > 1 else 45
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 45
Cn = int(sys.argv[3]) if len(sys.argv) > 3 else 64
B = int(sys.argv[4]) if len(sys.argv) > 4 else 128
L, F, D = 3, 5, 2
mols = [synthetic_molecule(i, nV=nV - i % 5)[:2] for i in range(B)]
tg = torch.tensor(np.array([float(nV)] * B, dtype=np.float32), device="cuda")
net = SMPOmega(L, Cn, F, D, cap, True)
p = torch.tensor(smp_params(Cn, F, D, L, 1).astype(np.float32), device="cuda")
g = torch.empty(net.n_params, device="cuda")
net.prepare(mols)
print("levels (nodes, rows, ppos):", [net.level_sizes(l) for l in range(L + 1)])
for _ in range(2):
    net.forward(p, tg); net.backward(p, g)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    net.forward(p, tg); net.backward(p, g)
torch.cuda.synchronize()
print("nV %d cap %d C %d batch %d: %.2f ms per forward + backward" % (nV, cap, Cn, B, (time.perf_counter() - t0) / n * 1e3))
ctx = net.ctx
ctx.set_timing(True)
net.forward(p, tg); net.backward(p, g)
torch.cuda.synchronize()
rows = sorted(ctx.timings().items(), key=lambda kv: -kv[1][0]) if hasattr(ctx, "timings") else []
for k, v in rows[:12]:
    print("  %-28s %8.3f ms  %d launches" % (k, v[0], v[1]))
