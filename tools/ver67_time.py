"""SMP_2D_ver6 / ver7 wirings (RisiContraction_10 / _50 per node, op by op) on the cfg3 batch: ms per forward + backward.
usage: python tools/ver67_time.py nContractions C [batch] [nV: molecules of nV - 4 .. nV atoms, no cap]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from inputs import synthetic_molecule
from graphflow_amd.smp import SMPOmega
nK = int(sys.argv[1]); Cn = int(sys.argv[2]); B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
nV = int(sys.argv[4]) if len(sys.argv) > 4 else 0
L, F, D, cap = 3, 5, 2, (nV or 29)
mols = [(synthetic_molecule(i, nV=nV - i % 5) if nV else synthetic_molecule(i))[:2] for i in range(B)]
tg = torch.tensor(np.array([synthetic_molecule(i)[2] for i in range(B)], dtype=np.float32), device="cuda")
net = SMPOmega(L, Cn, F, D, cap, True, nContractions=nK, custom_matmul=(nK != 18))
p = torch.tensor((np.random.default_rng(1).uniform(-1, 1, net.n_params) / np.sqrt(nK * Cn)).astype(np.float32), device="cuda")
g = torch.empty(net.n_params, device="cuda")
net.prepare(mols)
for _ in range(2):
    net.forward(p, tg); net.backward(p, g)
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    net.forward(p, tg); net.backward(p, g)
torch.cuda.synchronize()
print("nContractions %d C %d batch %d GF_SMP_PAD_CHANNELS=%s: %.2f ms per forward + backward, |g| max %.3e" %
      (nK, Cn, B, os.environ.get("GF_SMP_PAD_CHANNELS", "1"), (time.perf_counter() - t0) / n * 1e3, float(g.abs().max())), flush=True)
