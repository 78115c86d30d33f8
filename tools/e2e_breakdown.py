"""Where the host time of the end-to-end loop goes: launch calls vs gf_smp_prepare, per iteration.  usage: python tools/e2e_breakdown.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from inputs import smp_params, synthetic_molecule  # noqa: E402
from graphflow_amd.smp import SMPOmega  # noqa: E402

B, L, C, F, D, cap = 1024, 3, 64, 5, 5, 29
pool = [synthetic_molecule(i) for i in range(4 * B)]
batches = []
for i in range(4):
    sl = pool[i * B:(i + 1) * B]
    batches.append((SMPOmega.pack([(a, f) for a, f, _ in sl]), torch.as_tensor(np.array([t for *_, t in sl], dtype=np.float32)).cuda()))
p = torch.as_tensor(smp_params(C, F, D, L, 1).astype(np.float32)).cuda()
g = torch.empty_like(p)
nets = [SMPOmega(L, C, F, D, cap, True) for _ in range(2)]
for k, n in enumerate(nets):
    n.prepare(batches[k][0])
    n.forward(p, batches[k][1])
    n.backward(p, g)
torch.cuda.synchronize()
nets[0].prepare(batches[0][0])
tl = tp = 0.0
t00 = time.perf_counter()
for it in range(40):
    cur = nets[it % 2]
    t0 = time.perf_counter()
    cur.forward(p, batches[it % 4][1])
    cur.backward(p, g)
    nets[0].adam_step(p, g, 1e-5, B)
    t1 = time.perf_counter()
    nets[(it + 1) % 2].prepare(batches[(it + 1) % 4][0])
    t2 = time.perf_counter()
    tl += t1 - t0
    tp += t2 - t1
torch.cuda.synchronize()
tot = time.perf_counter() - t00
print("per iteration: launches %.2f ms, prepare %.2f ms, total %.2f ms" % (tl / 40 * 1e3, tp / 40 * 1e3, tot / 40 * 1e3))
