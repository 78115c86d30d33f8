"""Soak (round 6): one SMP_2D_ver6 handle whose batches alternate between embeddable (symmetric adjacency: the fused 18-slice level) and
not (one asymmetric molecule: the op-by-op `_10` levels), and one SMP_beta handle whose batches alternate between QM9-size molecules
and 44..48-atom ones (fields above 32 on the fused level): device memory stays bounded, nothing goes non-finite.
usage: python tools/soak_plans.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np, torch
from inputs import synthetic_molecule, smp_params
from graphflow_amd.smp import SMPOmega
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
L, F, D = 3, 5, 2
v6 = SMPOmega(L, 10, F, D, 29, True, nContractions=10, custom_matmul=True)
p6 = torch.as_tensor((np.random.default_rng(1).uniform(-1, 1, v6.n_params) / 10).astype(np.float32)).cuda()
beta = SMPOmega(L, 64, F, D, 48, True)
pb = torch.as_tensor(smp_params(64, F, D, L, 1).astype(np.float32)).cuda()
small = [synthetic_molecule(i) for i in range(512)]
big = [synthetic_molecule(i, nV=48 - i % 5) for i in range(96)]
rng = np.random.default_rng(0)
t0 = time.perf_counter()
for it in range(steps):
    idx = rng.choice(len(small), 200, replace=False)
    mols = [(small[i][0].copy(), small[i][1]) for i in idx]
    if it % 2:   # an asymmetric adjacency somewhere in the batch: the whole batch takes the op-by-op plan
        a = mols[3][0]
        i, j = np.argwhere(a > 0)[0]
        a[j, i] = 0
    tg = torch.as_tensor(np.array([small[i][2] for i in idx], dtype=np.float32)).cuda()
    g6 = torch.empty_like(p6)
    v6.prepare(mols)
    _, l6, _ = v6.forward(p6, tg)
    v6.backward(p6, g6)
    src = big if it % 2 else small[:96]
    gb = torch.empty_like(pb)
    beta.prepare([(m[0], m[1]) for m in src])
    _, lb, _ = beta.forward(pb, torch.as_tensor(np.array([m[2] for m in src], dtype=np.float32)).cuda())
    beta.backward(pb, gb)
    if it % 10 == 9 or it == steps - 1:
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        assert bool(torch.isfinite(g6).all()) and bool(torch.isfinite(gb).all()) and np.isfinite(float(l6.mean())) and np.isfinite(float(lb.mean()))
        print(f"step {it+1:4d}  device memory in use {(total-free)/2**30:6.2f} GiB  {(time.perf_counter()-t0)/(it+1)*1e3:6.1f} ms/step incl. prepare")
print("OK")
