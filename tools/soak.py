"""Soak: a training loop that prepares a NEW batch every step (varying size) on one handle; checks that device memory stays
bounded (pooled buffers) and that nothing goes non-finite.  usage: python tools/soak.py [steps] [C] [nContractions]"""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np, torch
from inputs import synthetic_molecule, smp_params
from graphflow_amd.smp import SMPOmega
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
L, C, F, D, cap = 3, (int(sys.argv[2]) if len(sys.argv) > 2 else 64), 5, 5, 29
nK = int(sys.argv[3]) if len(sys.argv) > 3 else 18
net = SMPOmega(L, C, F, D, cap, True, nContractions=nK, custom_matmul=(nK != 18))
p = torch.as_tensor((smp_params(C, F, D, L, 1) if nK == 18 else np.random.default_rng(1).uniform(-1, 1, net.n_params) / np.sqrt(nK * C)).astype(np.float32)).cuda()
g = torch.empty_like(p)
pool = [synthetic_molecule(i) for i in range(2048)]
rng = np.random.default_rng(0)
free0 = None
t0 = time.perf_counter()
for it in range(steps):
    nb = int(rng.integers(200, 1025))
    idx = rng.choice(len(pool), nb, replace=False)
    mols = [(pool[i][0], pool[i][1]) for i in idx]
    tg = torch.as_tensor(np.array([pool[i][2] for i in idx], dtype=np.float32)).cuda()
    net.prepare(mols)
    _, loss, _ = net.forward(p, tg)
    net.backward(p, g)
    net.adam_step(p, g, 1e-4, nb)
    if it % 10 == 9 or it == steps - 1:
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        l = float(loss.mean())
        assert np.isfinite(l) and bool(torch.isfinite(p).all())
        if free0 is None:
            free0 = free
        print(f"step {it+1:4d}  batch {nb:4d}  mean loss {l:10.4f}  device memory in use {(total-free)/2**30:6.2f} GiB  {(time.perf_counter()-t0)/(it+1)*1e3:6.1f} ms/step incl. prepare")
print("OK")
