"""Folded backward gather vs the two-kernel path over channel counts / caps (GPU).  Usage: python tools/check_gather.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from inputs import smp_params, synthetic_molecule  # noqa: E402
from graphflow_amd.smp import SMPOmega  # noqa: E402


def grads(mols, tg, C, L, cap, flag):
    os.environ["GF_SMP_BWD_GATHER"] = flag
    net = SMPOmega(L, C, 5, 2, cap, True)
    net.prepare(mols)
    p = torch.as_tensor(smp_params(C, 5, 2, L, 3).astype(np.float32)).cuda()
    net.forward(p, torch.as_tensor(np.array(tg, dtype=np.float32)).cuda())
    g = torch.empty(net.n_params, device="cuda")
    net.backward(p, g)
    return g.cpu().numpy().astype(np.float64)


worst = 0.0
for C in (4, 8, 20, 32, 64, 128, 256):
    for L, cap in ((2, 29), (3, 12), (3, 29)):
        mols, tg = [], []
        for seed in range(6):
            adj, feat, t = synthetic_molecule(40 + seed)
            mols.append((adj, feat))
            tg.append(t)
        g1, g0 = grads(mols, tg, C, L, cap, "1"), grads(mols, tg, C, L, cap, "0")
        err = float(np.abs(g1 - g0).max() / max(np.abs(g0).max(), 1e-30))
        worst = max(worst, err)
        print("C=%d L=%d cap=%d  rel diff %.2e" % (C, L, cap, err), flush=True)
assert worst <= 1e-6, worst
print("ok")
