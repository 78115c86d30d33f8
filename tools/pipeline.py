"""Training loop with a NEW batch every step: one handle (prepare and step alternate on the host) against two handles that
alternate, so that the host prepares batch i+1 while the device runs step i.  usage: python tools/pipeline.py [steps] [batch]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from inputs import smp_params, synthetic_molecule  # noqa: E402
from graphflow_amd.smp import SMPOmega  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
L, C, F, D, cap = 3, 64, 5, 5, 29
pool = [synthetic_molecule(i) for i in range(4 * B)]
batches = []
for i in range(4):
    sl = pool[i * B:(i + 1) * B]
    batches.append(([(a, f) for a, f, _ in sl], torch.as_tensor(np.array([t for *_, t in sl], dtype=np.float32)).cuda()))
p0 = torch.as_tensor(smp_params(C, F, D, L, 1).astype(np.float32)).cuda()


def run(nhandles):
    nets = [SMPOmega(L, C, F, D, cap, True) for _ in range(nhandles)]
    opt = SMPOmega(L, C, F, D, cap, True)   # holds the Adam moments only (a handle's optimiser state is its own)
    p, g = p0.clone(), torch.empty_like(p0)
    for n in nets:               # warm the pools
        n.prepare(batches[0][0])
        n.forward(p, batches[0][1])
        n.backward(p, g)
    torch.cuda.synchronize()
    nets[0].prepare(batches[0][0])
    t0 = time.perf_counter()
    for it in range(steps):
        cur = nets[it % nhandles]
        mols, tg = batches[it % 4]
        if nhandles == 1:
            cur.prepare(mols)
        cur.forward(p, tg)       # asynchronous launches
        cur.backward(p, g)
        opt.adam_step(p, g, 1e-5, B)
        if nhandles == 2:        # the host builds the next batch while the device works
            nets[(it + 1) % 2].prepare(batches[(it + 1) % 4][0])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return dt, p


t1, p1 = run(1)
t2, p2 = run(2)
same = bool(torch.equal(p1, p2))
print("one handle : %.2f ms per training step (prepare + step + Adam), %.0f molecules/s" % (t1 * 1e3, B / t1))
print("two handles: %.2f ms per training step (prepare overlapped),      %.0f molecules/s" % (t2 * 1e3, B / t2))
print("parameters after %d steps identical: %s" % (steps, same))
assert same
