"""Fused vs op-by-op gradients of a batch with fields above 32 positions, PARAMETER BLOCK BY BLOCK (H, the 18 slices of every K_l, b_l, W):
the comparison that found the lost rows of K_3 slice 11 in round 6 (NOTES.md) -- the global norm hid them.  usage: python tools/grad_blocks.py [C]"""
import sys, os, numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from inputs import synthetic_molecule, smp_params
from graphflow_amd.smp import SMPOmega
C = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nV, L, F, D = 48, 3, 5, 2
mols, tg = [], []
for seed in (8017, 8003, 8026, 8028, 8000):
    a, f, t = synthetic_molecule(seed, nV=nV); mols.append((a, f)); tg.append(t)
for i in range(7):
    a, f, t = synthetic_molecule(8100 + i); mols.append((a, f)); tg.append(t)
params = smp_params(C, F, D, L, 11)
def run(fused, sel=None):
    net = SMPOmega(L, C, F, D, nV, True); net.set_fused(fused)
    ms = mols if sel is None else [mols[i] for i in sel]
    t = np.array(tg if sel is None else [tg[i] for i in sel], dtype=np.float32)
    net.prepare(ms)
    p = torch.as_tensor(params.astype(np.float32)).cuda()
    pred, loss, feat = net.forward(p, torch.as_tensor(t).cuda())
    g = torch.zeros(net.n_params, device="cuda"); net.backward(p, g)
    return pred.cpu().numpy().astype(np.float64), g.cpu().numpy().astype(np.float64)
def blocks():
    FD = F * (D + 1); b = [("H", 0, C * FD)]; o = C * FD
    for l in range(1, L + 1):
        for k in range(18):
            b.append(("K%d.%d" % (l, k), o, o + C * C)); o += C * C
        b.append(("b%d" % l, o, o + C)); o += C
    b.append(("W", o, o + C)); return b
for sel in (None, [0], [5, 6, 7]):
    pa, ga = run(True, sel); pb, gb = run(False, sel)
    sc = max(np.abs(gb).max(), 1.0)
    print("sel", sel, "pred", np.abs(pa - pb).max() / max(np.abs(pb).max(), 1), "grad", np.abs(ga - gb).max() / sc)
    worst = sorted(((np.abs(ga[a:b] - gb[a:b]).max() / sc, np.abs(ga[a:b] - gb[a:b]).max() / max(np.abs(gb[a:b]).max(), 1e-30), n) for n, a, b in blocks()), reverse=True)[:8]
    for w in worst: print("   %-8s abs/global %.2e  rel/block %.2e" % (w[2], w[0], w[1]))
