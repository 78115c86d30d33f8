"""SMP_2D_ver6 wiring: the fused 18-slice embedding against the op-by-op `_10` level, per parameter block.  usage: python tools/ver6_compare.py [C]"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
L, F, D, cap = 2, 5, 2, 10
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from inputs import synthetic_molecule
    from graphflow_amd.smp import SMPOmega
    Cn, out = int(sys.argv[2]), sys.argv[3]
    mols, tg = [], []
    for seed in range(160):
        adj, feat, t = synthetic_molecule(5000 + seed, nV=6 + seed % 5)
        mols.append((adj, feat)); tg.append(t)
    net = SMPOmega(L, Cn, F, D, cap, True, nContractions=10, custom_matmul=True)
    p = torch.tensor((np.random.default_rng(8).uniform(-1, 1, net.n_params) / np.sqrt(10 * Cn)).astype(np.float32), device="cuda")
    g = torch.empty(net.n_params, device="cuda")
    net.prepare(mols)
    pred, loss, feat = net.forward(p, torch.tensor(np.array(tg, dtype=np.float32), device="cuda")); net.backward(p, g)
    torch.cuda.synchronize()
    np.save(out, g.cpu().numpy()); np.save(out + ".pred.npy", pred.cpu().numpy())
    sys.exit(0)
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 32
res = {}
for tag, env in (("fused", {}), ("opbyop", {"GF_SMP_VER6_FUSED": "0"}), ("opbyop_tables", {"GF_SMP_VER6_FUSED": "0", "GF_FAM10_GRAPH": "0"})):
    out = "/tmp/v6_%s.npy" % tag
    subprocess.run([sys.executable, __file__, "--child", str(Cn), out], env=dict(os.environ, **env), check=True)
    res[tag] = np.load(out).astype(np.float64)
FD = F * (D + 1)
blocks = [("H", 0, Cn * FD)]
o = Cn * FD
for l in range(1, L + 1):
    blocks.append(("K%d" % l, o, o + 10 * Cn * Cn)); o += 10 * Cn * Cn
    blocks.append(("b%d" % l, o, o + Cn)); o += Cn
blocks.append(("W", o, o + Cn))
ref = res["opbyop_tables"]
top = np.abs(ref).max()
print("global max |g| %.3e" % top)
for name, a, b in blocks:
    m = np.abs(ref[a:b]).max() + 1e-300
    d1 = np.abs(res["fused"][a:b] - ref[a:b]); d2 = np.abs(res["opbyop"][a:b] - ref[a:b])
    print("%-4s max|g| %10.3e   fused-ref %9.2e (of block) %9.2e (of global)   opbyop-ref %9.2e   n(|d| > 1e-6 global) %d" %
          (name, m, d1.max() / m, d1.max() / top, d2.max() / m, int((d1 > 1e-6 * top).sum())))
