"""Gradients of one SMP_omega step at C channels, computed at 16 and at 32 padded channels: per parameter block, max |diff| / max |g|.
usage: python tools/pad_compare.py [C] [batch] [scale]"""
import os, sys, subprocess, json
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from inputs import synthetic_molecule
    from graphflow_amd.smp import SMPOmega
    Cn, B, scale, out = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), sys.argv[5]
    L, F, D, cap = 2, 5, 4, 10
    mols = [synthetic_molecule(i)[:2] for i in range(B)]
    tg = torch.tensor(np.array([synthetic_molecule(i)[2] for i in range(B)], dtype=np.float32), device="cuda")
    net = SMPOmega(L, Cn, F, D, cap, True)
    p = torch.tensor((np.random.default_rng(1).uniform(-1, 1, net.n_params) * scale).astype(np.float32), device="cuda")
    g = torch.empty(net.n_params, device="cuda")
    net.prepare(mols)
    net.forward(p, tg); net.backward(p, g)
    torch.cuda.synchronize()
    np.save(out, g.cpu().numpy())
    sys.exit(0)
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
res = {}
for tag, env in (("p16", {}), ("p32", {"GF_SMP_PAD_CHANNELS": "3"}), ("p16_old", {"GF_SMP_WGRAD_ALL": "0"}), ("unfused", {"GF_SMP_PAD_CHANNELS": "0"})):
    out = "/tmp/g_%s.npy" % tag
    subprocess.run([sys.executable, __file__, "--child", str(Cn), str(B), str(scale), out], env=dict(os.environ, **env), check=True)
    res[tag] = np.load(out).astype(np.float64)
L, F, D = 2, 5, 4
FD = F * (D + 1) if True else F
n = len(res["p16"])
# blocks: H [C][FD'] | (K_l [18 C][C], b_l [C]) x L | W [C]
rest = n - L * (18 * Cn * Cn + Cn) - Cn
blocks = [("H", 0, rest)]
o = rest
for l in range(1, L + 1):
    for k in range(18):
        blocks.append(("K%d[%d]" % (l, k), o, o + Cn * Cn)); o += Cn * Cn
    blocks.append(("b%d" % l, o, o + Cn)); o += Cn
blocks.append(("W", o, o + Cn))
ref = res["unfused"]
print("block            max|g|      p16-unf    p32-unf    p16old-unf   (relative to the block's max)")
for name, a, b in blocks:
    m = np.abs(ref[a:b]).max() + 1e-300
    print("%-12s %10.3e  %10.2e %10.2e %10.2e" % (name, m, np.abs(res["p16"][a:b] - ref[a:b]).max() / m, np.abs(res["p32"][a:b] - ref[a:b]).max() / m,
                                                 np.abs(res["p16_old"][a:b] - ref[a:b]).max() / m))
