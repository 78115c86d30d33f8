"""Per-parameter-segment error of the headline-shape golden (tests/golden/smp_headline.npz) for the fused / op-by-op paths and
the kernel-selection switches.  usage (GPU box): python tools/headline_diag.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from test_smp_cpu import headline_golden  # noqa: E402
from graphflow_amd.smp import SMPOmega  # noqa: E402

c, (L, C, D, cap), params = headline_golden()
F = c["feature"].shape[1]
ref = c["grads"].astype(np.float64)
FD = F * (D + 1)
segs, o = [("H", 0, C * FD)], C * FD
for l in range(1, L + 1):
    segs.append(("K%d" % l, o, o + 18 * C * C)); o += 18 * C * C
    segs.append(("b%d" % l, o, o + C)); o += C
segs.append(("W", o, o + C))


def run(fused, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    net = SMPOmega(L, C, F, D, cap, True)
    net.set_fused(fused)
    net.prepare([(c["adj"], c["feature"])])
    p = torch.as_tensor(params.astype(np.float32)).cuda()
    t = torch.as_tensor(c["target"].astype(np.float32)).cuda()
    net.forward(p, t)
    g = torch.empty(net.n_params, device="cuda")
    net.backward(p, g)
    for k in (env or {}):
        del os.environ[k]
    return g.cpu().numpy().astype(np.float64)


gmax = np.abs(ref).max()
for name, fused, env in (("fused", True, None), ("op-by-op", False, None), ("fused, two-kernel gather", True, {"GF_SMP_BWD_GATHER": "0"}),
                         ("fused, tiled GEMMs", True, {"GF_SMP_ROWPANEL": "0", "GF_SMP_WGRAD": "0"})):
    g = run(fused, env)
    print("%-28s total %.2e |" % (name, np.abs(g - ref).max() / gmax),
          "  ".join("%s %.1e/%.1e" % (n, np.abs(g[a:b] - ref[a:b]).max() / gmax, np.abs(ref[a:b]).max() / gmax) for n, a, b in segs))
# blocks of K_l in the fused run
g = run(True)
for l in range(1, L + 1):
    a = [s for s in segs if s[0] == "K%d" % l][0][1]
    e = [np.abs(g[a + k * C * C:a + (k + 1) * C * C] - ref[a + k * C * C:a + (k + 1) * C * C]).max() / gmax for k in range(18)]
    m = [np.abs(ref[a + k * C * C:a + (k + 1) * C * C]).max() / gmax for k in range(18)]
    print("K%d blocks err:" % l, " ".join("%.0e" % x for x in e))
    print("K%d blocks mag:" % l, " ".join("%.0e" % x for x in m))
print("--- individual switches")
for name, env in (("ROWPANEL=0", {"GF_SMP_ROWPANEL": "0"}), ("WGRAD=0", {"GF_SMP_WGRAD": "0"})):
    g = run(True, env)
    print("%-28s total %.2e |" % (name, np.abs(g - ref).max() / gmax),
          "  ".join("%s %.1e" % (n, np.abs(g[a:b] - ref[a:b]).max() / gmax) for n, a, b in segs))
for sp in ("1", "4", "64"):
    g = run(True, {"GF_WGRAD_SPLITS": sp})
    print("WGRAD_SPLITS=%-4s total %.2e" % (sp, np.abs(g - ref).max() / gmax))
