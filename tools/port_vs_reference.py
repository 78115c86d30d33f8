#!/usr/bin/env python3
"""Build container only (needs /root/reference): run time of the CPU PORT (oracle/gf_oracle.c + oracle/smp_port.c, what
bench.py times on the GPU box as cpu_baseline kind "port") beside the REAL reference compiled from /root/reference
(oracle/_ref/libgf_ref.so), on the same inputs, same host, same compiler flags (-O2).  BASELINE.md section 4 step 2: the
port is trusted as the GPU-box baseline once the two agree within about 10 %.
usage: python tools/port_vs_reference.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from inputs import cfg_graph, smp_params, synthetic_molecule  # noqa: E402
from oracle import pyoracle  # noqa: E402


def main():
    out = {"host": {"cores": os.cpu_count()}, "flags": "-O2 (reference README:12), fp64"}
    ref = pyoracle.reference()
    assert ref is not None, "oracle/_ref/libgf_ref.so missing (build container only)"
    orc = pyoracle.oracle()
    # cfg2 shape: RisiContraction_18 fwd+bwd, one graph, N=32, C=64
    N, C = 32, 64
    P, A, G = cfg_graph(N, C, 1000, K=18)
    t_ref, kind, o_ref, d_ref = pyoracle.time_r18_fwd_bwd(P, A, G, prefer_reference=True)
    t_port, kind2, o_port, d_port = pyoracle.time_r18_fwd_bwd(P, A, G, prefer_reference=False)
    assert kind == "reference" and kind2 == "port"
    out["cfg2_r18_fwd_bwd_1graph"] = {"reference_s": round(t_ref, 3), "port_s": round(t_port, 3), "port_over_reference": round(t_port / t_ref, 3),
                                      "max_abs_diff": float(max(np.abs(o_ref - o_port).max(), np.abs(d_ref - d_port).max()))}
    print(out["cfg2_r18_fwd_bwd_1graph"], flush=True)
    # north_star's six-thread class: forward only (its backward races)
    t0 = time.perf_counter()
    o1 = ref.r18_thread_forward(P, A)
    t1 = time.perf_counter()
    o2 = orc.r18_thread_forward(P, A)
    t2 = time.perf_counter()
    out["cfg2_r18_thread_forward_1graph_6threads"] = {"reference_s": round(t1 - t0, 3), "port_s": round(t2 - t1, 3),
                                                      "port_over_reference": round((t2 - t1) / (t1 - t0), 3),
                                                      "max_abs_diff": float(np.abs(o1 - o2).max())}
    print(out["cfg2_r18_thread_forward_1graph_6threads"], flush=True)
    # cfg3 shape: SMP_omega step body on the first molecules of the bench batch (3 levels, C=64, F=5, D=5, cap 29)
    L, Cs, F, D, cap = 3, 64, 5, 5, 29
    mols, tg = [], []
    for i in range(4):
        a, f, t = synthetic_molecule(i)
        mols.append((a, f))
        tg.append(t)
    params = smp_params(Cs, F, D, L, 1)
    t_ref = pyoracle.time_reference_smp_omega(mols, tg, L, Cs, D, cap)
    t_port, _, _, _ = pyoracle.port_smp_batch(mols, tg, params, L, Cs, D, cap, 1)
    out["cfg3_smp_step_1thread"] = {"molecules_nV": [len(a) for a, _ in mols], "reference_s": round(t_ref, 3), "port_s": round(t_port, 3),
                                    "port_over_reference": round(t_port / t_ref, 3)}
    print(out["cfg3_smp_step_1thread"], flush=True)
    # all host cores, batch-parallel: the real Threaded_BatchLearn beside the port's wave driver
    nT = os.cpu_count() or 8
    mols, tg = [], []
    for i in range(nT):
        a, f, t = synthetic_molecule(i)
        mols.append((a, f))
        tg.append(t)
    t_ref = pyoracle.time_reference_smp_omega_threaded(mols, tg, L, Cs, D, cap, nT)
    t_port, _, _, _ = pyoracle.port_smp_batch(mols, tg, params, L, Cs, D, cap, nT)
    out["cfg3_smp_step_all_cores"] = {"threads": nT, "molecules": len(mols), "reference_s": round(t_ref, 3), "port_s": round(t_port, 3),
                                      "port_over_reference": round(t_port / t_ref, 3)}
    print(out["cfg3_smp_step_all_cores"], flush=True)
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02_port_vs_reference.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
