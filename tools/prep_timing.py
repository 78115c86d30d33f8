"""Host graph-preparation breakdown (GF_PREP_TIMING=1) for the cfg3 batch, steady state.  usage: python tools/prep_timing.py"""
import os
import sys
import time

os.environ["GF_PREP_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402,F401
from inputs import synthetic_molecule  # noqa: E402
from graphflow_amd.smp import SMPOmega  # noqa: E402

mols = [synthetic_molecule(i)[:2] for i in range(1024)]
net = SMPOmega(3, 64, 5, 5, 29, True)
pk = SMPOmega.pack(mols)
print("host cores:", os.cpu_count())
for i in range(5):
    t0 = time.perf_counter()
    net.prepare(pk)
    print("prepare %d: %.2f ms" % (i, 1e3 * (time.perf_counter() - t0)), flush=True)
