set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_smp_gpu.py -q -x -m gpu -s -k "fields_above_32" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -25
for e in 1 0; do echo "== L4 48 GF_SMP_BIG_FIELDS=$e"; GF_SMP_BIG_FIELDS=$e python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch
from inputs import synthetic_molecule, smp_params
from graphflow_amd.smp import SMPOmega
L, C, F, D, nV, B = 4, 64, 5, 2, 48, 128
mols = [synthetic_molecule(i, nV=nV - i % 5)[:2] for i in range(B)]
net = SMPOmega(L, C, F, D, nV, True); net.prepare(mols)
p = torch.as_tensor(smp_params(C, F, D, L, 1).astype(np.float32)).cuda(); g = torch.empty_like(p)
t = torch.full((B,), 1.0, device="cuda")
for _ in range(2): net.forward(p, t); net.backward(p, g)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): net.forward(p, t); net.backward(p, g)
torch.cuda.synchronize(); print("L=4, 48 atoms, batch 128: %.2f ms per step" % ((time.perf_counter() - t0) / 5 * 1e3))
PY
done
