"""RisiContraction_10 forward + backward at the cfg5 shape (N = 24, C = 32, batch 256): ms per step and the algorithmic HBM rate.
usage: python tools/fam10_time.py [K]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import graphflow_amd as gf  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N, C, B = 24, 32, 256
torch.manual_seed(0)
P = torch.rand(B, N, N, N, C, device="cuda") * 2 - 1
A = torch.rand(B, N, N, device="cuda") * 2 - 1
G = torch.rand(B, N, N, K, C, device="cuda")
for _ in range(3):
    out = gf.contract_forward(P, A, K)
    dP = gf.contract_backward(G, A, K)
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 20
for _ in range(steps):
    out = gf.contract_forward(P, A, K)
    dP = gf.contract_backward(G, A, K)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
alg = 2 * 4 * (N ** 3 * C + N * N + K * N * N * C) * B
print("RisiContraction_%d fwd+bwd N=%d C=%d batch=%d: %.3f ms per step, %.2f GB algorithmic -> %.2f TB/s" % (K, N, C, B, ms, alg / 1e9, alg / ms / 1e9))
