import sys; sys.path.insert(0, '/root/repo')
import torch, graphflow_amd as gf
M, K, N = (int(x) for x in sys.argv[1:4]); which = sys.argv[4] if len(sys.argv) > 4 else 'nn'
A = torch.rand((M, K), device='cuda'); B = torch.rand((K, N), device='cuda'); C = torch.empty((M, N), device='cuda')
dA = torch.empty((M, K), device='cuda'); dB = torch.empty((K, N), device='cuda')
for _ in range(3):
    if which == 'nn': gf.matmul_forward(A, B, out=C)
    elif which == 'nt': gf.matmul_backward(C, A, B, dA=dA)
    else: gf.matmul_backward(C, A, B, dB=dB)
torch.cuda.synchronize()
