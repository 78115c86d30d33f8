"""Skinny (N=64) GEMM microbench at the fused SMP level's shapes, vs hipBLASLt through torch.matmul."""
import sys, time; sys.path.insert(0, '/root/repo')
import torch, graphflow_amd as gf
ctx = gf.default_context()


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def bench(M, K, N):
    A = torch.rand((M, K), device='cuda'); B = torch.rand((K, N), device='cuda'); C = torch.empty((M, N), device='cuda')
    dC = torch.rand((M, N), device='cuda'); dA = torch.empty((M, K), device='cuda'); dB = torch.empty((K, N), device='cuda')
    flops = 2.0 * M * K * N; byts = 4.0 * (M * K + K * N + M * N)
    for name, fn, ref in (("nn", lambda: gf.matmul_forward(A, B, out=C), lambda: torch.matmul(A, B, out=C)),
                          ("nt", lambda: gf.matmul_backward(dC, A, B, dA=dA), lambda: torch.matmul(dC, B.t(), out=dA)),
                          ("tn", lambda: gf.matmul_backward(dC, A, B, dB=dB), lambda: torch.matmul(A.t(), dC, out=dB))):
        dt = timed(fn); dr = timed(ref)
        print(f"{name} M={M} K={K} N={N}: ours {dt*1e3:.3f} ms {flops/dt/1e12:.1f} TF/s {byts/dt/1e9:.0f} GB/s | hipBLASLt {dr*1e3:.3f} ms {flops/dr/1e12:.1f} TF/s {byts/dr/1e9:.0f} GB/s")


for K in (64, 128, 192, 384, 1152):
    bench(1937408, K, 64)
bench(1937408, 384, 320)
bench(8192, 8192, 8192)
