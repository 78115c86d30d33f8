import sys, time; sys.path.insert(0,'/root/repo')
import torch, graphflow_amd as gf
ctx = gf.default_context()
def bench(M,K,N, which='nn', reps=5):
    A = torch.rand((M,K), device='cuda'); B = torch.rand((K,N), device='cuda'); C = torch.empty((M,N), device='cuda')
    dC = torch.rand((M,N), device='cuda'); dA = torch.empty((M,K), device='cuda'); dB = torch.empty((K,N), device='cuda')
    def run():
        if which=='nn': gf.matmul_forward(A,B,out=C)
        elif which=='nt': gf.matmul_backward(dC,A,B,dA=dA)
        else: gf.matmul_backward(dC,A,B,dB=dB)
    run(); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(reps): run()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/reps
    flops=2.0*M*K*N; byts=4.0*(M*K+K*N+M*N)
    print(f"{which} M={M} K={K} N={N}: {dt*1e3:.3f} ms  {flops/dt/1e12:.1f} TF/s  {byts/dt/1e9:.0f} GB/s algorithmic")
    if which=='nn':
        t=time.perf_counter()
        for _ in range(reps): torch.matmul(A,B,out=C)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t)/reps
        print(f"   torch(hipBLASLt) {dt*1e3:.3f} ms {flops/dt/1e12:.1f} TF/s")
bench(8192,8192,8192,'nn',3)
bench(4096,4096,4096,'nn')
bench(1937408,1152,64,'nn')
bench(1937408,1152,64,'nt')
bench(1937408,1152,64,'tn')
bench(262144,1152,64,'nn')
