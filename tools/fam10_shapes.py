import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import graphflow_amd as gf
for N, C, B in ((32, 32, 256), (16, 64, 512), (24, 32, 96), (12, 32, 1024), (8, 16, 4096)):
    P = torch.rand(B, N, N, N, C, device="cuda") * 2 - 1
    A = torch.rand(B, N, N, device="cuda")
    G = torch.rand(B, N, N, 10, C, device="cuda")
    for mode in ("1", "0"):
        os.environ["GF_FAM10_GRAPH"] = mode
        for _ in range(3):
            gf.contract_forward(P, A, 10); gf.contract_backward(G, A, 10)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            gf.contract_forward(P, A, 10); gf.contract_backward(G, A, 10)
        torch.cuda.synchronize()
        print("N %d C %d batch %d GF_FAM10_GRAPH=%s: %.3f ms" % (N, C, B, mode, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
