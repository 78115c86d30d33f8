"""GPU parity suite for the dense feature mixers (MatMul, MatTensorMul, TensorMatMul, StackTensor3D) vs goldens/oracle."""
import numpy as np
import pytest

from inputs import f32exact
from util import REL_TOL_F32, golden_cases, rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def test_matmul_golden(gf, golden):
    cases = golden_cases(golden, "mm_")
    assert len(cases) == 3
    for tag, c in cases.items():
        A, B, dC = dev(c["A"]), dev(c["B"]), dev(c["dC"])
        assert rel_err(host(gf.matmul_forward(A, B)), c["C"]) <= REL_TOL_F32, tag
        dA, dB = dev(c["dA0"]), dev(c["dB0"])
        gf.matmul_backward(dC, A, B, dA=dA, dB=dB, accumulate=True)
        assert rel_err(host(dA), c["dA"]) <= REL_TOL_F32, tag
        assert rel_err(host(dB), c["dB"]) <= REL_TOL_F32, tag


@pytest.mark.parametrize("M,K,N", [(1, 1, 1), (3, 5, 2), (64, 32, 64), (65, 33, 63), (100, 257, 7), (1024, 1152, 64),
                                   (37, 1152, 64), (4096, 72, 8), (130, 20, 130)])
def test_matmul_vs_oracle(gf, oracle, M, K, N):
    """A=I-style transposition traps are avoided by asymmetric random operands (cdna guide: asymmetric B)."""
    rng = np.random.default_rng(M * 7 + K * 3 + N)
    A = f32exact(rng.uniform(-1, 1, (M, K)))
    B = f32exact(rng.uniform(-1, 1, (K, N)))
    dC = f32exact(rng.uniform(-1, 1, (M, N)))
    assert rel_err(host(gf.matmul_forward(dev(A), dev(B))), A @ B) <= REL_TOL_F32
    dA = torch.empty((M, K), device="cuda")
    dB = torch.empty((K, N), device="cuda")
    gf.matmul_backward(dev(dC), dev(A), dev(B), dA=dA, dB=dB, accumulate=False)
    assert rel_err(host(dA), dC @ B.T) <= REL_TOL_F32
    assert rel_err(host(dB), A.T @ dC) <= REL_TOL_F32
    if M * K * N <= 200000:  # the oracle's own triple loop on the small shapes
        assert rel_err(host(gf.matmul_forward(dev(A), dev(B))), oracle.matmul_forward(A, B)) <= REL_TOL_F32
        oa, ob = oracle.matmul_backward(dC, A, B)
        assert rel_err(host(dA), oa) <= REL_TOL_F32 and rel_err(host(dB), ob) <= REL_TOL_F32


def test_kprojection_splitk_is_deterministic(gf):
    """dB = A^T dC with a long reduction (the K-projection weight gradient) goes through split-K + ordered reduce."""
    M, K, N = 20000, 1152, 64
    gen = torch.Generator(device="cuda").manual_seed(3)
    A = torch.rand((M, K), device="cuda", generator=gen) - 0.5
    dC = torch.rand((M, N), device="cuda", generator=gen) - 0.5
    B = torch.rand((K, N), device="cuda", generator=gen)
    dB1 = torch.empty((K, N), device="cuda")
    dB2 = torch.empty((K, N), device="cuda")
    gf.matmul_backward(dC, A, B, dB=dB1)
    gf.matmul_backward(dC, A, B, dB=dB2)
    assert torch.equal(dB1, dB2)
    ref = (A.double().T @ dC.double())
    assert float((dB1.double() - ref).abs().max() / ref.abs().max()) <= 1e-5


def test_promotion_pair_golden(gf, golden):
    """MatTensorMul then TensorMatMul = X F X^T, with a 0/1 selection X (the SMP use) and with a dense X."""
    for tag, c in golden_cases(golden, "promote_").items():
        X, F, G2 = dev(c["X"]), dev(c["F"]), dev(c["G2"])
        XT = X.t().contiguous()
        T1 = gf.mattensormul_forward(X, F)
        assert rel_err(host(T1), c["T1"]) <= REL_TOL_F32, tag
        T2 = gf.tensormatmul_forward(T1, XT)
        assert rel_err(host(T2), c["T2"]) <= REL_TOL_F32, tag
        dT1 = torch.zeros_like(T1)
        dY = torch.zeros_like(XT)
        gf.tensormatmul_backward(G2, T1, XT, dF=dT1, dY=dY, accumulate=True)
        assert rel_err(host(dT1), c["dT1"]) <= REL_TOL_F32, tag
        assert rel_err(host(dY), c["dY"]) <= REL_TOL_F32, tag
        dX = torch.zeros_like(X)
        dF = torch.zeros_like(F)
        gf.mattensormul_backward(dT1, X, F, dX=dX, dF=dF, accumulate=True)
        assert rel_err(host(dF), c["dF"]) <= REL_TOL_F32, tag
        assert rel_err(host(dX), c["dX"]) <= REL_TOL_F32, tag


@pytest.mark.parametrize("R,Kd,J,D", [(1, 1, 1, 1), (5, 3, 4, 2), (29, 23, 23, 64), (32, 32, 32, 16)])
def test_tensor_mixers_vs_oracle(gf, oracle, R, Kd, J, D):
    rng = np.random.default_rng(R + Kd + J + D)
    X = f32exact(rng.uniform(-1, 1, (R, Kd)))
    F = f32exact(rng.uniform(-1, 1, (Kd, J, D)))
    G = f32exact(rng.uniform(-1, 1, (R, J, D)))
    assert rel_err(host(gf.mattensormul_forward(dev(X), dev(F))), oracle.mattensormul_forward(X, F)) <= REL_TOL_F32
    dX, dF = torch.empty((R, Kd), device="cuda"), torch.empty((Kd, J, D), device="cuda")
    gf.mattensormul_backward(dev(G), dev(X), dev(F), dX=dX, dF=dF)
    oX, oF = oracle.mattensormul_backward(G, X, F)
    assert rel_err(host(dX), oX) <= REL_TOL_F32 and rel_err(host(dF), oF) <= REL_TOL_F32
    F2 = f32exact(rng.uniform(-1, 1, (R, Kd, D)))
    Y = f32exact(rng.uniform(-1, 1, (Kd, J)))
    assert rel_err(host(gf.tensormatmul_forward(dev(F2), dev(Y))), oracle.tensormatmul_forward(F2, Y)) <= REL_TOL_F32
    dF2, dY = torch.empty((R, Kd, D), device="cuda"), torch.empty((Kd, J), device="cuda")
    gf.tensormatmul_backward(dev(G), dev(F2), dev(Y), dF=dF2, dY=dY)
    oF2, oY = oracle.tensormatmul_backward(G, F2, Y)
    assert rel_err(host(dF2), oF2) <= REL_TOL_F32 and rel_err(host(dY), oY) <= REL_TOL_F32


def test_custommatmultensor_golden(gf, golden):
    cases = golden_cases(golden, "cmix_")
    assert len(cases) == 3
    for tag, c in cases.items():
        W, T, G = dev(c["W"]), dev(c["T"]), dev(c["G"])
        assert rel_err(host(gf.custommatmultensor_forward(W, T)), c["Out"]) <= REL_TOL_F32, tag
        dW, dT = dev(c["dW0"]), dev(c["dT0"])
        gf.custommatmultensor_backward(G, W, T, dW=dW, dT=dT, accumulate=True)
        assert rel_err(host(dW), c["dW"]) <= REL_TOL_F32, tag
        assert rel_err(host(dT), c["dT"]) <= REL_TOL_F32, tag


@pytest.mark.parametrize("I,J,V,Kout", [(2, 3, 5, 7), (29, 29, 1152, 64), (64, 64, 180, 10), (17, 5, 64, 33)])
def test_custommatmultensor_vs_oracle(gf, oracle, I, J, V, Kout):
    rng = np.random.default_rng(I + 3 * J + 5 * V + 7 * Kout)
    W = f32exact(rng.uniform(-1, 1, (Kout, V)))
    T = f32exact(rng.uniform(-1, 1, (I, J, V)))
    G = f32exact(rng.uniform(-1, 1, (I, J, Kout)))
    out = host(gf.custommatmultensor_forward(dev(W), dev(T)))
    assert rel_err(out, np.einsum("kv,ijv->ijk", W, T)) <= REL_TOL_F32
    dW = torch.empty((Kout, V), device="cuda")
    dT = torch.empty((I, J, V), device="cuda")
    gf.custommatmultensor_backward(dev(G), dev(W), dev(T), dW=dW, dT=dT, accumulate=False)
    assert rel_err(host(dW), np.einsum("ijk,ijv->kv", G, T)) <= REL_TOL_F32
    assert rel_err(host(dT), np.einsum("ijk,kv->ijv", G, W)) <= REL_TOL_F32
    if I * J * V * Kout <= 300000:
        assert rel_err(out, oracle.custommatmultensor_forward(W, T)) <= REL_TOL_F32
        oW, oT = oracle.custommatmultensor_backward(G, W, T)
        assert rel_err(host(dW), oW) <= REL_TOL_F32 and rel_err(host(dT), oT) <= REL_TOL_F32


def test_stack_round_trip(gf):
    """StackTensor3D: forward is a bit-exact copy, backward scatter-adds (`+=`) into the sources' gradients."""
    N, C = 7, 12
    ts = [torch.rand((N, N, C), device="cuda") for _ in range(N)]
    st = gf.stack_forward(ts)
    assert torch.equal(st, torch.stack(ts))
    grads = [torch.rand((N, N, C), device="cuda") for _ in range(N)]
    g0 = [g.clone() for g in grads]
    G = torch.rand((N, N, N, C), device="cuda")
    gf.stack_backward(G, grads)
    for r in range(N):
        assert torch.equal(grads[r], g0[r] + G[r])


def test_stack_golden_vectors(gf, golden):
    """The reference's own StackTensor3D outputs (tests/golden/stack.npz, StackTensor3D.h:54-90): bit-exact -- the forward is a copy,
    the backward one fp32 add per element of fp32-representable operands whose sum the fixture holds in fp64."""
    from util import golden_cases
    cases = golden_cases(golden, "stack_")
    assert len(cases) == 3
    for tag, c in cases.items():
        ts = [dev(c["T"][r]) for r in range(c["T"].shape[0])]
        st = gf.stack_forward(ts)
        assert np.array_equal(st.cpu().numpy().astype(np.float64), c["Out"]), tag
        grads = [dev(c["dT0"][r]) for r in range(c["T"].shape[0])]
        gf.stack_backward(dev(c["G"]), grads)
        got = np.stack([g.cpu().numpy() for g in grads])
        assert np.array_equal(got, c["dT"].astype(np.float32)), tag


def test_host_pointer_matmul_f64(gf, oracle):
    import ctypes as C
    from graphflow_amd import _lib
    lib = _lib.load()
    ctx = gf.Context(0)
    rng = np.random.default_rng(1)
    M, K, N = 36, 144, 8
    A, B, dC = (f32exact(rng.uniform(-1, 1, s)) for s in ((M, K), (K, N), (M, N)))
    out = np.zeros((M, N))
    dp = C.POINTER(C.c_double)
    p = lambda a: a.ctypes.data_as(dp)
    ctx.check(lib.gf_matmul_forward_host_f64(ctx.handle, p(A), p(B), p(out), M, K, N))
    assert rel_err(out, A @ B) <= REL_TOL_F32
    dA, dB = f32exact(rng.uniform(-1, 1, (M, K))), f32exact(rng.uniform(-1, 1, (K, N)))
    a0, b0 = dA.copy(), dB.copy()
    ctx.check(lib.gf_matmul_backward_host_f64(ctx.handle, p(dC), p(A), p(B), p(dA), p(dB), M, K, N))
    assert rel_err(dA, a0 + dC @ B.T) <= REL_TOL_F32 and rel_err(dB, b0 + A.T @ dC) <= REL_TOL_F32


@pytest.mark.parametrize("n", [4, 1024, 4096 + 8, (1 << 22) + 12])
def test_hbm_copy_probe_copies(gf, n):
    """gf_hbm_copy_probe_f32 (the bench's practical-ceiling probe): both kernels copy exactly -- whole 16 KiB tiles and the ragged tail --
    and report a positive rate; bad arguments are refused."""
    ctx = gf.Context(0)
    src = torch.arange(n, dtype=torch.float32, device="cuda") * 0.5 + 1.0
    for mode in (0, 1):
        dst = torch.zeros(n, device="cuda")
        rate = ctx.hbm_copy_probe(dst, src, mode, iters=2)
        assert rate > 0 and torch.equal(dst, src), (n, mode)
    with pytest.raises(gf.GraphFlowHipError):
        ctx.hbm_copy_probe(torch.zeros(8, device="cuda"), torch.zeros(8, device="cuda"), 2, 1)
    ctx.close()
