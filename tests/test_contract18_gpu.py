"""GPU parity suite for RisiContraction_18: HIP kernels (through the C ABI) vs the fp64 oracle / golden vectors."""
import numpy as np
import pytest

from inputs import adjacency, cfg_graph, f32exact
from util import REL_TOL_F32, golden_cases, rel_err, rel_err_slices

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def force_generic(gf, on):
    """GF_OPT_R18_GENERIC_KERNELS on the default context: the layout-agnostic second implementation of RisiContraction_18."""
    from graphflow_amd import _lib
    from graphflow_amd.ops import default_context
    default_context(0).set_option(_lib.GF_OPT_R18_GENERIC_KERNELS, 1 if on else 0)


@pytest.fixture(autouse=True)
def _reset_generic(gf):
    yield
    force_generic(gf, False)


@pytest.mark.parametrize("generic", [False, True])
def test_golden_vectors(gf, golden, generic):
    force_generic(gf, generic)
    cases = golden_cases(golden, "r18_")
    assert len(cases) >= 10
    for tag, c in cases.items():
        out = gf.contract_forward(dev(c["P"][None]), dev(c["A"][None]), 18)
        assert rel_err_slices(host(out)[0], c["Out"]) <= REL_TOL_F32, tag
        dP = dev(c["dP0"][None])
        gf.contract_backward(dev(c["G"][None]), dev(c["A"][None]), 18, dP=dP, accumulate=True)
        assert rel_err(host(dP)[0], c["dP"]) <= REL_TOL_F32, tag


SHAPES = [(1, 4), (2, 8), (3, 64), (4, 4), (5, 12), (7, 64), (8, 16), (9, 32), (13, 64), (16, 8), (17, 64), (24, 32),
          (29, 64), (32, 64), (33, 8), (6, 128), (5, 48), (6, 3), (4, 10), (10, 5), (12, 1), (40, 4),
          # round 4: beyond eight wave loads per row the channel window is halved (fast_shape): 32-channel windows to N = 64, 16 to N = 128
          (33, 64), (40, 64), (50, 64), (64, 64), (45, 32), (70, 32), (100, 16), (128, 8), (129, 4)]


@pytest.mark.parametrize("N,C", SHAPES)
def test_forward_backward_vs_oracle(gf, oracle, N, C):
    """Seeded inputs, every lane-mapping family (LPC 4/8/16, NI 1/2/4/8, ragged and full, generic fallback)."""
    rng = np.random.default_rng(100 * N + C)
    B = 3
    kinds = ["sym01", "weighted", "signed"]
    P = f32exact(rng.uniform(-1, 1, (B, N, N, N, C)))
    A = np.stack([adjacency(kinds[g % 3], N, rng) for g in range(B)])
    G = f32exact(rng.uniform(0, 1, (B, N, N, 18, C)))
    d0 = f32exact(rng.uniform(-1, 1, (B, N, N, N, C)))
    # the oracle's spec form is O(N^5 C); keep big shapes to a channel subset on the CPU side
    fs = slice(0, C) if N * N * N * C <= 40000 else slice(0, max(1, 40000 // (N * N * N)))
    out = host(gf.contract_forward(dev(P), dev(A), 18))
    dP_w = host(gf.contract_backward(dev(G), dev(A), 18))
    dP_acc = dev(d0)
    gf.contract_backward(dev(G), dev(A), 18, dP=dP_acc, accumulate=True)
    dP_acc = host(dP_acc)
    # ... and big N to fewer graphs (N = 50: 18 s per graph and channel); beyond that the generic kernels below are the checker
    for g in range(B if N <= 40 else 1 if N <= 50 else 0):
        ref_out = oracle.contract_forward(18, P[g][..., fs], A[g])
        assert rel_err_slices(out[g][..., fs], ref_out) <= REL_TOL_F32
        ref_dp = oracle.contract_backward(18, G[g][..., fs], A[g])
        assert rel_err(dP_w[g][..., fs], ref_dp) <= REL_TOL_F32
        assert rel_err(dP_acc[g][..., fs], ref_dp + d0[g][..., fs]) <= REL_TOL_F32
    if fs != slice(0, C):  # channels are independent: the unchecked ones must agree with the generic kernels
        force_generic(gf, True)
        out_g = host(gf.contract_forward(dev(P), dev(A), 18))
        dP_g = host(gf.contract_backward(dev(G), dev(A), 18))
        assert rel_err_slices(out, out_g) <= REL_TOL_F32
        assert rel_err(dP_w, dP_g) <= REL_TOL_F32


def test_fast_and_generic_paths_agree(gf):
    rng = np.random.default_rng(7)
    B, N, C = 4, 24, 32
    P, G = dev(rng.uniform(-1, 1, (B, N, N, N, C))), dev(rng.uniform(0, 1, (B, N, N, 18, C)))
    A = dev(np.stack([adjacency("weighted", N, rng) for _ in range(B)]))
    o1, d1 = host(gf.contract_forward(P, A, 18)), host(gf.contract_backward(G, A, 18))
    force_generic(gf, True)
    o2, d2 = host(gf.contract_forward(P, A, 18)), host(gf.contract_backward(G, A, 18))
    assert rel_err_slices(o1, o2) <= REL_TOL_F32 and rel_err(d1, d2) <= REL_TOL_F32


def test_cfg2_shape_one_graph_vs_oracle_loop_nests(gf, oracle):
    """cfg2 shape (N=32, C=64): graph 0 of the batch against the reference-structured loop nests on 2 channels."""
    import ctypes as C
    from oracle import pyoracle as po
    N, Cc = 32, 64
    P, A, G = cfg_graph(N, Cc, 1000)
    out = host(gf.contract_forward(dev(P[None]), dev(A[None]), 18))[0]
    dP = host(gf.contract_backward(dev(G[None]), dev(A[None]), 18))[0]
    sub = [0, 37]
    Ps, Gs = np.ascontiguousarray(P[..., sub]), np.ascontiguousarray(G[..., sub])
    ref_out = np.zeros((N, N, 18, 2))
    f = oracle.lib.gfo_r18_loops_forward
    f.argtypes = [po._dp, po._dp, po._dp, C.c_int, C.c_int]
    f.restype = None
    f(Ps, A, ref_out, N, 2)
    assert rel_err_slices(out[..., sub], ref_out) <= REL_TOL_F32
    ref_dp = np.zeros((N, N, N, 2))
    b = oracle.lib.gfo_r18_loops_backward
    b.argtypes = [po._dp, po._dp, po._dp, C.c_int, C.c_int]
    b.restype = None
    b(Gs, A, ref_dp, N, 2)
    assert rel_err(dP[..., sub], ref_dp) <= REL_TOL_F32


def test_cfg2_full_size_properties(gf):
    """BASELINE cfg2 (N=32, C=64, batch 256): size-independent properties of the pair (forward, backward)."""
    B, N, C = 256, 32, 64
    gen = torch.Generator(device="cuda").manual_seed(2)
    P = torch.rand((B, N, N, N, C), device="cuda", generator=gen) * 2 - 1
    P2 = torch.rand((B, N, N, N, C), device="cuda", generator=gen) * 2 - 1
    G = torch.rand((B, N, N, 18, C), device="cuda", generator=gen)
    U = (torch.rand((B, N, N), device="cuda", generator=gen) < 0.5).float().triu(1)
    A = U + U.transpose(1, 2) + torch.eye(N, device="cuda")
    out = gf.contract_forward(P, A, 18)
    # determinism (no atomics anywhere): bit-identical on a second run
    assert torch.equal(out, gf.contract_forward(P, A, 18))
    # linearity in P
    out2 = gf.contract_forward(P2, A, 18)
    lin = gf.contract_forward(P * 0.5 + P2 * 2.0, A, 18)
    scale = float(out.abs().max())
    assert float((lin - (0.5 * out + 2.0 * out2)).abs().max()) <= 2e-5 * scale
    # adjoint identity <F(P), G> == <P, F^T(G)> per graph, in fp64 accumulation
    dP = gf.contract_backward(G, A, 18)
    lhs = (out.double() * G.double()).sum(dim=(1, 2, 3, 4))
    rhs = (P.double() * dP.double()).sum(dim=(1, 2, 3, 4))
    # both sides are signed sums with cancellation: normalise by the magnitude of what was summed
    mag = (out.double().abs() * G.double()).sum(dim=(1, 2, 3, 4)).clamp_min(1.0)
    assert float(((lhs - rhs).abs() / mag).max()) <= 1e-5
    # accumulate == write-only + old value
    d0 = torch.rand_like(dP)
    acc = d0.clone()
    gf.contract_backward(G, A, 18, dP=acc, accumulate=True)
    assert float((acc - (d0 + dP)).abs().max()) <= 1e-5 * float(dP.abs().max())
    # graphs are independent: graph 5 alone gives the same bits as inside the batch
    assert torch.equal(gf.contract_forward(P[5:6].contiguous(), A[5:6].contiguous(), 18)[0], out[5])


def test_zero_adjacency_and_empty_batch(gf):
    N, C = 8, 16
    P = torch.rand((2, N, N, N, C), device="cuda")
    A = torch.zeros((2, N, N), device="cuda")
    A[1] = -1.0  # all gated away
    assert float(gf.contract_forward(P, A, 18).abs().max()) == 0.0
    out = gf.contract_forward(P[:0].contiguous(), A[:0].contiguous(), 18)
    assert out.shape == (0, N, N, 18, C)


def test_errors_are_reported_not_fatal(gf):
    from graphflow_amd import _lib
    ctx = gf.default_context()
    P = torch.rand((1, 4, 4, 4, 4), device="cuda")
    A = torch.rand((1, 4, 4), device="cuda")
    with pytest.raises(gf.GraphFlowHipError) as ei:
        gf.contract_forward(P, A, K=7)
    assert ei.value.status == _lib.GF_ERR_INVALID and "K=7" in str(ei.value)
    gf.contract_forward(P, A, 18)  # context still usable
    ctx.synchronize()


def test_host_pointer_mode_f64(gf, oracle):
    """Mode A (what the Entity-style op calls): N separate host tensors in, host value/gradient out, `+=` honoured."""
    import ctypes as C
    from graphflow_amd import _lib
    lib = _lib.load()
    ctx = gf.Context(0, own_stream=True)
    rng = np.random.default_rng(11)
    N, Cc = 6, 8
    tensors = [f32exact(rng.uniform(-1, 1, (N, N, Cc))) for _ in range(N)]
    A = adjacency("weighted", N, rng)
    dp = C.POINTER(C.c_double)
    arr = (dp * N)(*[t.ctypes.data_as(dp) for t in tensors])
    out = np.zeros((N, N, 18, Cc))
    ctx.check(lib.gf_contract_forward_host_f64(ctx.handle, 18, arr, A.ctypes.data_as(dp), out.ctypes.data_as(dp), N, Cc))
    P = np.stack(tensors)
    assert rel_err_slices(out, oracle.contract_forward(18, P, A)) <= REL_TOL_F32
    G = f32exact(rng.uniform(0, 1, (N, N, 18, Cc)))
    grads = [f32exact(rng.uniform(-1, 1, (N, N, Cc))) for _ in range(N)]
    g0 = np.stack(grads).copy()
    garr = (dp * N)(*[t.ctypes.data_as(dp) for t in grads])
    ctx.check(lib.gf_contract_backward_host_f64(ctx.handle, 18, G.ctypes.data_as(dp), A.ctypes.data_as(dp), garr, N, Cc))
    assert rel_err(np.stack(grads), oracle.contract_backward(18, G, A, g0)) <= REL_TOL_F32
    ctx.close()


def test_r18_dropout_golden(gf, golden):
    """Slice dropout (RisiContraction_18_dropout.h): train-mode masks as the reference drew them, test-mode scaling."""
    cases = golden_cases(golden, "drop_")
    assert len(cases) == 4
    for tag, c in cases.items():
        seed, nKept, train = (int(x) for x in c["cfg"])
        P, A, G = (torch.as_tensor(c[k]).cuda()[None] for k in ("P", "A", "G"))
        out = gf.contract18_dropout_forward(P, A, c["use"], train=bool(train), nKept=nKept)
        assert rel_err_slices(out[0].cpu().numpy().astype(np.float64), c["Out"]) <= REL_TOL_F32, tag
        if train:
            dropped = [k for k in range(18) if not c["use"][k]]
            assert not out[0][:, :, dropped, :].any(), tag
            dP = torch.as_tensor(c["dP0"]).cuda()[None].clone()
            gf.contract18_dropout_backward(G, A, c["use"], dP=dP, accumulate=True)
            assert rel_err(dP[0].cpu().numpy().astype(np.float64), c["dP"]) <= REL_TOL_F32, tag


def test_r18_dropout_batch_vs_oracle(gf, oracle):
    rng = np.random.default_rng(31)
    B, N, C = 3, 8, 8
    use = np.zeros(18, dtype=np.int32)
    use[[0, 3, 4, 9, 16, 17]] = 1
    P = f32exact(rng.uniform(-1, 1, (B, N, N, N, C)))
    A = f32exact((rng.uniform(0, 1, (B, N, N)) < 0.5) * rng.uniform(0.5, 2, (B, N, N)))
    G = f32exact(rng.uniform(-1, 1, (B, N, N, 18, C)))
    out = gf.contract18_dropout_forward(torch.as_tensor(P, dtype=torch.float32).cuda(), torch.as_tensor(A, dtype=torch.float32).cuda(), use)
    dP = gf.contract18_dropout_backward(torch.as_tensor(G, dtype=torch.float32).cuda(), torch.as_tensor(A, dtype=torch.float32).cuda(), use)
    for b in range(B):
        o, d, _ = oracle.r18_dropout(use, True, 6, P[b], A[b], G[b])
        assert rel_err_slices(out[b].cpu().numpy().astype(np.float64), o) <= REL_TOL_F32
        assert rel_err(dP[b].cpu().numpy().astype(np.float64), d) <= REL_TOL_F32
