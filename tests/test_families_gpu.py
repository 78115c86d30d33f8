"""GPU parity suite for RisiContraction_4 / _10 / _50 (through the C ABI) vs golden vectors and the fp64 oracle."""
import json
import os

import numpy as np
import pytest

from inputs import adjacency, f32exact
from util import REL_TOL_F32, golden_cases, rel_err, rel_err_slices

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("K", [4, 10, 50])
def test_golden_vectors(gf, golden, K):
    cases = golden_cases(golden, "r%d_" % K)
    assert cases
    for tag, c in cases.items():
        A = dev(c["A"][None]) if "A" in c else None
        out = gf.contract_forward(dev(c["P"][None]), A, K)
        assert rel_err_slices(host(out)[0], c["Out"]) <= REL_TOL_F32, tag
        dP = dev(c["dP0"][None])
        gf.contract_backward(dev(c["G"][None]), A, K, dP=dP, accumulate=True)
        assert rel_err(host(dP)[0], c["dP"]) <= REL_TOL_F32, tag


@pytest.mark.parametrize("K", [4, 10, 50])
@pytest.mark.parametrize("N,C", [(1, 1), (2, 3), (6, 5), (9, 4), (12, 2)])
def test_forward_backward_vs_oracle(gf, oracle, K, N, C):
    rng = np.random.default_rng(1000 * K + 10 * N + C)
    B = 3
    P = f32exact(rng.uniform(-1, 1, (B, N, N, N, C)))
    A = np.stack([adjacency(k, N, rng) for k in ("sym01", "weighted", "signed")])
    G = f32exact(rng.uniform(-1, 1, (B, N, N, K, C)))
    d0 = f32exact(rng.uniform(-1, 1, (B, N, N, N, C)))
    Ad = dev(A) if K != 4 else None
    out = host(gf.contract_forward(dev(P), Ad, K))
    dw = host(gf.contract_backward(dev(G), Ad, K))
    da = dev(d0)
    gf.contract_backward(dev(G), Ad, K, dP=da, accumulate=True)
    da = host(da)
    for g in range(B):
        assert rel_err_slices(out[g], oracle.contract_forward(K, P[g], A[g])) <= REL_TOL_F32
        ref = oracle.contract_backward(K, G[g], A[g])
        assert rel_err(dw[g], ref) <= REL_TOL_F32
        assert rel_err(da[g], ref + d0[g]) <= REL_TOL_F32


@pytest.mark.parametrize("N,C", [(1, 4), (3, 8), (5, 16), (7, 12), (17, 32), (24, 32), (32, 64), (33, 64), (20, 128), (70, 256)])
def test_r4_slab_kernels_vs_oracle(gf, oracle, N, C):
    """RisiContraction_4 on the slab-streaming kernels of round 4 (C % 4 == 0 and C / 4 a power of two; (7, 12) takes the thread kernels):
    one to sixteen position slots per lane, ragged last slots, fewer rows than waves, write-only and accumulating backward -- against
    the fp64 oracle's spec form (RisiContraction_4.h:79-120), slice by slice."""
    rng = np.random.default_rng(4000 + 10 * N + C)
    B = 2
    P = f32exact(rng.uniform(-1, 1, (B, N, N, N, C)))
    G = f32exact(rng.uniform(-1, 1, (B, N, N, 4, C)))
    d0 = f32exact(rng.uniform(-1, 1, (B, N, N, N, C)))
    out = host(gf.contract_forward(dev(P), None, 4))
    dw = host(gf.contract_backward(dev(G), None, 4))
    da = dev(d0)
    gf.contract_backward(dev(G), None, 4, dP=da, accumulate=True)
    da = host(da)
    A = np.zeros((N, N))
    for g in range(B):
        assert rel_err_slices(out[g], oracle.contract_forward(4, P[g], A)) <= REL_TOL_F32
        ref = oracle.contract_backward(4, G[g], A)
        assert rel_err(dw[g], ref) <= REL_TOL_F32
        assert rel_err(da[g], ref + d0[g]) <= REL_TOL_F32


@pytest.mark.parametrize("N,C", [(1, 4), (2, 8), (3, 8), (5, 16), (8, 32), (17, 32), (24, 32), (32, 32), (16, 64), (15, 64), (7, 128), (9, 12), (24, 64)])
def test_r10_graph_stream_kernels_vs_oracle_and_table_kernels(gf, oracle, monkeypatch, N, C):
    """RisiContraction_10 on the one-stream-per-graph kernels of round 4 (r10_fwd_graph / r10_bwd_graph: P read once, the three pair
    marginals never stored as tables): one to four position slots per lane, odd N (a last wave with one row), ragged last slots,
    768- and 1024-thread builds, write-only and accumulating backward, symmetric / weighted / signed adjacencies -- against the fp64
    oracle's spec form (RisiContraction_10.h:94-142) slice by slice, and against the table kernels (GF_FAM10_GRAPH=0).  (9, 12) and
    (24, 64) are outside the kernels' shapes (C / 4 not a power of two; six slots): both legs run the table kernels."""
    rng = np.random.default_rng(10000 + 10 * N + C)
    B = 3
    P = f32exact(rng.uniform(-1, 1, (B, N, N, N, C)))
    A = np.stack([adjacency(k, N, rng) for k in ("sym01", "weighted", "signed")])
    G = f32exact(rng.uniform(-1, 1, (B, N, N, 10, C)))
    d0 = f32exact(rng.uniform(-1, 1, (B, N, N, N, C)))
    got = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("GF_FAM10_GRAPH", mode)
        out = host(gf.contract_forward(dev(P), dev(A), 10))
        dw = host(gf.contract_backward(dev(G), dev(A), 10))
        da = dev(d0)
        gf.contract_backward(dev(G), dev(A), 10, dP=da, accumulate=True)
        got[mode] = (out, dw, host(da))
    out, dw, da = got["2"]
    for g in range(B):
        assert rel_err_slices(out[g], oracle.contract_forward(10, P[g], A[g])) <= REL_TOL_F32
        ref = oracle.contract_backward(10, G[g], A[g])
        assert rel_err(dw[g], ref) <= REL_TOL_F32
        assert rel_err(da[g], ref + d0[g]) <= REL_TOL_F32
    for x, y in zip(got["2"], got["0"]):
        assert rel_err(x, y) <= REL_TOL_F32


def test_structural_50_collapse_on_gpu(gf):
    """The reference's own known-answer (tests/test_RisiContraction_50.cpp): 50 slices -> the 18 recorded groups,
    bit-identical, for integer tensors symmetric in (b,c) and a symmetric zero-diagonal 0/1 adjacency."""
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "structural_50.json")) as fh:
        expected = json.load(fh)["groups"]
    rng = np.random.default_rng(50)
    N, C = 10, 5
    P = rng.integers(0, 100, (N, N, N, C)).astype(np.float64)
    P = np.triu(P.transpose(0, 3, 1, 2), 0)
    P = (P + np.triu(P, 1).transpose(0, 1, 3, 2)).transpose(0, 2, 3, 1).copy()
    U = np.triu((rng.uniform(0, 1, (N, N)) < 0.5).astype(np.float64), 1)
    A = U + U.T
    out = host(gf.contract_forward(dev(P[None]), dev(A[None]), 50))[0]
    free, groups = [True] * 50, []
    for i in range(50):
        if free[i]:
            grp = [j + 1 for j in range(i, 50) if np.array_equal(out[:, :, i, :], out[:, :, j, :])]
            for j in grp:
                free[j - 1] = False
            groups.append(grp)
    assert groups == expected


def test_r18_is_the_gated_subset_of_r50(gf):
    """RisiContraction_18's slices are cases {1,3,5,6,10,11,13,17,18,23,26,27,28,38,40,43,46,50} of _50 on A+."""
    rng = np.random.default_rng(18)
    N, C = 7, 8
    P = dev(rng.uniform(-1, 1, (2, N, N, N, C)))
    A = rng.uniform(-1, 1, (2, N, N))
    o18 = host(gf.contract_forward(P, dev(A), 18))
    o50 = host(gf.contract_forward(P, dev(np.where(A > 0, A, 0.0)), 50))
    sel = [c - 1 for c in (1, 3, 5, 6, 10, 11, 13, 17, 18, 23, 26, 27, 28, 38, 40, 43, 46, 50)]
    assert rel_err_slices(o18, o50[:, :, :, sel, :]) <= REL_TOL_F32


@pytest.mark.parametrize("lds", ["0", "threads"])
@pytest.mark.parametrize("K", [50, 10])
def test_cfg5_shape_one_graph_vs_oracle_spec_form(gf, oracle, monkeypatch, K, lds):
    """BASELINE configs[4]'s shape (N = 24, C = 32): the lane mapping of fam_tables<50,4> / fam_forward<50,4> / fam_bwd_tables<50,1>
    is only reached at C % 4 == 0 and this size.  Graph 0 of a 3-graph batch (weighted adjacency, so the no-gate rule of _50 / _10
    matters) against the oracle's table-driven spec form (RisiContraction_50.h:94-430) on two channels -- channels are independent,
    so the oracle runs the O(N^5) form on a 2-channel copy."""
    # "0": the default kernels (K = 50: fam50_forward_mfma / fam50_bwd_tables_mfma, the matrix-pipe forms); "threads": the
    # thread-per-element kernels the matrix-pipe forms replaced (they serve every other shape and _10)
    if lds == "threads":
        monkeypatch.setenv("GF_FAM_FWD_MFMA", "0")
        monkeypatch.setenv("GF_FAM_BWD_MFMA", "0")
    rng = np.random.default_rng(5050 + K)
    B, N, C = 3, 24, 32
    P = f32exact(rng.uniform(-1, 1, (B, N, N, N, C)))
    A = np.stack([adjacency(k, N, rng) for k in ("signed", "weighted", "sym01")])
    G = f32exact(rng.uniform(-1, 1, (B, N, N, K, C)))
    out = host(gf.contract_forward(dev(P), dev(A), K))
    dP = host(gf.contract_backward(dev(G), dev(A), K))
    sub = [3, 30]
    for g in range(2):
        ref_out = oracle.contract_forward(K, np.ascontiguousarray(P[g][..., sub]), A[g])
        assert rel_err_slices(out[g][..., sub], ref_out) <= REL_TOL_F32, g
        ref_dp = oracle.contract_backward(K, np.ascontiguousarray(G[g][..., sub]), A[g])
        assert rel_err(dP[g][..., sub], ref_dp) <= REL_TOL_F32, g


@pytest.mark.parametrize("N,C", [(1, 32), (2, 32), (7, 32), (15, 64), (16, 32), (17, 96), (23, 32), (25, 64), (31, 32), (32, 64)])
def test_r50_matrix_pipe_kernels_vs_thread_kernels_and_oracle(gf, oracle, monkeypatch, N, C):
    """fam50_forward_mfma / fam50_bwd_tables_mfma (C % 32 == 0, N <= 32; wave per (graph, row, 32-channel window), z steps of two
    with odd and even N, padded accumulator rows, all three z-step instantiations) against the thread-per-element kernels on
    the whole batch and against the oracle's spec form (RisiContraction_50.h:94-430) on two channels of every graph."""
    K, B = 50, 3
    rng = np.random.default_rng(50000 + 100 * N + C)
    P = f32exact(rng.uniform(-1, 1, (B, N, N, N, C)))
    A = np.stack([adjacency(k, N, rng) for k in ("signed", "weighted", "sym01")])
    G = f32exact(rng.uniform(-1, 1, (B, N, N, K, C)))
    d0 = f32exact(rng.uniform(-1, 1, (B, N, N, N, C)))
    out = host(gf.contract_forward(dev(P), dev(A), K))
    dP = host(gf.contract_backward(dev(G), dev(A), K))
    da = dev(d0)
    gf.contract_backward(dev(G), dev(A), K, dP=da, accumulate=True)
    monkeypatch.setenv("GF_FAM_FWD_MFMA", "0")
    monkeypatch.setenv("GF_FAM_BWD_MFMA", "0")
    out_t = host(gf.contract_forward(dev(P), dev(A), K))
    dP_t = host(gf.contract_backward(dev(G), dev(A), K))
    for g in range(B):
        assert rel_err_slices(out[g], out_t[g]) <= REL_TOL_F32, g
        assert rel_err(dP[g], dP_t[g]) <= REL_TOL_F32, g
        assert rel_err(host(da)[g], dP_t[g] + d0[g]) <= REL_TOL_F32, g
    sub = [0, C - 1] if N > 12 else list(range(0, C, max(1, C // 8)))
    for g in range(B if N <= 17 else 1):
        ref_out = oracle.contract_forward(K, np.ascontiguousarray(P[g][..., sub]), A[g])
        assert rel_err_slices(out[g][..., sub], ref_out) <= REL_TOL_F32, g
        ref_dp = oracle.contract_backward(K, np.ascontiguousarray(G[g][..., sub]), A[g])
        assert rel_err(dP[g][..., sub], ref_dp) <= REL_TOL_F32, g


def test_cfg5_full_size_properties(gf):
    """BASELINE cfg5 (RisiContraction_50, N=24, C=32), batch 64: adjoint identity and linearity."""
    B, N, C, K = 64, 24, 32, 50
    gen = torch.Generator(device="cuda").manual_seed(5)
    P = torch.rand((B, N, N, N, C), device="cuda", generator=gen) * 2 - 1
    G = torch.rand((B, N, N, K, C), device="cuda", generator=gen)
    U = (torch.rand((B, N, N), device="cuda", generator=gen) < 0.5).float().triu(1)
    A = U + U.transpose(1, 2) + torch.eye(N, device="cuda")
    out = gf.contract_forward(P, A, K)
    dP = gf.contract_backward(G, A, K)
    lhs = (out.double() * G.double()).sum(dim=(1, 2, 3, 4))
    rhs = (P.double() * dP.double()).sum(dim=(1, 2, 3, 4))
    mag = (out.double().abs() * G.double()).sum(dim=(1, 2, 3, 4)).clamp_min(1.0)
    assert float(((lhs - rhs).abs() / mag).max()) <= 1e-5
    assert torch.equal(out, gf.contract_forward(P, A, K))
