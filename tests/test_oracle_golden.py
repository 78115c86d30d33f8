"""CPU suite: pins oracle/gf_oracle.c against the golden vectors captured from the real reference
(tests/golden/make_golden.py) and, when oracle/_ref/libgf_ref.so is present, against the reference itself."""
import json
import os

import numpy as np
import pytest

from inputs import adjacency, f32exact
from util import golden_cases

TIGHT = 1e-12  # fp64 vs fp64, different summation order only


def _close(a, b, tol=TIGHT):
    scale = max(np.abs(b).max(), 1.0)
    assert np.abs(a - b).max() / scale <= tol


@pytest.mark.parametrize("K", [4, 10, 18, 50])
def test_contractions_match_golden(oracle, golden, K):
    cases = golden_cases(golden, "r%d_" % K)
    assert cases, "no fixtures for K=%d" % K
    for tag, c in cases.items():
        A = c.get("A")
        out = oracle.contract_forward(K, c["P"].astype(np.float64), None if A is None else A.astype(np.float64))
        _close(out, c["Out"])
        dP = oracle.contract_backward(K, c["G"].astype(np.float64), None if A is None else A.astype(np.float64),
                                      c["dP0"].astype(np.float64))
        _close(dP, c["dP"])


def test_survey_8c_shapes_are_all_there(golden):
    """every K at (N, C) in {(3,2), (5,3), (8,4), (16,8)} (round 6 added (16, 8) for `_10` / `_50`)."""
    for K in (4, 10, 18, 50):
        tags = set(golden_cases(golden, "r%d_" % K))
        for N, C in ((3, 2), (5, 3), (8, 4), (16, 8)):
            assert any(t.startswith("r%d_N%d_C%d_" % (K, N, C)) for t in tags), (K, N, C)


def test_stack_matches_golden(oracle, golden):
    """StackTensor3D fixtures written by the real StackTensor3D::forward / backward (StackTensor3D.h:54-90): a copy and a `+=`."""
    cases = golden_cases(golden, "stack_")
    assert len(cases) == 3
    for tag, c in cases.items():
        assert np.array_equal(c["Out"], c["T"].astype(np.float64)), tag                       # forward is a pure copy
        assert np.array_equal(oracle.stack_forward(c["T"]), c["Out"]), tag
        assert np.array_equal(oracle.stack_backward(c["G"], c["dT0"]), c["dT"]), tag          # exact: one add per element


def test_r18_gate_is_pinned(golden):
    """The 'signed' adjacency fixtures distinguish A>0 gating (r18) from no gating (r10/r50, r18_thread)."""
    c = golden_cases(golden, "r18_N5_C3_signed")["r18_N5_C3_signed"]
    A = c["A"].astype(np.float64)
    assert (A < 0).any()
    # slice k=4 is A+[d,e] * sum(P): zero exactly where A <= 0
    tot = c["P"].astype(np.float64).sum(axis=(0, 1, 2))
    expect = np.where(A > 0, A, 0.0)[:, :, None] * tot[None, None, :]
    _close(c["Out"][:, :, 4, :], expect)


def test_r18_thread_forward_is_ungated(oracle, golden):
    c = golden_cases(golden, "r18thread")["r18thread_N5_C3_signed"]
    out = oracle.r18_thread_forward(c["P"].astype(np.float64), c["A"].astype(np.float64))
    _close(out, c["Out"])
    gated = oracle.contract_forward(18, c["P"].astype(np.float64), c["A"].astype(np.float64))
    assert np.abs(gated - c["Out"]).max() > 1e-3  # the two semantics really differ on signed A


def test_r18_loop_nest_form_equals_spec_form(oracle):
    import ctypes as C
    from oracle import pyoracle as po
    rng = np.random.default_rng(5)
    N, Cc = 6, 3
    P = f32exact(rng.uniform(-1, 1, (N, N, N, Cc)))
    A = adjacency("signed", N, rng)
    G = f32exact(rng.uniform(0, 1, (N, N, 18, Cc)))
    out = np.zeros((N, N, 18, Cc))
    f = oracle.lib.gfo_r18_loops_forward
    f.argtypes = [po._dp, po._dp, po._dp, C.c_int, C.c_int]
    f.restype = None
    f(P, A, out, N, Cc)
    _close(out, oracle.contract_forward(18, P, A))
    dP = np.zeros((N, N, N, Cc))
    b = oracle.lib.gfo_r18_loops_backward
    b.argtypes = [po._dp, po._dp, po._dp, C.c_int, C.c_int]
    b.restype = None
    b(G, A, dP, N, Cc)
    _close(dP, oracle.contract_backward(18, G, A))


def test_r50_loop_nest_port_equals_spec_form_and_goldens(oracle, golden):
    """The CPU baseline of cfg5 (gfo_r50_loops_*: one five-deep nest per channel with the 50 predicated updates inside, the
    structure of RisiContraction_50.h:73-802) against the spec form on a signed adjacency and against the real reference's goldens."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(6)
    N, Cc = 5, 3
    P = f32exact(rng.uniform(-1, 1, (N, N, N, Cc)))
    A = adjacency("signed", N, rng)
    G = f32exact(rng.uniform(0, 1, (N, N, 50, Cc)))
    _, out, dP = po.time_r50_fwd_bwd(P, A, G)
    _close(out, oracle.contract_forward(50, P, A))
    _close(dP, oracle.contract_backward(50, G, A))
    cases = golden_cases(golden, "r50_")
    assert cases
    for tag, c in cases.items():
        _, out, dP = po.time_r50_fwd_bwd(c["P"], c["A"], c["G"])
        _close(out, c["Out"])
        _close(dP + c["dP0"], c["dP"])


def test_structural_50_collapse(oracle):
    """Known-answer of the reference's own tests/test_RisiContraction_50.cpp: with tensors symmetric in (b,c) and a
    symmetric zero-diagonal 0/1 adjacency the 50 slices fall into exactly the 18 recorded groups (bit-identical)."""
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "structural_50.json")) as fh:
        expected = json.load(fh)["groups"]
    rng = np.random.default_rng(50)
    N, Cc = 10, 5
    P = rng.integers(0, 100, (N, N, N, Cc)).astype(np.float64)
    P = np.triu(P.transpose(0, 3, 1, 2), 0)
    P = (P + np.triu(P, 1).transpose(0, 1, 3, 2)).transpose(0, 2, 3, 1).copy()
    U = np.triu((rng.uniform(0, 1, (N, N)) < 0.5).astype(np.float64), 1)
    A = U + U.T
    out = oracle.contract_forward(50, P, A)
    free, groups = [True] * 50, []
    for i in range(50):
        if free[i]:
            grp = [j + 1 for j in range(i, 50) if np.array_equal(out[:, :, i, :], out[:, :, j, :])]
            for j in grp:
                free[j - 1] = False
            groups.append(grp)
    assert groups == expected
    # and the first member of each group is RisiContraction_18's "(k/50)" label
    assert [g[0] for g in groups] == [oracle.lib.gfo_r18_case_of_50(k) for k in range(18)]


def test_mixers_match_golden(oracle, golden):
    for tag, c in golden_cases(golden, "mm_").items():
        A, B, dC = (c[k].astype(np.float64) for k in ("A", "B", "dC"))
        _close(oracle.matmul_forward(A, B), c["C"])
        dA, dB = oracle.matmul_backward(dC, A, B, c["dA0"].astype(np.float64), c["dB0"].astype(np.float64))
        _close(dA, c["dA"])
        _close(dB, c["dB"])
    for tag, c in golden_cases(golden, "promote_").items():
        X, F, G2 = (c[k].astype(np.float64) for k in ("X", "F", "G2"))
        T1 = oracle.mattensormul_forward(X, F)
        _close(T1, c["T1"])
        T2 = oracle.tensormatmul_forward(T1, X.T.copy())
        _close(T2, c["T2"])
        dT1, dY = oracle.tensormatmul_backward(G2, T1, X.T.copy())
        _close(dT1, c["dT1"])
        _close(dY, c["dY"])
        dX, dF = oracle.mattensormul_backward(dT1, X, F)
        _close(dF, c["dF"])
        _close(dX, c["dX"])


def test_custommatmultensor_matches_golden(oracle, golden):
    cases = golden_cases(golden, "cmix_")
    assert len(cases) == 3
    for tag, c in cases.items():
        W, T, G = (c[k].astype(np.float64) for k in ("W", "T", "G"))
        _close(oracle.custommatmultensor_forward(W, T), c["Out"])
        _close(oracle.custommatmultensor_forward(W, T), np.einsum("kv,ijv->ijk", W, T))
        dW, dT = oracle.custommatmultensor_backward(G, W, T, c["dW0"].astype(np.float64), c["dT0"].astype(np.float64))
        _close(dW, c["dW"])
        _close(dT, c["dT"])


def test_selection_promotion_is_a_gather(golden):
    """SURVEY 8(a8/a9): with a 0/1 selection X, X F X^T is exactly F[pi(i), pi(j), :] or 0."""
    c = golden_cases(golden, "promote_sel")["promote_sel"]
    X, F, T2 = c["X"], c["F"].astype(np.float64), c["T2"]
    s = X.shape[0]
    pi = [int(np.argmax(X[i])) if X[i].any() else -1 for i in range(s)]
    for i in range(s):
        for j in range(s):
            expect = F[pi[i], pi[j]] if pi[i] >= 0 and pi[j] >= 0 else np.zeros(F.shape[2])
            assert np.array_equal(T2[i, j], expect)


def test_oracle_against_live_reference(oracle, reference):
    if reference is None:
        pytest.skip("oracle/_ref/libgf_ref.so not present (reference not mounted when building)")
    rng = np.random.default_rng(99)
    for K in (4, 10, 18, 50):
        for (N, Cc) in ((2, 1), (4, 3), (7, 2)):
            P = rng.uniform(-1, 1, (N, N, N, Cc))
            A = rng.uniform(-1, 1, (N, N))
            G = rng.uniform(-1, 1, (N, N, K, Cc))
            d0 = rng.uniform(-1, 1, (N, N, N, Cc))
            _close(oracle.contract_forward(K, P, A), reference.contract_forward(K, P, A))
            _close(oracle.contract_backward(K, G, A, d0), reference.contract_backward(K, G, A, d0))


def test_r18_dropout_matches_golden(oracle, golden):
    """RisiContraction_18_dropout (RisiContraction_18_dropout.h:106-783): masks drawn by the reference itself."""
    cases = golden_cases(golden, "drop_")
    assert len(cases) == 4
    for tag, c in cases.items():
        seed, nKept, train = (int(x) for x in c["cfg"])
        assert int(c["use"].sum()) == (nKept if train else 18), tag
        P, A, G = (c[k].astype(np.float64) for k in ("P", "A", "G"))
        out, dP, _ = oracle.r18_dropout(c["use"], bool(train), nKept, P, A, G if train else None,
                                        c["dP0"].astype(np.float64) if train else None)
        _close(out, c["Out"])
        if train:
            _close(dP, c["dP"])
            dropped = [k for k in range(18) if not c["use"][k]]
            assert not np.any(c["Out"][:, :, dropped, :])
        else:
            _close(c["Out"], oracle.contract_forward(18, P, A) * (nKept / 18.0))
