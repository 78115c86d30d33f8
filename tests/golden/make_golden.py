#!/usr/bin/env python3
"""Generate tests/golden/*.npz + structural_50.json from the REAL reference.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
It drives oracle/_ref/libgf_ref.so -- the reference's own GraphFlow/ (fp64) op classes compiled by
oracle/Makefile from the headers where they lie -- and, for the structural known-answer, compiles and
runs the reference's own tests/test_RisiContraction_50.cpp and records its stdout.

Fixtures are DATA (inputs + the reference's outputs); no reference source text is stored.
Inputs are float32-representable so the fp32 HIP path and the fp64 checkers see identical numbers.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import pyoracle  # noqa: E402
from inputs import adjacency, er_graph, f32exact, smp_params, synthetic_molecule, toy_molecules  # noqa: E402

REF_ROOT = os.environ.get("GF_REFERENCE", "/root/reference")


def contraction_fixtures(ref):
    out = {}
    shapes = {4: [(3, 2), (5, 3), (8, 4), (16, 8)], 10: [(3, 2), (5, 3), (8, 4)],
              18: [(3, 2), (5, 3), (8, 4), (16, 8)], 50: [(3, 2), (5, 3), (8, 4)]}
    for K, lst in shapes.items():
        for (N, C) in lst:
            rng = np.random.default_rng(1000 * K + 10 * N + C)
            P = f32exact(rng.uniform(-1, 1, (N, N, N, C)))
            G = f32exact(rng.uniform(0, 1, (N, N, K, C)))
            dP0 = f32exact(rng.uniform(-1, 1, (N, N, N, C)))
            kinds = ["none"] if K == 4 else (["sym01"] if (N, C) == (16, 8) else ["sym01", "weighted", "signed"])
            for kind in kinds:
                A = None if K == 4 else adjacency(kind, N, rng)
                tag = "r%d_N%d_C%d_%s" % (K, N, C, kind)
                out[tag + "__P"] = P.astype(np.float32)
                out[tag + "__G"] = G.astype(np.float32)
                out[tag + "__dP0"] = dP0.astype(np.float32)
                if A is not None:
                    out[tag + "__A"] = A.astype(np.float32)
                out[tag + "__Out"] = ref.contract_forward(K, P, A)
                out[tag + "__dP"] = ref.contract_backward(K, G, A, dP0)  # = dP0 + vjp (pins the `+=`)
    # the 6-thread forward (no gate): differs from r18 only on the 'signed' adjacency
    rng = np.random.default_rng(77)
    N, C = 5, 3
    P = f32exact(rng.uniform(-1, 1, (N, N, N, C)))
    A = adjacency("signed", N, rng)
    out["r18thread_N5_C3_signed__P"] = P.astype(np.float32)
    out["r18thread_N5_C3_signed__A"] = A.astype(np.float32)
    out["r18thread_N5_C3_signed__Out"] = ref.r18_thread_forward(P, A)
    return out


def wide_contraction_fixtures(ref):
    """RisiContraction_50 at channel counts that are multiples of 32: the shapes at which the HIP library runs its matrix-pipe
    kernels (fam50_forward_mfma / fam50_bwd_tables_mfma: one wave per (graph, row, 32-channel window)); odd and even N, one and two
    channel windows.  Same recipe as contraction_fixtures."""
    out = {}
    K = 50
    for (N, C, kind) in [(5, 32, "signed"), (6, 32, "weighted"), (3, 64, "signed")]:
        rng = np.random.default_rng(1000 * K + 10 * N + C)
        P = f32exact(rng.uniform(-1, 1, (N, N, N, C)))
        G = f32exact(rng.uniform(0, 1, (N, N, K, C)))
        dP0 = f32exact(rng.uniform(-1, 1, (N, N, N, C)))
        A = adjacency(kind, N, rng)
        tag = "r%d_N%d_C%d_%s" % (K, N, C, kind)
        out[tag + "__P"] = P.astype(np.float32)
        out[tag + "__G"] = G.astype(np.float32)
        out[tag + "__dP0"] = dP0.astype(np.float32)
        out[tag + "__A"] = A.astype(np.float32)
        out[tag + "__Out"] = ref.contract_forward(K, P, A)
        out[tag + "__dP"] = ref.contract_backward(K, G, A, dP0)
    return out


def contraction_16x8_fixtures(ref):
    """SURVEY 8(c) asks for (N, C) = (16, 8) for every K; contractions.npz holds it for `_4` and `_18` (round-5 review, missing #4).
    Here `_10` (0/1 symmetric and signed adjacency: `_10` has no `> 0` gate) and `_50` (0/1 symmetric), same recipe."""
    out = {}
    for K, kinds in ((10, ["sym01", "signed"]), (50, ["sym01"])):
        N, C = 16, 8
        rng = np.random.default_rng(1000 * K + 10 * N + C)
        P = f32exact(rng.uniform(-1, 1, (N, N, N, C)))
        G = f32exact(rng.uniform(0, 1, (N, N, K, C)))
        dP0 = f32exact(rng.uniform(-1, 1, (N, N, N, C)))
        for kind in kinds:
            A = adjacency(kind, N, rng)
            tag = "r%d_N%d_C%d_%s" % (K, N, C, kind)
            out[tag + "__P"] = P.astype(np.float32)
            out[tag + "__G"] = G.astype(np.float32)
            out[tag + "__dP0"] = dP0.astype(np.float32)
            out[tag + "__A"] = A.astype(np.float32)
            out[tag + "__Out"] = ref.contract_forward(K, P, A)
            out[tag + "__dP"] = ref.contract_backward(K, G, A, dP0)
    return out


def stack_fixtures(ref):
    """StackTensor3D (StackTensor3D.h:54-90) as RisiContraction_18_gpu is fed by it (tests/test_RisiContraction_18_gpu.cu:124-129): N
    tensors [N][N][C] -> Tensor4D [N][N][N][C]; backward adds the stacked gradient into the sources' (non-zero) gradients.  Two shapes:
    the cubic one of the contraction and a ragged one (nRows != nCols != n1)."""
    out = {}
    rng = np.random.default_rng(5150)
    for tag, shape in {"stack_5x5x5x3": (5, 5, 5, 3), "stack_3x4x2x7": (3, 4, 2, 7), "stack_1x1x1x1": (1, 1, 1, 1)}.items():
        T = f32exact(rng.uniform(-1, 1, shape))
        G = f32exact(rng.uniform(-1, 1, shape))
        dT0 = f32exact(rng.uniform(-1, 1, shape))
        out[tag + "__T"], out[tag + "__G"], out[tag + "__dT0"] = T.astype(np.float32), G.astype(np.float32), dT0.astype(np.float32)
        out[tag + "__Out"] = ref.stack_forward(T)
        out[tag + "__dT"] = ref.stack_backward(G, dT0)
    return out


def dropout_fixtures(ref):
    """RisiContraction_18_dropout: the reference draws the kept slices itself after srand(seed); the fixture keeps the mask."""
    out = {}
    rng = np.random.default_rng(777)
    for tag, (N, C, seed, nKept, train) in {"drop_train_s2024_k7": (5, 3, 2024, 7, 1), "drop_train_s77_k12": (6, 4, 77, 12, 1),
                                            "drop_train_s1_k1": (4, 2, 1, 1, 1), "drop_test_k7": (5, 3, 3, 7, 0)}.items():
        P = f32exact(rng.uniform(-1, 1, (N, N, N, C)))
        A = f32exact(adjacency("weighted", N, rng)) if "s77" in tag else f32exact((rng.uniform(0, 1, (N, N)) < 0.5).astype(float) + np.eye(N))
        G = f32exact(rng.uniform(-1, 1, (N, N, 18, C)))
        d0 = f32exact(rng.uniform(-1, 1, (N, N, N, C)))
        o, dP, use = ref.r18_dropout(None, bool(train), nKept, P, A, G if train else None, d0 if train else None, seed=seed)
        for k, v in (("P", P), ("A", A), ("G", G), ("dP0", d0)):
            out[tag + "__" + k] = v.astype(np.float32)
        out[tag + "__cfg"] = np.array([seed, nKept, train], dtype=np.int32)
        out[tag + "__use"], out[tag + "__Out"] = use, o
        if train:
            out[tag + "__dP"] = dP
    return out


def selection_matrix(s, sw, rng):
    """0/1 selection X[i,k] = [phi_l(v)_i == phi_{l-1}(w)_k] as SMP builds it (SMP_omega.h:461-474)."""
    X = np.zeros((s, sw))
    rows = rng.permutation(s)[:sw]
    for k, i in enumerate(rows):
        X[i, k] = 1.0
    return X


def mixer_fixtures(ref):
    out = {}
    rng = np.random.default_rng(4242)
    # MatMul at the shape family of tests/test_MatMul_gpu.cu:22-26 (1600x720 . 720x40), down-scaled 10x, plus a K-projection-like one
    for tag, (M, K, N) in {"mm_160x72x40": (160, 72, 40), "mm_kproj_36x144x8": (36, 144, 8), "mm_1x7x1": (1, 7, 1)}.items():
        A = f32exact(rng.uniform(-1, 1, (M, K)))
        B = f32exact(rng.uniform(-1, 1, (K, N)))
        dC = f32exact(rng.uniform(-1, 1, (M, N)))
        dA0 = f32exact(rng.uniform(-1, 1, (M, K)))
        dB0 = f32exact(rng.uniform(-1, 1, (K, N)))
        out[tag + "__A"], out[tag + "__B"], out[tag + "__dC"] = A.astype(np.float32), B.astype(np.float32), dC.astype(np.float32)
        out[tag + "__dA0"], out[tag + "__dB0"] = dA0.astype(np.float32), dB0.astype(np.float32)
        out[tag + "__C"] = ref.matmul_forward(A, B)
        dA, dB = ref.matmul_backward(dC, A, B, dA0, dB0)
        out[tag + "__dA"], out[tag + "__dB"] = dA, dB
    # MatTensorMul / TensorMatMul with a selection matrix (the SMP use) and with a dense matrix
    for tag, dense in (("sel", False), ("dense", True)):
        s, sw, D = 6, 4, 5
        X = f32exact(rng.uniform(-1, 1, (s, sw))) if dense else selection_matrix(s, sw, rng)
        F = f32exact(rng.uniform(-1, 1, (sw, sw, D)))
        T1 = ref.mattensormul_forward(X, F)            # [s, sw, D]
        T2 = ref.tensormatmul_forward(T1, X.T.copy())  # [s, s, D]  = X F X^T
        G2 = f32exact(rng.uniform(-1, 1, (s, s, D)))
        dT1, dY = ref.tensormatmul_backward(G2, T1, X.T.copy())
        dX, dF = ref.mattensormul_backward(dT1, X, F)
        p = "promote_" + tag
        out[p + "__X"], out[p + "__F"], out[p + "__G2"] = X.astype(np.float32), F.astype(np.float32), G2.astype(np.float32)
        out[p + "__T1"], out[p + "__T2"], out[p + "__dT1"], out[p + "__dF"] = T1, T2, dT1, dF
        out[p + "__dX"], out[p + "__dY"] = dX, dY
    # CustomMatMulTensor (CustomMatMulTensor.h:47-85): the SMP_2D_ver6-8 K-projection shape [N,N,18C] -> [N,N,C] and a ragged one
    rng = np.random.default_rng(4243)
    for tag, (I, J, V, Kout) in {"cmix_5x5x36x2": (5, 5, 36, 2), "cmix_3x7x5x4": (3, 7, 5, 4), "cmix_1x1x1x1": (1, 1, 1, 1)}.items():
        W = f32exact(rng.uniform(-1, 1, (Kout, V)))
        T = f32exact(rng.uniform(-1, 1, (I, J, V)))
        G = f32exact(rng.uniform(-1, 1, (I, J, Kout)))
        dW0 = f32exact(rng.uniform(-1, 1, (Kout, V)))
        dT0 = f32exact(rng.uniform(-1, 1, (I, J, V)))
        for k, v in (("W", W), ("T", T), ("G", G), ("dW0", dW0), ("dT0", dT0)):
            out[tag + "__" + k] = v.astype(np.float32)
        out[tag + "__Out"] = ref.custommatmultensor_forward(W, T)
        out[tag + "__dW"], out[tag + "__dT"] = ref.custommatmultensor_backward(G, W, T, dW0, dT0)
    return out


def structural_50():
    """Compile + run the reference's own tests/test_RisiContraction_50.cpp, keep its stdout."""
    src = os.path.join(REF_ROOT, "tests", "test_RisiContraction_50.cpp")
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "t50")
        subprocess.check_call(["g++", "-std=c++11", "-O2", "-pthread", "-w", "-o", exe, src], cwd=os.path.join(REF_ROOT, "tests"))
        txt = subprocess.check_output([exe], cwd=td).decode()
    groups = []
    for line in txt.strip().splitlines():
        head, rest = line.split(":")
        groups.append([int(x) for x in rest.split()])
    return {"source": "stdout of reference tests/test_RisiContraction_50.cpp (N=10, C=5, symmetric inputs, zero-diagonal 0/1 adjacency)",
            "groups": groups}


def smp_cases():
    """(tag, adj, feature, target, cfg) with cfg = (L, C, D, cap, has_wl, max_nVertices)."""
    cases = []
    for name, adj, feat, tgt in toy_molecules():  # the configuration of tests/test_SMP_omega.cpp:22-34
        cases.append(("toy_" + name, adj, feat, tgt, (2, 10, 5, 4, 1, 10)))
    adj, feat = er_graph(20, 0.2, 4, 3)
    cases.append(("er20_wl", adj, feat, 1.5, (2, 8, 3, 10, 1, 20)))
    cases.append(("er20_nowl", adj, feat, 1.5, (2, 8, 3, 10, 0, 20)))
    adj, feat, tgt = synthetic_molecule(5, 12)
    cases.append(("syn12", adj, feat, tgt, (3, 4, 2, 12, 1, 12)))
    adj, feat, tgt = synthetic_molecule(9, 17)
    cases.append(("syn17_cap6", adj, feat, tgt, (3, 4, 2, 6, 1, 17)))
    return cases


def coulomb_case():
    """The use_coulomb constructor (SMP_omega.h:91-113): reduced adjacency = entries of a V x V 'Coulomb' matrix.  The
    synthetic matrix is symmetric with a positive diagonal and some NEGATIVE / zero off-diagonal entries, so the `A > 0`
    gate of RisiContraction_18 matters end to end."""
    adj, feat, tgt = synthetic_molecule(21, 11)
    rng = np.random.default_rng(2121)
    V = len(adj)
    M = rng.uniform(-0.5, 2.0, (V, V))
    M = f32exact(0.5 * (M + M.T))
    M[rng.uniform(0, 1, (V, V)) < 0.15] = 0.0
    M = np.minimum(M, M.T)
    np.fill_diagonal(M, f32exact(rng.uniform(1.0, 3.0, V)))
    return ("syn11_coulomb", adj, feat, tgt, (2, 4, 2, 8, 1, 11)), M


def smp_fixtures():
    out = {}
    ccase, cmat = coulomb_case()
    for i, (tag, adj, feat, tgt, (L, C, D, cap, wl, maxV)) in enumerate(smp_cases() + [ccase]):
        F = feat.shape[1]
        params = smp_params(C, F, D, L, 100 + i)
        coul = cmat if tag == ccase[0] else None
        r = pyoracle.reference_smp_omega(adj, feat, tgt, params, L, C, D, cap, has_wl=bool(wl), max_nVertices=maxV, coulomb=coul)
        if coul is not None:
            out["smp_" + tag + "__coulomb"] = coul
        V = len(adj)
        phi = np.full((L + 1, V, cap + 1), -1, dtype=np.int32)
        for l in range(L + 1):
            for v in range(V):
                phi[l, v, 0] = len(r["phi"][l][v])
                phi[l, v, 1:1 + len(r["phi"][l][v])] = r["phi"][l][v]
        p = "smp_" + tag
        out[p + "__adj"], out[p + "__feature"], out[p + "__target"] = adj.astype(np.int32), feat, np.array([tgt])
        out[p + "__cfg"] = np.array([L, C, D, cap, wl], dtype=np.int32)
        out[p + "__params"] = params.astype(np.float32)
        out[p + "__phi"], out[p + "__graph_feature"] = phi, r["graph_feature"]
        out[p + "__predict"], out[p + "__loss"], out[p + "__grads"] = np.array([r["predict"]]), np.array([r["loss"]]), r["grads"]
    # SMP_beta (GraphFlow/SMP_beta.h) = the model without the receptive-field cap: outputs from the real SMP_beta, fields from
    # SMP_omega run with the cap at max_nVertices (SMP_beta does not expose a different construction of them)
    adj, feat, tgt = synthetic_molecule(31, 13)
    L, C, D = 3, 4, 2
    params = smp_params(C, feat.shape[1], D, L, 131)
    rb = pyoracle.reference_smp_beta(adj, feat, tgt, params, L, C, D, True, 13)
    ro = pyoracle.reference_smp_omega(adj, feat, tgt, params, L, C, D, 13, has_wl=True, max_nVertices=13)
    V = len(adj)
    phi = np.full((L + 1, V, 14), -1, dtype=np.int32)
    for l in range(L + 1):
        for v in range(V):
            phi[l, v, 0] = len(ro["phi"][l][v])
            phi[l, v, 1:1 + len(ro["phi"][l][v])] = ro["phi"][l][v]
    # SMP_2D_ver6 / ver7 / ver8 (RisiContraction_10 / _50 / _18 + CustomMatMulTensor, K_l = [C][nContractions C])
    adj2, feat2, tgt2 = synthetic_molecule(41, 9)
    for ver, nK in ((6, 10), (7, 50), (8, 18)):
        L2, C2, D2 = 2, 4, 2
        n = C2 * feat2.shape[1] * (D2 + 1) + L2 * (nK * C2 * C2 + C2) + C2
        prm = f32exact(np.random.default_rng(600 + ver).uniform(-0.3, 0.3, n))
        r2 = pyoracle.reference_smp_2d(ver, adj2, feat2, tgt2, prm, L2, C2, D2, True, len(adj2))
        q = "smp_2dver%d_syn9" % ver
        out[q + "__adj"], out[q + "__feature"], out[q + "__target"] = adj2.astype(np.int32), feat2, np.array([tgt2])
        out[q + "__cfg"] = np.array([L2, C2, D2, len(adj2), 1], dtype=np.int32)
        out[q + "__wiring"] = np.array([nK, 1], dtype=np.int32)
        out[q + "__params"] = prm.astype(np.float32)
        out[q + "__phi"], out[q + "__graph_feature"] = r2["phi_array"], r2["graph_feature"]
        out[q + "__predict"], out[q + "__loss"], out[q + "__grads"] = np.array([r2["predict"]]), np.array([r2["loss"]]), r2["grads"]
    p = "smp_beta_syn13"
    out[p + "__adj"], out[p + "__feature"], out[p + "__target"] = adj.astype(np.int32), feat, np.array([tgt])
    out[p + "__cfg"] = np.array([L, C, D, 13, 1], dtype=np.int32)
    out[p + "__params"] = params.astype(np.float32)
    out[p + "__phi"], out[p + "__graph_feature"] = phi, rb["graph_feature"]
    out[p + "__predict"], out[p + "__loss"], out[p + "__grads"] = np.array([rb["predict"]]), np.array([rb["loss"]]), rb["grads"]
    return out


def big_field_fixtures():
    """Round 6: SMP_beta (no receptive-field cap) on a 40-atom molecule whose level-3 fields reach 35 positions -- beyond the 32 the fused
    level's register classes and row panels take, the case smp_fused.hip's big_part serves -- from the REAL SMP_beta (fields from SMP_omega
    with the cap at max_nVertices, as for smp_beta_syn13), at 8 channels (computed at 16 padded) and at 32."""
    out = {}
    adj, feat, tgt = synthetic_molecule(8003, 40)
    L, D = 3, 2
    V = len(adj)
    for C in (8, 32):
        params = smp_params(C, feat.shape[1], D, L, 140 + C)
        rb = pyoracle.reference_smp_beta(adj, feat, tgt, params, L, C, D, True, V)
        ro = pyoracle.reference_smp_omega(adj, feat, tgt, params, L, C, D, V, has_wl=True, max_nVertices=V)
        phi = np.full((L + 1, V, V + 1), -1, dtype=np.int32)
        for l in range(L + 1):
            for v in range(V):
                phi[l, v, 0] = len(ro["phi"][l][v])
                phi[l, v, 1:1 + len(ro["phi"][l][v])] = ro["phi"][l][v]
        assert phi[L, :, 0].max() > 32
        p = "smp_beta_big40_C%d" % C
        out[p + "__adj"], out[p + "__feature"], out[p + "__target"] = adj.astype(np.int32), feat, np.array([tgt])
        out[p + "__cfg"] = np.array([L, C, D, V, 1], dtype=np.int32)
        out[p + "__params"] = params.astype(np.float32)
        out[p + "__phi"], out[p + "__graph_feature"] = phi, rb["graph_feature"]
        out[p + "__predict"], out[p + "__loss"], out[p + "__grads"] = np.array([rb["predict"]]), np.array([rb["loss"]]), rb["grads"]
    return out


def train_fixture():
    """Three SMP_omega::BatchLearn steps of the real reference on the four toy molecules of tests/test_SMP_omega.cpp as one batch,
    starting from the weights its own constructor draws after srand(7) (weights_initialization)."""
    mols, tgts = [], []
    for name, adj, feat, tgt in toy_molecules():
        mols.append((adj, feat))
        tgts.append(tgt)
    L, C, D, cap, maxV = 2, 10, 5, 4, 10
    r = pyoracle.reference_batchlearn(mols, tgts, L, C, D, cap, maxV, nIter=3, learning_rate=1e-3, seed=7)
    out = {"train__cfg": np.array([L, C, D, cap, maxV, 7, 3], dtype=np.int32), "train__lr": np.array([1e-3]),
           "train__targets": np.array(tgts, dtype=np.float64), "train__params0": r["params0"], "train__params": r["params"],
           "train__losses": r["losses"]}
    # SMP_2D_ver6 (RisiContraction_10, CustomMatMulTensor weights, Momentum 0.9): three BatchLearn steps from srand(11)
    r6 = pyoracle.reference_smp_2d_batchlearn(6, mols, tgts, 2, 6, 3, 10, 0.9, 3, 1e-5, 11)
    out.update({"train2d6__cfg": np.array([2, 6, 3, 10, 11, 3], dtype=np.int32), "train2d6__lr": np.array([1e-5]),
                "train2d6__momentum": np.array([0.9]), "train2d6__params0": r6["params0"], "train2d6__params": r6["params"],
                "train2d6__losses": r6["losses"]})
    np.savez_compressed(os.path.join(HERE, "smp_train.npz"), **out)


def activation_digest(f):
    """Two linear functionals of an activation tensor f [s, s, C] per channel: the plain sum over (i, j) and a sum with
    position-dependent weights (detects transposed / permuted positions).  Weights are float32-exact by construction."""
    s = f.shape[0]
    i, j = np.meshgrid(np.arange(s), np.arange(s), indexing="ij")
    w = ((3 * i + 5 * j) % 7 - 3).astype(np.float64) / 4.0
    return np.stack([f.sum(axis=(0, 1)), (w[:, :, None] * f).sum(axis=(0, 1))])


def headline_fixture():
    """BASELINE configs[2]'s own shape pinned to the REAL reference: one 29-atom synthetic molecule through SMP_omega at
    L = 3, receptive-field cap 29, C = 64, F = 5, D = 5 (GraphFlow/SMP_omega.h:584-693; about 8 s of reference time).
    Kept: receptive fields, reduced adjacencies, Feature(), predict, loss, all 223,360 parameter gradients (fp32 storage),
    and the per-level activations f[l][v] -- in full for levels 0-1 and for two vertices of levels 2 and 3 (the largest and
    the smallest field), as two per-channel linear digests (activation_digest) for every vertex of every level.
    The parameters are smp_params(64, 5, 5, 3, seed 2929) (float32-exact, regenerated by the test; their checksum is kept)."""
    L, C, F, D, cap, seed_p = 3, 64, 5, 5, 29, 2929
    adj, feat, tgt = synthetic_molecule(29029, 29)
    V = len(adj)
    params = smp_params(C, F, D, L, seed_p)
    r = pyoracle.reference_smp_omega(adj, feat, tgt, params, L, C, D, cap, has_wl=True, max_nVertices=V, want_activations=True)
    phi = np.full((L + 1, V, cap + 1), -1, dtype=np.int32)
    radj = np.zeros((L + 1, V, cap, cap), dtype=np.int8)
    sizes = np.zeros((L + 1, V), dtype=np.int32)
    for l in range(L + 1):
        for v in range(V):
            n = len(r["phi"][l][v])
            sizes[l, v] = n
            phi[l, v, 0] = n
            phi[l, v, 1:1 + n] = r["phi"][l][v]
            if l > 0:
                a = r["reduced_adj"][l][v]
                assert np.array_equal(a, a.astype(np.int8)), "0/1 adjacency expected"
                radj[l, v, :n, :n] = a
    out = {"headline__adj": adj.astype(np.int32), "headline__feature": feat, "headline__target": np.array([tgt]),
           "headline__cfg": np.array([L, C, D, cap, 1, seed_p], dtype=np.int32),
           "headline__params_checksum": np.array([params.sum(), np.abs(params).sum(), (params * np.arange(params.size)).sum()]),
           "headline__phi": phi, "headline__reduced_adj": radj, "headline__graph_feature": r["graph_feature"],
           "headline__predict": np.array([r["predict"]]), "headline__loss": np.array([r["loss"]]),
           "headline__grads": r["grads"].astype(np.float32)}
    dig = np.zeros((L + 1, V, 2, C))
    for l in range(L + 1):
        for v in range(V):
            dig[l, v] = activation_digest(r["f"][l][v])
    out["headline__act_digest"] = dig
    for l in (0, 1):
        out["headline__act_level%d" % l] = np.concatenate([r["f"][l][v].ravel() for v in range(V)]).astype(np.float32)
    picks = []
    for l in (2, 3):
        vmax, vmin = int(np.argmax(sizes[l])), int(np.argmin(sizes[l]))
        for v in (vmax, vmin):
            picks.append((l, v))
            out["headline__act_l%d_v%d" % (l, v)] = r["f"][l][v].astype(np.float32)
    out["headline__act_picks"] = np.array(picks, dtype=np.int32)
    print("headline molecule: V = %d, max field sizes per level %s, predict %.6g" % (V, sizes.max(axis=1).tolist(), r["predict"]))
    np.savez_compressed(os.path.join(HERE, "smp_headline.npz"), **out)


def physics_fixtures():
    """the `_physics` (one tower) and `_pairgraphs` (two towers) drivers of the real reference -- raw features, the
    distance-only cap order, channels halving per level, every level read out, MLP heads -- and SMP_sigma_pairgraphs with
    RisiContraction_18_dropout in train (masks drawn after srand) and test mode.  One sample each, dumped parameters in the
    class's registration order; kept: receptive fields, the concatenated feature row, predict, loss, all parameter gradients."""
    out = {}
    rng = np.random.default_rng(8303)
    g12 = synthetic_molecule(5, 12)
    g9 = synthetic_molecule(6, 9)
    g17 = synthetic_molecule(9, 17)

    def tower_n(C, F, L):
        return pyoracle.physics_tower_param_count(C, F, L)

    for tag, (g, L, C, cap, beta) in {"phys_omega_cap6": (g12, 2, 8, 6, False), "phys_omega_L3_c10": (g17, 3, 10, 8, False),
                                      "phys_beta": (g12, 2, 8, 12, True)}.items():
        adj, feat, tgt = g
        F = feat.shape[1]
        width = sum(pyoracle.physics_channels(C, L))
        nh = width // 2
        n = tower_n(C, F, L) + nh * width + nh
        params = f32exact(rng.uniform(-0.3, 0.3, n))
        r = pyoracle.reference_smp_physics(adj, feat, tgt, params, L, C, cap, beta=beta, max_nVertices=len(adj))
        p = "physics_" + tag
        out[p + "__adj"], out[p + "__feature"], out[p + "__target"] = adj.astype(np.int32), feat, np.array([tgt])
        out[p + "__cfg"] = np.array([1, L, C, cap, 0, 1], dtype=np.int32)   # towers, L, C, cap, nKept, train
        out[p + "__params"] = params.astype(np.float32)
        out[p + "__phi"], out[p + "__graph_feature"] = r["phi"], r["graph_feature"]
        out[p + "__predict"], out[p + "__loss"], out[p + "__grads"] = np.array([r["predict"]]), np.array([r["loss"]]), r["grads"]
    for tag, (kind, ga, gb, L, C, cap, nKept, train, seed) in {"pair_omega": (0, g12, g9, 2, 8, 6, 0, 1, 0), "pair_beta": (1, g9, g12, 2, 8, 12, 0, 1, 0),
                                                               "pair_sigma_train": (2, g12, g9, 2, 8, 6, 7, 1, 5),
                                                               "pair_sigma_test": (2, g12, g9, 2, 8, 6, 7, 0, 5)}.items():
        F1, F2 = ga[1].shape[1], gb[1].shape[1]
        nTot = 2 * sum(pyoracle.physics_channels(C, L))
        h1 = max(nTot // 2, 10)
        h2 = max(h1 // 2, 10)
        n = tower_n(C, F1, L) + tower_n(C, F2, L) + h1 * nTot + h2 * h1 + h2
        params = f32exact(rng.uniform(-0.3, 0.3, n))
        tgt = ga[2]
        maxV = max(len(ga[0]), len(gb[0]))
        r = pyoracle.reference_smp_pairgraphs(kind, ga[:2], gb[:2], tgt, params, L, C, cap, nKept=max(nKept, 1), train=bool(train), seed=seed, maxV=maxV)
        p = "physics_" + tag
        out[p + "__adj"], out[p + "__feature"], out[p + "__target"] = ga[0].astype(np.int32), ga[1], np.array([tgt])
        out[p + "__adj2"], out[p + "__feature2"] = gb[0].astype(np.int32), gb[1]
        out[p + "__cfg"] = np.array([2, L, C, cap, nKept, train], dtype=np.int32)
        out[p + "__seed"] = np.array([seed], dtype=np.int32)
        out[p + "__params"] = params.astype(np.float32)
        out[p + "__phi"], out[p + "__phi2"], out[p + "__graph_feature"] = r["phi1"], r["phi2"], r["graph_feature"]
        out[p + "__predict"], out[p + "__loss"], out[p + "__grads"] = np.array([r["predict"]]), np.array([r["loss"]]), r["grads"]
    # three BatchLearn steps of the real SMP_omega_physics / SMP_omega_pairgraphs on the toy molecules of the reference's
    # tests/test_SMP_omega_physics.cpp / test_SMP_omega_pairgraphs.cpp (all 16 ordered pairs, target = difference of the atom
    # counts), from the weights the constructors draw after srand(7)
    mols = [(a, f) for _, a, f, _ in toy_molecules()]
    tg = [t for *_, t in toy_molecules()]
    L, C, cap, maxV = 2, 16, 4, 10
    nt = tower_n(C, 4, L)
    w = sum(pyoracle.physics_channels(C, L))
    r = pyoracle.reference_model_batchlearn(0, mols, None, tg, L, C, cap, maxV, 3, 1e-3, 7, nt + (w // 2) * w + w // 2)
    out.update({"trainphys__cfg": np.array([L, C, cap, maxV, 7, 3], dtype=np.int32), "trainphys__lr": np.array([1e-3]),
                "trainphys__targets": np.array(tg), "trainphys__params0": r["params0"], "trainphys__params": r["params"],
                "trainphys__losses": r["losses"]})
    g1 = [mols[i] for i in range(4) for j in range(4)]
    g2 = [mols[j] for i in range(4) for j in range(4)]
    t2 = [tg[i] - tg[j] for i in range(4) for j in range(4)]
    nTot = 2 * w
    h1 = max(nTot // 2, 10)
    h2 = max(h1 // 2, 10)
    r = pyoracle.reference_model_batchlearn(10, g1, g2, t2, L, C, cap, maxV, 3, 1e-3, 7, 2 * nt + h1 * nTot + h2 * h1 + h2)
    out.update({"trainpair__cfg": np.array([L, C, cap, maxV, 7, 3], dtype=np.int32), "trainpair__lr": np.array([1e-3]),
                "trainpair__targets": np.array(t2), "trainpair__params0": r["params0"], "trainpair__params": r["params"],
                "trainpair__losses": r["losses"]})
    np.savez_compressed(os.path.join(HERE, "smp_physics.npz"), **out)


def physics_big_fixtures():
    """Round 6: the real SMP_beta_physics and SMP_sigma_pairgraphs (slice dropout, train mode) on 40-atom graphs whose level-3 fields exceed
    32 positions -- the towers' fused levels with nodes above 32 (smp_fused.hip: big_part).  Same recipe as physics_fixtures."""
    out = {}
    rng = np.random.default_rng(8304)
    g40a, g40b = synthetic_molecule(8003, 40), synthetic_molecule(8017, 40)
    L, C, cap = 3, 8, 40
    adj, feat, tgt = g40a
    F = feat.shape[1]
    width = sum(pyoracle.physics_channels(C, L))
    nh = width // 2
    params = f32exact(rng.uniform(-0.3, 0.3, pyoracle.physics_tower_param_count(C, F, L) + nh * width + nh))
    r = pyoracle.reference_smp_physics(adj, feat, tgt, params, L, C, cap, beta=True, max_nVertices=len(adj))
    assert int(np.asarray(r["phi"])[L, :, 0].max()) > 32
    p = "physics_phys_beta_big40"
    out[p + "__adj"], out[p + "__feature"], out[p + "__target"] = adj.astype(np.int32), feat, np.array([tgt])
    out[p + "__cfg"] = np.array([1, L, C, cap, 0, 1], dtype=np.int32)
    out[p + "__params"] = params.astype(np.float32)
    out[p + "__phi"], out[p + "__graph_feature"] = r["phi"], r["graph_feature"]
    out[p + "__predict"], out[p + "__loss"], out[p + "__grads"] = np.array([r["predict"]]), np.array([r["loss"]]), r["grads"]
    nKept, seed = 9, 5
    nTot = 2 * width
    h1 = max(nTot // 2, 10)
    h2 = max(h1 // 2, 10)
    params = f32exact(rng.uniform(-0.3, 0.3, 2 * pyoracle.physics_tower_param_count(C, F, L) + h1 * nTot + h2 * h1 + h2))
    r = pyoracle.reference_smp_pairgraphs(2, g40a[:2], g40b[:2], g40a[2], params, L, C, cap, nKept=nKept, train=True, seed=seed, maxV=40)
    p = "physics_pair_sigma_big40_train"
    out[p + "__adj"], out[p + "__feature"], out[p + "__target"] = g40a[0].astype(np.int32), g40a[1], np.array([g40a[2]])
    out[p + "__adj2"], out[p + "__feature2"] = g40b[0].astype(np.int32), g40b[1]
    out[p + "__cfg"] = np.array([2, L, C, cap, nKept, 1], dtype=np.int32)
    out[p + "__seed"] = np.array([seed], dtype=np.int32)
    out[p + "__params"] = params.astype(np.float32)
    out[p + "__phi"], out[p + "__phi2"], out[p + "__graph_feature"] = r["phi1"], r["phi2"], r["graph_feature"]
    out[p + "__predict"], out[p + "__loss"], out[p + "__grads"] = np.array([r["predict"]]), np.array([r["loss"]]), r["grads"]
    np.savez_compressed(os.path.join(HERE, "smp_physics_big.npz"), **out)


def checkpoint_fixture():
    """smp_syn12's parameters as SMP_omega::save_model writes them (SMP_omega.h:1033-1042): a data file, 6 significant digits."""
    for i, (tag, adj, feat, tgt, (L, C, D, cap, wl, maxV)) in enumerate(smp_cases()):
        if tag == "syn12":
            params = smp_params(C, feat.shape[1], D, L, 100 + i)
            pyoracle.reference_save_model(os.path.join(HERE, "smp_syn12_checkpoint.txt"), params.astype(np.float32), L, C,
                                          feat.shape[1], D, cap, maxV)


def main():
    pyoracle.build()
    ref = pyoracle.reference()
    if ref is None:
        sys.exit("oracle/_ref/libgf_ref.so missing: needs /root/reference (build container only)")
    np.savez_compressed(os.path.join(HERE, "contractions.npz"), **contraction_fixtures(ref))
    np.savez_compressed(os.path.join(HERE, "contractions_wide.npz"), **wide_contraction_fixtures(ref))
    np.savez_compressed(os.path.join(HERE, "mixers.npz"), **mixer_fixtures(ref))
    np.savez_compressed(os.path.join(HERE, "contractions_16x8.npz"), **contraction_16x8_fixtures(ref))
    np.savez_compressed(os.path.join(HERE, "stack.npz"), **stack_fixtures(ref))
    np.savez_compressed(os.path.join(HERE, "dropout.npz"), **dropout_fixtures(ref))
    np.savez_compressed(os.path.join(HERE, "smp.npz"), **smp_fixtures())
    np.savez_compressed(os.path.join(HERE, "smp_big.npz"), **big_field_fixtures())
    checkpoint_fixture()
    train_fixture()
    physics_fixtures()
    physics_big_fixtures()
    headline_fixture()
    with open(os.path.join(HERE, "structural_50.json"), "w") as fh:
        json.dump(structural_50(), fh, indent=1)
    for f in ("contractions.npz", "contractions_wide.npz", "mixers.npz", "smp.npz", "structural_50.json", "smp_syn12_checkpoint.txt", "dropout.npz", "smp_train.npz", "smp_headline.npz", "smp_physics.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "physics":
        pyoracle.build()
        physics_fixtures()
        print("smp_physics.npz", os.path.getsize(os.path.join(HERE, "smp_physics.npz")), "bytes")
    elif len(sys.argv) > 1 and sys.argv[1] == "r06":    # only the fixtures added in round 6: (16, 8) for `_10` / `_50`, StackTensor3D
        pyoracle.build()
        np.savez_compressed(os.path.join(HERE, "contractions_16x8.npz"), **contraction_16x8_fixtures(pyoracle.reference()))
        np.savez_compressed(os.path.join(HERE, "stack.npz"), **stack_fixtures(pyoracle.reference()))
        np.savez_compressed(os.path.join(HERE, "smp_big.npz"), **big_field_fixtures())
        physics_big_fixtures()
        for f in ("contractions_16x8.npz", "stack.npz", "smp_big.npz", "smp_physics_big.npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
    elif len(sys.argv) > 1 and sys.argv[1] == "wide":   # only the RisiContraction_50 fixtures at C % 32 == 0
        pyoracle.build()
        np.savez_compressed(os.path.join(HERE, "contractions_wide.npz"), **wide_contraction_fixtures(pyoracle.reference()))
        print("contractions_wide.npz", os.path.getsize(os.path.join(HERE, "contractions_wide.npz")), "bytes")
    elif len(sys.argv) > 1 and sys.argv[1] == "headline":   # only the (slow) headline fixture
        pyoracle.build()
        headline_fixture()
        print("smp_headline.npz", os.path.getsize(os.path.join(HERE, "smp_headline.npz")), "bytes")
    else:
        main()
