"""fp64 restatement of GraphFlow's SMP_omega forward/backward for ONE molecule (numpy + the C oracle's contraction).

TEST INFRASTRUCTURE ONLY (see oracle/gf_oracle.c header).  Parity status: PINNED -- tests/test_smp_oracle.py checks this
against golden vectors captured from the real reference SMP_omega (tests/golden/smp.npz, generator
tests/golden/make_golden.py -> oracle/ref_shim.cpp: ref_smp_omega_run) and, when oracle/_ref is present, against the
reference itself on random molecules.  Pure-Python loops: small molecules only.

Every step cites the reference lines it follows (GraphFlow/SMP_omega.h unless noted).
"""
import numpy as np

from . import pyoracle

INF = 10 ** 9  # :1064
ALPHA = 0.01   # LeakyReLU3D.h:41
K18 = 18


def hop_distances(adj):
    """:358-380 (initialisation order matters for asymmetric inputs: an edge writes both [i][j] and [j][i])."""
    V = len(adj)
    sp = [[0] * V for _ in range(V)]
    for i in range(V):
        for j in range(V):
            sp[i][j] = 0 if i == j else INF
            if i != j and adj[i][j] > 0:
                sp[i][j] = 1
                sp[j][i] = 1
    for k in range(V):
        for i in range(V):
            for j in range(V):
                sp[i][j] = min(sp[i][j], sp[i][k] + sp[k][j])
    return sp


def wl_features(feature, sp, nDepth):
    """:382-404."""
    V, F = feature.shape
    h = np.zeros((V, F * (nDepth + 1)))
    for v in range(V):
        for d in range(nDepth + 1):
            for u in range(V):
                if sp[u][v] == d:
                    h[v, d * F:(d + 1) * F] += feature[u]
    return h


def rank_vertices(h):
    """:406-434: exchange sort into descending lexicographic order."""
    V = len(h)

    def cmp(u, v):
        for a, b in zip(h[u], h[v]):
            if a < b:
                return -1
            if a > b:
                return 1
        return 0

    order = list(range(V))
    for i in range(V):
        for j in range(i + 1, V):
            if cmp(order[i], order[j]) < 0:
                order[i], order[j] = order[j], order[i]
    rank = [0] * V
    for i, v in enumerate(order):
        rank[v] = i
    return rank


def receptive_fields(sp, rank, nLevels, cap, has_wl):
    """:509-537 with limit_receptive_field :476-507 and sort :451-459."""
    V = len(sp)
    phi = [[[v] for v in range(V)]]
    for l in range(1, nLevels + 1):
        cur = []
        for v in range(V):
            A = []
            for u in range(V):
                if sp[u][v] <= 1:
                    for x in phi[l - 1][u]:
                        if x not in A:
                            A.append(x)
            if len(A) > cap:
                for i in range(len(A)):
                    for j in range(i + 1, len(A)):
                        di, dj = sp[v][A[i]], sp[v][A[j]]
                        if di > dj or (di == dj and rank[A[i]] > rank[A[j]]):
                            A[i], A[j] = A[j], A[i]
                while len(A) > cap:
                    d = sp[v][A[-1]]
                    while A and sp[v][A[-1]] == d:
                        A.pop()
            if has_wl:
                for i in range(len(A)):
                    for j in range(i + 1, len(A)):
                        if rank[A[i]] > rank[A[j]]:
                            A[i], A[j] = A[j], A[i]
            cur.append(A)
        phi.append(cur)
    return phi


def lrelu(z):
    return np.where(z > 0, z, ALPHA * z)


def split_params(params, C, FD, L, nK=K18, custom=False):
    """Registration order :289-295: H[C,FD], (K_l, b_l[C]) for l=1..L, W[C].  K_l is [nK C, C] (SMP_omega: Reshape2D +
    MatMul) or, with custom=True, [C, nK C] (SMP_2D_ver6-8: CustomMatMulTensor, SMP_2D_ver6.h:130); returned as [nK C, C]."""
    p = np.asarray(params, dtype=np.float64)
    o = 0
    H = p[o:o + C * FD].reshape(C, FD)
    o += C * FD
    K, b = [None], [None]
    for _ in range(L):
        blk = p[o:o + nK * C * C]
        K.append(blk.reshape(C, nK * C).T.copy() if custom else blk.reshape(nK * C, C))
        o += nK * C * C
        b.append(p[o:o + C])
        o += C
    W = p[o:o + C]
    assert o + C == p.size
    return H, K, b, W


def run(adj, feature, target, params, nLevels, C, nDepth, cap, has_wl=True, want_grads=True, coulomb=None, nK=K18,
        custom=False):
    """One molecule through SMP_omega::complete_computation_graph + forward (+ backward).
    nK in {10, 18, 50} selects the contraction family and custom=True the CustomMatMulTensor weight layout: together they
    are SMP_2D_ver6 (10) / ver7 (50) / ver8 (18) (GraphFlow/SMP_2D_ver6.h:456-560), which have no receptive-field cap.
    Returns dict(phi, reduced_adj, graph_feature, predict, loss, grads)."""
    orc = pyoracle.oracle()
    adj = np.asarray(adj)
    feature = np.asarray(feature, dtype=np.float64)
    V, F = feature.shape
    FD = F * (nDepth + 1)
    L = nLevels
    H, K, b, W = split_params(params, C, FD, L, nK, custom)
    sp = hop_distances(adj.tolist())
    x = wl_features(feature, sp, nDepth)
    rank = rank_vertices(x.tolist())
    phi = receptive_fields(sp, rank, L, cap, has_wl)

    # level 0 (:617-626): MatMul(H, x_v) -> reshape (1,1,C) -> LeakyReLU3D
    z = [[None] * V for _ in range(L + 1)]   # pre-activations
    f = [[None] * V for _ in range(L + 1)]
    for v in range(V):
        z[0][v] = (H @ x[v]).reshape(1, 1, C)
        f[0][v] = lrelu(z[0][v])
    X = [[dict() for _ in range(V)] for _ in range(L + 1)]
    Ared = [[None] * V for _ in range(L + 1)]
    Pst = [[None] * V for _ in range(L + 1)]
    Q = [[None] * V for _ in range(L + 1)]
    for l in range(1, L + 1):
        for v in range(V):
            fld = phi[l][v]
            s = len(fld)
            # reduced adjacency (:556-581, adjacency mode)
            if coulomb is None:
                A = np.array([[1.0 if fld[i] == fld[j] else float(adj[fld[i]][fld[j]]) for j in range(s)] for i in range(s)])
            else:   # use_coulomb (:568-579): the Coulomb entries, diagonal included
                A = np.array([[float(coulomb[fld[i]][fld[j]]) for j in range(s)] for i in range(s)])
            Ared[l][v] = A
            P = np.zeros((s, s, s, C))
            for a, w in enumerate(fld):
                wf = phi[l - 1][w]
                Xm = np.array([[1.0 if fld[i] == wf[k] else 0.0 for k in range(len(wf))] for i in range(s)])  # :461-474
                X[l][v][w] = Xm
                T1 = np.einsum("ik,kjd->ijd", Xm, f[l - 1][w])          # MatTensorMul.h:47-68
                P[a] = np.einsum("ikd,kj->ijd", T1, Xm.T)               # TensorMatMul.h:46-67
            Pst[l][v] = P
            Q[l][v] = orc.contract_forward(nK, P, A)                    # RisiContraction_18.h:73-331 (or _10 / _50)
            z[l][v] = (Q[l][v].reshape(s * s, nK * C) @ K[l]).reshape(s, s, C) + b[l]   # :654-666
            f[l][v] = lrelu(z[l][v])
    sh = np.stack([f[L][v].sum(axis=(0, 1)) for v in range(V)])         # ShrinkTensor.h:37-50
    vf = lrelu(sh)
    g = vf.sum(axis=0)                                                   # SumVectors
    y = float(g @ W)
    loss = 0.5 * (y - target) ** 2
    out = {"phi": phi, "reduced_adj": Ared, "graph_feature": g, "predict": y, "loss": loss, "x": x}
    if not want_grads:
        return out

    # reverse sweep (GraphFlow.h:729)
    dH, dW = np.zeros_like(H), np.zeros_like(W)
    dK = [None] + [np.zeros_like(K[l]) for l in range(1, L + 1)]
    db = [None] + [np.zeros_like(b[l]) for l in range(1, L + 1)]
    dy = y - target                                                      # SquaredLoss.h:55-61
    dW += dy * g                                                         # InnerProduct.h:48-53
    dg = dy * W
    df = [[np.zeros_like(f[l][v]) for v in range(V)] for l in range(L + 1)]
    for v in range(V):
        dsh = dg * np.where(sh[v] > 0, 1.0, ALPHA)
        df[L][v] += dsh[None, None, :]                                   # ShrinkTensor.h:52-61
    for l in range(L, 0, -1):
        for v in reversed(range(V)):
            fld = phi[l][v]
            s = len(fld)
            dz = df[l][v] * np.where(z[l][v] > 0, 1.0, ALPHA)            # LeakyReLU3D.h:60-68
            db[l] += dz.sum(axis=(0, 1))                                 # VectorAddTensor.h:61-72
            dz2 = dz.reshape(s * s, C)
            Q2 = Q[l][v].reshape(s * s, nK * C)
            dK[l] += Q2.T @ dz2                                          # MatMul.h:69-82
            dQ = (dz2 @ K[l].T).reshape(s, s, nK, C)
            dP = orc.contract_backward(nK, dQ, Ared[l][v])               # RisiContraction_18.h:333-560
            for a, w in enumerate(fld):
                Xm = X[l][v][w]
                dT1 = np.einsum("ijd,kj->ikd", dP[a], Xm.T)              # TensorMatMul.h:69-84 (first operand)
                df[l - 1][w] += np.einsum("ik,ijd->kjd", Xm, dT1)        # MatTensorMul.h:70-85 (second operand)
    for v in range(V):
        dz0 = (df[0][v] * np.where(z[0][v] > 0, 1.0, ALPHA)).reshape(C)
        dH += np.outer(dz0, x[v])                                        # MatMul.h:69-82
    grads = [dH.ravel()]
    for l in range(1, L + 1):
        grads += [(dK[l].T if custom else dK[l]).ravel(), db[l].ravel()]
    grads.append(dW.ravel())
    out["grads"] = np.concatenate(grads)
    return out


def adam_learn(params, grads_sum, m, v, n0, alpha, nBatch, beta1=0.9, beta2=0.999, eps=1e-8):
    """Adam::Learn(alpha, nBatch) (GraphFlow/Adam.h:106-133) on the flat parameter vector, fp64.  The reference multiplies
    its bias-correction powers once PER ELEMENT (:121,:125), carried across calls: element i uses beta^(n0 + i + 1).
    Returns (params, m, v, n0 + len(params)); the powers are formed by the same running products."""
    p = np.array(params, dtype=np.float64)
    m = np.array(m, dtype=np.float64)
    v = np.array(v, dtype=np.float64)
    # running products from the start of training (n0 factors already applied)
    b1t, b2t = 1.0, 1.0
    for _ in range(int(n0)):
        b1t *= beta1
        b2t *= beta2
    for i in range(p.size):
        g = grads_sum[i] / nBatch
        m[i] = beta1 * m[i] + (1.0 - beta1) * g
        v[i] = beta2 * v[i] + (1.0 - beta2) * g * g
        b1t *= beta1
        mh = m[i] / (1.0 - b1t)
        b2t *= beta2
        vh = v[i] / (1.0 - b2t)
        p[i] -= alpha * mh / (np.sqrt(vh) + eps)
    return p, m, v, int(n0) + p.size
