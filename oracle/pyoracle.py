"""ctypes bindings for the CHECKERS: oracle/libgf_oracle.so (our fp64 restatement) and, when present,
oracle/_ref/libgf_ref.so (the real reference, built from /root/reference by oracle/Makefile).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (graphflow_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i = C.c_int


def build(force=False):
    """Compile the C restatement (and the reference shim when /root/reference is mounted)."""
    so = os.path.join(_HERE, "libgf_oracle.so")
    src = os.path.join(_HERE, "gf_oracle.c")
    srcs = [src, os.path.join(_HERE, "smp_port.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libgf_oracle.so"])
    ref_root = os.environ.get("GF_REFERENCE", "/root/reference")
    ref_so = os.path.join(_HERE, "_ref", "libgf_ref.so")
    shim = os.path.join(_HERE, "ref_shim.cpp")
    if os.path.isdir(os.path.join(ref_root, "GraphFlow")):
        if force or not os.path.exists(ref_so) or os.path.getmtime(ref_so) < os.path.getmtime(shim):
            subprocess.check_call(["make", "-C", _HERE, "ref", "GF_REFERENCE=" + ref_root])


def _load(path):
    return C.CDLL(path) if os.path.exists(path) else None


class _Lib:
    """Shared numpy-facing surface of both checkers (prefix 'gfo_' or 'ref_')."""

    def __init__(self, lib, kind):
        self.lib = lib
        self.kind = kind

    # -- contractions -------------------------------------------------------------------------
    def contract_forward(self, K, P, A=None):
        P = np.ascontiguousarray(P, dtype=np.float64)
        N, C_ = P.shape[0], P.shape[3]
        out = np.zeros((N, N, K, C_), dtype=np.float64)
        if K == 4:
            f = getattr(self.lib, self._n("r4_forward"))
            f.argtypes = [_dp, _dp, _i, _i]
            f.restype = None
            f(P, out, N, C_)
            return out
        A = np.ascontiguousarray(A, dtype=np.float64)
        if self.kind == "oracle":
            f = self.lib.gfo_contract_forward
            f.argtypes = [_i, _dp, _dp, _dp, _i, _i]
            f.restype = _i
            assert f(K, P, A, out, N, C_) == 0
        else:
            f = getattr(self.lib, "ref_r%d_forward" % K)
            f.argtypes = [_dp, _dp, _dp, _i, _i]
            f.restype = None
            f(P, A, out, N, C_)
        return out

    def contract_backward(self, K, G, A=None, dP0=None):
        """Returns dP0 + vjp (the reference accumulates into the inputs' gradient)."""
        G = np.ascontiguousarray(G, dtype=np.float64)
        N, C_ = G.shape[0], G.shape[3]
        dP = np.zeros((N, N, N, C_), dtype=np.float64) if dP0 is None else np.array(dP0, dtype=np.float64, order="C")
        if K == 4:
            f = getattr(self.lib, self._n("r4_backward"))
            f.argtypes = [_dp, _dp, _i, _i]
            f.restype = None
            f(G, dP, N, C_)
            return dP
        A = np.ascontiguousarray(A, dtype=np.float64)
        if self.kind == "oracle":
            f = self.lib.gfo_contract_backward
            f.argtypes = [_i, _dp, _dp, _dp, _i, _i]
            f.restype = _i
            assert f(K, G, A, dP, N, C_) == 0
        else:
            f = getattr(self.lib, "ref_r%d_backward" % K)
            f.argtypes = [_dp, _dp, _dp, _i, _i]
            f.restype = None
            f(G, A, dP, N, C_)
        return dP

    def r18_thread_forward(self, P, A):
        P = np.ascontiguousarray(P, dtype=np.float64)
        A = np.ascontiguousarray(A, dtype=np.float64)
        N, C_ = P.shape[0], P.shape[3]
        out = np.zeros((N, N, 18, C_), dtype=np.float64)
        f = getattr(self.lib, self._n("r18_thread_forward"))
        f.argtypes = [_dp, _dp, _dp, _i, _i]
        f.restype = None
        f(P, A, out, N, C_)
        return out

    # -- mixers ---------------------------------------------------------------------------------
    def matmul_forward(self, A, B):
        A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        M, K = A.shape
        N = B.shape[1]
        out = np.zeros((M, N))
        f = getattr(self.lib, self._n("matmul_forward"))
        f.argtypes = [_dp, _dp, _dp, _i, _i, _i]
        f.restype = None
        f(A, B, out, M, K, N)
        return out

    def matmul_backward(self, dC, A, B, dA0=None, dB0=None):
        A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        dC = np.ascontiguousarray(dC, dtype=np.float64)
        M, K = A.shape
        N = B.shape[1]
        dA = np.zeros_like(A) if dA0 is None else np.array(dA0, dtype=np.float64, order="C")
        dB = np.zeros_like(B) if dB0 is None else np.array(dB0, dtype=np.float64, order="C")
        f = getattr(self.lib, self._n("matmul_backward"))
        f.argtypes = [_dp, _dp, _dp, _dp, _dp, _i, _i, _i]
        f.restype = None
        f(dC, A, B, dA, dB, M, K, N)
        return dA, dB

    def _tm(self, name, first, second, out_shape, dims):
        out = np.zeros(out_shape)
        f = getattr(self.lib, self._n(name))
        f.argtypes = [_dp, _dp, _dp, _i, _i, _i, _i]
        f.restype = None
        f(np.ascontiguousarray(first, dtype=np.float64), np.ascontiguousarray(second, dtype=np.float64), out, *dims)
        return out

    def mattensormul_forward(self, X, F):
        R, Kd = X.shape
        _, J, D = F.shape
        return self._tm("mattensormul_forward", X, F, (R, J, D), (R, Kd, J, D))

    def tensormatmul_forward(self, F, Y):
        R, Kd, D = F.shape
        J = Y.shape[1]
        return self._tm("tensormatmul_forward", F, Y, (R, J, D), (R, Kd, J, D))

    def _tmb(self, name, G, first, second, dims):
        d1 = np.zeros(np.shape(first))
        d2 = np.zeros(np.shape(second))
        f = getattr(self.lib, self._n(name))
        f.argtypes = [_dp, _dp, _dp, _dp, _dp, _i, _i, _i, _i]
        f.restype = None
        f(np.ascontiguousarray(G, dtype=np.float64), np.ascontiguousarray(first, dtype=np.float64),
          np.ascontiguousarray(second, dtype=np.float64), d1, d2, *dims)
        return d1, d2

    def mattensormul_backward(self, G, X, F):
        R, Kd = X.shape
        _, J, D = F.shape
        return self._tmb("mattensormul_backward", G, X, F, (R, Kd, J, D))  # (dX, dF)

    def tensormatmul_backward(self, G, F, Y):
        R, Kd, D = F.shape
        J = Y.shape[1]
        return self._tmb("tensormatmul_backward", G, F, Y, (R, Kd, J, D))  # (dF, dY)

    def r18_dropout(self, use, train, nKept, P, A, G=None, dP0=None, seed=None):
        """oracle: mask `use` given; reference: mask drawn by the reference after srand(seed) and returned."""
        P = np.ascontiguousarray(P, dtype=np.float64)
        A = np.ascontiguousarray(A, dtype=np.float64)
        N, Cc = P.shape[0], P.shape[3]
        out = np.zeros((N, N, 18, Cc))
        dP = None
        ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        if G is not None:
            G = np.ascontiguousarray(G, dtype=np.float64)
            dP = np.zeros_like(P) if dP0 is None else np.array(dP0, dtype=np.float64, order="C")
        if self.kind == "oracle":
            u = np.ascontiguousarray(use, dtype=np.int32)
            f = self.lib.gfo_r18_dropout_forward
            f.argtypes = [ip, _i, _i, _dp, _dp, _dp, _i, _i]
            f.restype = _i
            assert f(u, 1 if train else 0, nKept, P, A, out, N, Cc) == 0
            if G is not None:
                b = self.lib.gfo_r18_dropout_backward
                b.argtypes = [ip, _dp, _dp, _dp, _i, _i]
                b.restype = _i
                assert b(u, G, A, dP, N, Cc) == 0
            return out, dP, u
        u = np.zeros(18, dtype=np.int32)
        f = self.lib.ref_r18_dropout
        vp = C.c_void_p
        f.argtypes = [C.c_uint, _i, _i, _dp, _dp, vp, _dp, vp, ip, _i, _i]
        f.restype = None
        f(int(seed), nKept, 1 if train else 0, P, A, G.ctypes.data if G is not None else None, out,
          dP.ctypes.data if dP is not None else None, u, N, Cc)
        return out, dP, u

    def custommatmultensor_forward(self, W, T):
        W = np.ascontiguousarray(W, dtype=np.float64)
        T = np.ascontiguousarray(T, dtype=np.float64)
        I, J, V = T.shape
        Kout = W.shape[0]
        out = np.zeros((I, J, Kout))
        f = getattr(self.lib, self._n("custommatmultensor_forward"))
        if self.kind == "oracle":
            f.argtypes = [_dp, _dp, _dp, _i, _i, _i]
            f.restype = None
            f(W, T, out, I * J, V, Kout)
        else:
            f.argtypes = [_dp, _dp, _dp, _i, _i, _i, _i]
            f.restype = None
            f(W, T, out, I, J, V, Kout)
        return out

    def custommatmultensor_backward(self, G, W, T, dW0=None, dT0=None):
        W = np.ascontiguousarray(W, dtype=np.float64)
        T = np.ascontiguousarray(T, dtype=np.float64)
        G = np.ascontiguousarray(G, dtype=np.float64)
        I, J, V = T.shape
        Kout = W.shape[0]
        dW = np.zeros_like(W) if dW0 is None else np.array(dW0, dtype=np.float64, order="C")
        dT = np.zeros_like(T) if dT0 is None else np.array(dT0, dtype=np.float64, order="C")
        f = getattr(self.lib, self._n("custommatmultensor_backward"))
        if self.kind == "oracle":
            f.argtypes = [_dp] * 5 + [_i, _i, _i]
            f.restype = None
            f(G, W, T, dW, dT, I * J, V, Kout)
        else:
            f.argtypes = [_dp] * 5 + [_i, _i, _i, _i]
            f.restype = None
            f(G, W, T, dW, dT, I, J, V, Kout)
        return dW, dT

    # -- StackTensor3D (StackTensor3D.h:54-90) -------------------------------------------------------
    def stack_forward(self, T):
        """T [nRows][nCols][n1][n2] = the nRows source tensors; returns the stacked Tensor4D's value."""
        T = np.ascontiguousarray(T, dtype=np.float64)
        nRows, nCols, n1, n2 = T.shape
        out = np.zeros_like(T)
        if self.kind == "oracle":
            f = self.lib.gfo_stack_forward
            f.argtypes = [C.c_void_p, _dp, _i, C.c_size_t]
            f.restype = None
            ptrs = (C.c_void_p * nRows)(*[T[r].ctypes.data for r in range(nRows)])
            f(C.cast(ptrs, C.c_void_p), out, nRows, nCols * n1 * n2)
        else:
            f = self.lib.ref_stack_forward
            f.argtypes = [_dp, _dp, _i, _i, _i, _i]
            f.restype = None
            f(T, out, nRows, nCols, n1, n2)
        return out

    def stack_backward(self, G, dT0=None):
        """Returns dT0 + the scatter of the stacked gradient G [nRows][nCols][n1][n2] (the `+=` of StackTensor3D::backward)."""
        G = np.ascontiguousarray(G, dtype=np.float64)
        nRows, nCols, n1, n2 = G.shape
        dT = np.zeros_like(G) if dT0 is None else np.array(dT0, dtype=np.float64, order="C")
        if self.kind == "oracle":
            f = self.lib.gfo_stack_backward
            f.argtypes = [_dp, C.c_void_p, _i, C.c_size_t]
            f.restype = None
            ptrs = (C.c_void_p * nRows)(*[dT[r].ctypes.data for r in range(nRows)])
            f(G, C.cast(ptrs, C.c_void_p), nRows, nCols * n1 * n2)
        else:
            f = self.lib.ref_stack_backward
            f.argtypes = [_dp, _dp, _i, _i, _i, _i]
            f.restype = None
            f(G, dT, nRows, nCols, n1, n2)
        return dT

    def _n(self, name):
        return ("gfo_" if self.kind == "oracle" else "ref_") + name


def oracle():
    build()
    return _Lib(C.CDLL(os.path.join(_HERE, "libgf_oracle.so")), "oracle")


def reference():
    """The real reference, or None when oracle/_ref/libgf_ref.so has not been built/shipped."""
    lib = _load(os.path.join(_HERE, "_ref", "libgf_ref.so"))
    return _Lib(lib, "reference") if lib is not None else None


# -- CPU-baseline timing entry points (bench.py) -----------------------------------------------------
def time_r18_fwd_bwd(P, A, G, prefer_reference=True):
    """Seconds for ONE RisiContraction_18 forward+backward on one graph, single host thread.
    Returns (seconds, kind) with kind 'reference' (real GraphFlow code) or 'port' (gf_oracle.c loop nests)."""
    import time

    P = np.ascontiguousarray(P, dtype=np.float64)
    A = np.ascontiguousarray(A, dtype=np.float64)
    G = np.ascontiguousarray(G, dtype=np.float64)
    N, C_ = P.shape[0], P.shape[3]
    out = np.zeros((N, N, 18, C_))
    dP = np.zeros_like(P)
    ref = reference() if prefer_reference else None
    if ref is not None:
        f = ref.lib.ref_r18_fwd_bwd
        f.argtypes = [_dp, _dp, _dp, _dp, _dp, _i, _i]
        f.restype = None
        t0 = time.perf_counter()
        f(P, A, G, out, dP, N, C_)
        return time.perf_counter() - t0, "reference", out, dP
    orc = oracle()
    fw = orc.lib.gfo_r18_loops_forward
    fw.argtypes = [_dp, _dp, _dp, _i, _i]
    fw.restype = None
    bw = orc.lib.gfo_r18_loops_backward
    bw.argtypes = [_dp, _dp, _dp, _i, _i]
    bw.restype = None
    t0 = time.perf_counter()
    fw(P, A, out, N, C_)
    bw(G, A, dP, N, C_)
    return time.perf_counter() - t0, "port", out, dP


def time_r50_fwd_bwd(P, A, G):
    """Seconds for ONE RisiContraction_50 forward + backward on one graph, single host thread, through the loop-nest port
    (gfo_r50_loops_*: the reference's one-nest-per-channel structure with all 50 predicated updates inside).  Returns (s, out, dP)."""
    import time

    P = np.ascontiguousarray(P, dtype=np.float64)
    A = np.ascontiguousarray(A, dtype=np.float64)
    G = np.ascontiguousarray(G, dtype=np.float64)
    N, C_ = P.shape[0], P.shape[3]
    out = np.zeros((N, N, 50, C_))
    dP = np.zeros_like(P)
    orc = oracle()
    fw, bw = orc.lib.gfo_r50_loops_forward, orc.lib.gfo_r50_loops_backward
    fw.argtypes = bw.argtypes = [_dp, _dp, _dp, _i, _i]
    fw.restype = bw.restype = None
    t0 = time.perf_counter()
    fw(P, A, out, N, C_)
    bw(G, A, dP, N, C_)
    return time.perf_counter() - t0, out, dP


def reference_smp_beta(adj, feature, target, params, nLevels, nChanels, nDepth, has_wl=True, max_nVertices=None):
    """The REAL SMP_beta (no receptive-field cap) on one molecule with dumped parameters."""
    ref = reference()
    if ref is None:
        return None
    adj = np.ascontiguousarray(adj, dtype=np.int32)
    feature = np.ascontiguousarray(feature, dtype=np.float64)
    params = np.ascontiguousarray(params, dtype=np.float64)
    V, F = feature.shape
    gfeat, pred, loss, grads = np.zeros(nChanels), np.zeros(1), np.zeros(1), np.zeros_like(params)
    f = ref.lib.ref_smp_beta_run
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    f.argtypes = [_i] * 7 + [ip, _dp, C_double, _dp, _dp, _dp, _dp, _dp]
    f.restype = _i
    n = f(max_nVertices or V, nLevels, nChanels, F, nDepth, 1 if has_wl else 0, V, adj, feature, float(target), params, gfeat,
          pred, loss, grads)
    assert n == params.size
    return {"graph_feature": gfeat, "predict": float(pred[0]), "loss": float(loss[0]), "grads": grads}


def reference_smp_2d(version, adj, feature, target, params, nLevels, nChanels, nDepth, has_wl=True, max_nVertices=None):
    """The REAL SMP_2D_ver6 / ver7 / ver8 on one molecule with dumped parameters (contraction _10 / _50 / _18)."""
    ref = reference()
    if ref is None:
        return None
    adj = np.ascontiguousarray(adj, dtype=np.int32)
    feature = np.ascontiguousarray(feature, dtype=np.float64)
    params = np.ascontiguousarray(params, dtype=np.float64)
    V, F = feature.shape
    maxV = max_nVertices or V
    gfeat, pred, loss, grads = np.zeros(nChanels), np.zeros(1), np.zeros(1), np.zeros_like(params)
    phi = np.zeros((nLevels + 1, V, maxV + 1), dtype=np.int32)
    f = ref.lib.ref_smp_2d_run
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    f.argtypes = [_i] * 8 + [ip, _dp, C_double, _dp, _dp, _dp, _dp, _dp, ip, _i]
    f.restype = _i
    n = f(version, maxV, nLevels, nChanels, F, nDepth, 1 if has_wl else 0, V, adj, feature, float(target), params, gfeat, pred,
          loss, grads, phi, maxV + 1)
    assert n == params.size, (n, params.size)
    fields = [[list(phi[l, v, 1:1 + phi[l, v, 0]]) for v in range(V)] for l in range(nLevels + 1)]
    return {"phi": fields, "phi_array": phi, "graph_feature": gfeat, "predict": float(pred[0]), "loss": float(loss[0]), "grads": grads}


def reference_smp_2d_batchlearn(version, mols, targets, nLevels, nChanels, nDepth, max_nVertices, momentum, nIter, learning_rate, seed):
    """nIter x the REAL SMP_2D_ver6/7/8::BatchLearn (Momentum) from the constructor's own srand(seed) weights."""
    ref = reference()
    if ref is None:
        return None
    nK = {6: 10, 7: 50, 8: 18}[version]
    nV = np.array([len(a) for a, _ in mols], dtype=np.int32)
    F = mols[0][1].shape[1]
    adj = np.concatenate([np.asarray(a, dtype=np.int32).ravel() for a, _ in mols])
    feat = np.concatenate([np.asarray(f, dtype=np.float64).ravel() for _, f in mols])
    tg = np.ascontiguousarray(targets, dtype=np.float64)
    n = nChanels * F * (nDepth + 1) + nLevels * (nK * nChanels * nChanels + nChanels) + nChanels
    p0, p1, losses = np.zeros(n), np.zeros(n), np.zeros((nIter, 2))
    f = ref.lib.ref_smp_2d_batchlearn
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    f.argtypes = [_i] * 6 + [C_double, _i, ip, ip, _dp, _dp, _i, _dp, _i, C_double, _dp, _dp]
    f.restype = _i
    got = f(version, max_nVertices, nLevels, nChanels, F, nDepth, float(momentum), len(mols), nV, adj, feat, tg, int(seed), p0,
            nIter, float(learning_rate), losses, p1)
    assert got == n, (got, n)
    return {"params0": p0, "params": p1, "losses": losses}


def reference_batchlearn(mols, targets, nLevels, nChanels, nDepth, cap, max_nVertices, nIter, learning_rate, params=None, seed=-1):
    """nIter x the REAL SMP_omega::BatchLearn(nBatch, molecules, targets, learning_rate).  params given, or drawn by the
    reference's own constructor after srand(seed).  Returns dict(params0, params, losses[nIter, 2])."""
    ref = reference()
    if ref is None:
        return None
    nV = np.array([len(a) for a, _ in mols], dtype=np.int32)
    F = mols[0][1].shape[1]
    adj = np.concatenate([np.asarray(a, dtype=np.int32).ravel() for a, _ in mols])
    feat = np.concatenate([np.asarray(f, dtype=np.float64).ravel() for _, f in mols])
    tg = np.ascontiguousarray(targets, dtype=np.float64)
    n = smp_param_count_(nChanels, F, nDepth, nLevels)
    p_in = np.zeros(n) if params is None else np.ascontiguousarray(params, dtype=np.float64)
    p0, p1, losses = np.zeros(n), np.zeros(n), np.zeros((nIter, 2))
    f = ref.lib.ref_smp_omega_batchlearn
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    f.argtypes = [_i] * 7 + [ip, ip, _dp, _dp, _dp, _i, _dp, _i, C_double, _dp, _dp]
    f.restype = _i
    got = f(max_nVertices, cap, nLevels, nChanels, F, nDepth, len(mols), nV, adj, feat, tg, p_in, int(seed), p0, nIter,
            float(learning_rate), losses, p1)
    assert got == n, (got, n)
    return {"params0": p0, "params": p1, "losses": losses}


def smp_param_count_(C_, F, D, L):
    return C_ * F * (D + 1) + L * (18 * C_ * C_ + C_) + C_


def reference_save_model(path, params, nLevels, nChanels, nFeatures, nDepth, cap, max_nVertices):
    """Text checkpoint written by the REAL reference's SMP_omega::save_model for the given parameter values."""
    ref = reference()
    if ref is None:
        return None
    params = np.ascontiguousarray(params, dtype=np.float64)
    f = ref.lib.ref_smp_omega_save_model
    f.argtypes = [_i] * 6 + [_dp, C.c_char_p]
    f.restype = _i
    n = f(max_nVertices, cap, nLevels, nChanels, nFeatures, nDepth, params, path.encode())
    assert n == params.size, (n, params.size)
    return path


def reference_smp_omega(adj, feature, target, params, nLevels, C, nDepth, cap, has_wl=True, max_nVertices=None, coulomb=None,
                        want_activations=False):
    """Run the REAL reference SMP_omega on one molecule with the given (dumped) parameters.  None if _ref is absent.
    want_activations: also return "f" = [[level[l]->f[v]->value as [s, s, C]]] for every level and vertex."""
    ref = reference()
    if ref is None:
        return None
    adj = np.ascontiguousarray(adj, dtype=np.int32)
    feature = np.ascontiguousarray(feature, dtype=np.float64)
    params = np.ascontiguousarray(params, dtype=np.float64)
    V, F = feature.shape
    maxV = max_nVertices or V
    L = nLevels
    gfeat = np.zeros(C)
    pred = np.zeros(1)
    loss = np.zeros(1)
    grads = np.zeros_like(params)
    phi = np.zeros((L + 1, V, cap + 1), dtype=np.int32)
    radj = np.zeros((L + 1, V, cap * cap))
    f = ref.lib.ref_smp_omega_run_acts
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    f.argtypes = [_i] * 8 + [ip, _dp, C_double, _dp, _dp, _dp, _dp, _dp, ip, _dp, ctypes_vp, ctypes_vp]
    f.restype = _i
    cm = None if coulomb is None else np.ascontiguousarray(coulomb, dtype=np.float64)
    acts = np.zeros((L + 1) * V * cap * cap * C) if want_activations else None   # upper bound; trimmed below
    n = f(maxV, cap, L, C, F, nDepth, 1 if has_wl else 0, V, adj, feature, float(target), params, gfeat, pred, loss, grads, phi, radj,
          None if cm is None else cm.ctypes.data, None if acts is None else acts.ctypes.data)
    assert n == params.size, (n, params.size)
    fields = [[list(phi[l, v, 1:1 + phi[l, v, 0]]) for v in range(V)] for l in range(L + 1)]
    fl = None
    if want_activations:
        fl, o = [[None] * V for _ in range(L + 1)], 0
        for l in range(L + 1):
            for v in range(V):
                s = int(phi[l, v, 0])
                fl[l][v] = acts[o:o + s * s * C].reshape(s, s, C).copy()
                o += s * s * C
    red = [[None] * V for _ in range(L + 1)]
    for l in range(1, L + 1):
        for v in range(V):
            s = phi[l, v, 0]
            red[l][v] = radj[l, v, :s * s].reshape(s, s).copy()
    return {"phi": fields, "reduced_adj": red, "graph_feature": gfeat, "predict": float(pred[0]), "loss": float(loss[0]),
            "grads": grads, "f": fl}


C_double = C.c_double
ctypes_vp = C.c_void_p


def time_reference_smp_omega(molecules, targets, nLevels, nChanels, nDepth, cap):
    """Seconds the REAL reference spends on complete_computation_graph + forward + backward over `molecules`
    (list of (adj, feature)), single host thread, or None when oracle/_ref is absent."""
    ref = reference()
    if ref is None:
        return None
    nV = np.array([len(a) for a, _ in molecules], dtype=np.int32)
    adj = np.concatenate([np.ascontiguousarray(a, dtype=np.int32).ravel() for a, _ in molecules])
    feat = np.concatenate([np.ascontiguousarray(f, dtype=np.float64).ravel() for _, f in molecules])
    tg = np.ascontiguousarray(targets, dtype=np.float64)
    F = molecules[0][1].shape[1]
    f = ref.lib.ref_smp_omega_time
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    f.argtypes = [_i] * 7 + [ip, ip, _dp, _dp]
    f.restype = C_double
    return float(f(int(nV.max()), cap, nLevels, nChanels, F, nDepth, len(molecules), nV, adj, feat, tg))


# -- the C port of the SMP step (oracle/smp_port.c): second restatement + the "port"-kind CPU baseline ---------------------
def _port_inputs(molecules, L, nDepth, cap, has_wl=True):
    """Graph preparation by the numpy restatement (oracle/smp_oracle.py), packed for gfo_smp_molecule / gfo_smp_batch."""
    from . import smp_oracle
    xs, phis, adjs, nV = [], [], [], []
    for adj, feat in molecules:
        adj = np.asarray(adj)
        feat = np.asarray(feat, dtype=np.float64)
        V = len(adj)
        sp = smp_oracle.hop_distances(adj.tolist())
        x = smp_oracle.wl_features(feat, sp, nDepth)
        rank = smp_oracle.rank_vertices(x.tolist())
        fields = smp_oracle.receptive_fields(sp, rank, L, cap, has_wl)
        phi = np.full((L + 1, V, cap + 1), -1, dtype=np.int32)
        for l in range(L + 1):
            for v in range(V):
                phi[l, v, 0] = len(fields[l][v])
                phi[l, v, 1:1 + len(fields[l][v])] = fields[l][v]
        xs.append(np.ascontiguousarray(x))
        phis.append(phi)
        adjs.append(np.ascontiguousarray(adj, dtype=np.int32))
        nV.append(V)
    return xs, phis, adjs, nV


def port_smp_molecule(adj, feature, target, params, L, C, nDepth, cap, has_wl=True, coulomb=None, activation=None,
                      ext_sign=None, kink_tol=0.0, want_acts=False):
    """One molecule through the C port.  activation=(level, vertex): also return that f[l][v] as "f".
    ext_sign: list [l][v] of arrays [s, s, C] (any dtype; > 0 means the implementation under test saw a positive activation):
    inside the LeakyReLU kink tolerance (|z| <= kink_tol * max|z| of the level) the reverse sweep takes that slope; the
    result then carries "n_override" / "n_conflict" (see gfo_smp_molecule_ex).  want_acts: "acts" = [[f[l][v]]]."""
    lib = oracle().lib
    xs, phis, adjs, nV = _port_inputs([(adj, feature)], L, nDepth, cap, has_wl)
    x, phi, a, V = xs[0], phis[0], adjs[0], nV[0]
    FD = x.shape[1]
    params = np.ascontiguousarray(params, dtype=np.float64)
    gfeat, pred, loss, grads = np.zeros(C), np.zeros(1), np.zeros(1), np.zeros_like(params)
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    f = lib.gfo_smp_molecule_ex
    f.argtypes = [_i] * 5 + [_dp, ip, ip, ctypes_vp, _dp, C_double, _i, _dp, _dp, _dp, _dp, _i, _i, ctypes_vp,
                  ctypes_vp, C_double, ctypes_vp, ctypes_vp, ctypes_vp]
    f.restype = _i
    cm = None if coulomb is None else np.ascontiguousarray(coulomb, dtype=np.float64)
    act = None
    al, av = (-1, -1)
    if activation is not None:
        al, av = activation
        s = int(phi[al, av, 0])
        act = np.zeros((s, s, C))
    sizes = [[int(phi[l, v, 0]) for v in range(V)] for l in range(L + 1)]
    total = sum(s * s * C for row in sizes for s in row)
    sign = None
    if ext_sign is not None:
        sign = np.concatenate([np.where(np.asarray(ext_sign[l][v]) > 0, 1, -1).astype(np.int8).ravel() for l in range(L + 1) for v in range(V)])
        assert sign.size == total
    acts = np.zeros(total) if want_acts else None
    import ctypes as ct
    n_over, n_conf = ct.c_longlong(0), ct.c_longlong(0)
    rc = f(V, FD, L, C, cap, x, phi, a, None if cm is None else cm.ctypes.data, params, float(target), 1, gfeat, pred, loss, grads,
           al, av, None if act is None else act.ctypes.data, None if sign is None else sign.ctypes.data, float(kink_tol),
           ct.addressof(n_over), ct.addressof(n_conf), None if acts is None else acts.ctypes.data)
    assert rc == 0
    out = {"graph_feature": gfeat, "predict": float(pred[0]), "loss": float(loss[0]), "grads": grads, "f": act,
           "n_override": int(n_over.value), "n_conflict": int(n_conf.value), "n_elements": total}
    if want_acts:
        out["acts"], o = [[None] * V for _ in range(L + 1)], 0
        for l in range(L + 1):
            for v in range(V):
                s = sizes[l][v]
                out["acts"][l][v] = acts[o:o + s * s * C].reshape(s, s, C)
                o += s * s * C
    return out


def port_smp_batch(molecules, targets, params, L, C, nDepth, cap, nThreads=1):
    """(seconds, predict, loss, grads): the batch through gfo_smp_batch -- waves of nThreads molecules on host threads, the
    shape of SMP_omega::Threaded_BatchLearn (nThreads = 1: the serial gradient loop of BatchLearn).  The clock covers the
    op DAG (forward + backward of every molecule) only, not the numpy graph preparation."""
    import time
    lib = oracle().lib
    xs, phis, adjs, nV = _port_inputs(molecules, L, nDepth, cap)
    FD = xs[0].shape[1]
    lp = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    x_off = np.cumsum([0] + [x.size for x in xs[:-1]]).astype(np.int64)
    p_off = np.cumsum([0] + [p.size for p in phis[:-1]]).astype(np.int64)
    a_off = np.cumsum([0] + [a.size for a in adjs[:-1]]).astype(np.int64)
    x = np.concatenate([x.ravel() for x in xs])
    phi = np.concatenate([p.ravel() for p in phis])
    adj = np.concatenate([a.ravel() for a in adjs])
    params = np.ascontiguousarray(params, dtype=np.float64)
    tg = np.ascontiguousarray(targets, dtype=np.float64)
    n = len(molecules)
    pred, loss, grads = np.zeros(n), np.zeros(n), np.zeros_like(params)
    f = lib.gfo_smp_batch
    f.argtypes = [_i, ip, _i, _i, _i, _i, _dp, lp, ip, lp, ip, lp, _dp, _dp, _i, _dp, _dp, _dp]
    f.restype = _i
    t0 = time.perf_counter()
    rc = f(n, np.array(nV, dtype=np.int32), FD, L, C, cap, x, x_off, phi, p_off, adj, a_off, params, tg, int(nThreads), pred, loss, grads)
    secs = time.perf_counter() - t0
    assert rc == 0
    return secs, pred, loss, grads


def time_reference_smp_omega_threaded(molecules, targets, nLevels, nChanels, nDepth, cap, nThreads):
    """Seconds of the REAL SMP_omega::Threaded_BatchLearn over `molecules` with init_multi_threads(nThreads); None without _ref."""
    ref = reference()
    if ref is None:
        return None
    nV = np.array([len(a) for a, _ in molecules], dtype=np.int32)
    adj = np.concatenate([np.ascontiguousarray(a, dtype=np.int32).ravel() for a, _ in molecules])
    feat = np.concatenate([np.ascontiguousarray(f, dtype=np.float64).ravel() for _, f in molecules])
    tg = np.ascontiguousarray(targets, dtype=np.float64)
    F = molecules[0][1].shape[1]
    f = ref.lib.ref_smp_omega_threaded_time
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    f.argtypes = [_i] * 8 + [ip, ip, _dp, _dp]
    f.restype = C_double
    return float(f(int(nThreads), int(nV.max()), cap, nLevels, nChanels, F, nDepth, len(molecules), nV, adj, feat, tg))


def port_r18_batch_threads(P, A, G, nGraphs, nThreads):
    """Seconds for nGraphs x (RisiContraction_18 forward + backward, port loop nests) spread over nThreads host threads."""
    import time
    lib = oracle().lib
    P = np.ascontiguousarray(P, dtype=np.float64)
    A = np.ascontiguousarray(A, dtype=np.float64)
    G = np.ascontiguousarray(G, dtype=np.float64)
    f = lib.gfo_r18_batch_threads
    f.argtypes = [_dp, _dp, _dp, _i, _i, _i, _i]
    f.restype = _i
    t0 = time.perf_counter()
    assert f(P, A, G, P.shape[0], P.shape[3], int(nGraphs), int(nThreads)) == 0
    return time.perf_counter() - t0


# -- the `_physics` / `_pairgraphs` drivers of the real reference (goldens for SURVEY 8 f3) ------------------------------------
def physics_channels(C_, L):
    return [max(1, C_ >> l) for l in range(L + 1)]


def physics_tower_param_count(C_, F, L):
    ch = physics_channels(C_, L)
    return C_ * F + sum(18 * ch[l - 1] * ch[l] + ch[l] for l in range(1, L + 1))


def reference_smp_physics(adj, feature, target, params, nLevels, C_, cap, beta=False, max_nVertices=None):
    """The REAL SMP_omega_physics (or SMP_beta_physics) on one molecule, parameters in its registration order
    H, (K_l, b_l)..., W1, W2."""
    ref = reference()
    if ref is None:
        return None
    adj = np.ascontiguousarray(adj, dtype=np.int32)
    feature = np.ascontiguousarray(feature, dtype=np.float64)
    params = np.ascontiguousarray(params, dtype=np.float64)
    V, F = feature.shape
    maxV = max_nVertices or V
    stride = maxV + 1
    width = sum(physics_channels(C_, nLevels))
    pred, loss, grads, gfeat = np.zeros(1), np.zeros(1), np.zeros_like(params), np.zeros(width)
    phi = np.zeros((nLevels + 1, V, stride), dtype=np.int32)
    f = ref.lib.ref_smp_physics_run
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    f.argtypes = [_i] * 7 + [ip, _dp, C_double, _dp, _dp, _dp, _dp, ip, _i, _dp]
    f.restype = _i
    n = f(1 if beta else 0, maxV, cap, nLevels, C_, F, V, adj, feature, float(target), params, pred, loss, grads, phi, stride, gfeat)
    assert n == params.size, (n, params.size)
    return {"phi": phi, "predict": float(pred[0]), "loss": float(loss[0]), "grads": grads, "graph_feature": gfeat}


def reference_smp_pairgraphs(kind, g1, g2, target, params, nLevels, C_, cap, nKept=18, train=True, seed=0, maxV=None):
    """The REAL SMP_omega_pairgraphs (kind 0) / SMP_beta_pairgraphs (1) / SMP_sigma_pairgraphs (2) on one pair of graphs
    g = (adj, feature); parameters in registration order H_1, H_2, (K1_l, b1_l, K2_l, b2_l)..., W1, W2, W3."""
    ref = reference()
    if ref is None:
        return None
    a1, f1 = np.ascontiguousarray(g1[0], dtype=np.int32), np.ascontiguousarray(g1[1], dtype=np.float64)
    a2, f2 = np.ascontiguousarray(g2[0], dtype=np.int32), np.ascontiguousarray(g2[1], dtype=np.float64)
    params = np.ascontiguousarray(params, dtype=np.float64)
    V1, V2 = len(a1), len(a2)
    m = maxV or max(V1, V2)
    stride = m + 1
    width = 2 * sum(physics_channels(C_, nLevels))
    pred, loss, grads, gfeat = np.zeros(1), np.zeros(1), np.zeros_like(params), np.zeros(width)
    phi1 = np.zeros((nLevels + 1, V1, stride), dtype=np.int32)
    phi2 = np.zeros((nLevels + 1, V2, stride), dtype=np.int32)
    f = ref.lib.ref_smp_pairgraphs_run
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    f.argtypes = [_i] * 9 + [ip, _dp, _i, ip, _dp, C_double, _dp, _i, _i, _i, _dp, _dp, _dp, ip, ip, _i, _dp]
    f.restype = _i
    n = f(kind, m, m, cap, nLevels, C_, f1.shape[1], f2.shape[1], V1, a1, f1, V2, a2, f2, float(target), params, int(nKept), 1 if train else 0,
          int(seed), pred, loss, grads, phi1, phi2, stride, gfeat)
    assert n == params.size, (n, params.size)
    return {"phi1": phi1, "phi2": phi2, "predict": float(pred[0]), "loss": float(loss[0]), "grads": grads, "graph_feature": gfeat}


def reference_model_batchlearn(kind, graphs1, graphs2, targets, nLevels, C_, cap, maxV, nIter, learning_rate, seed, n_params):
    """nIter x the REAL BatchLearn of SMP_omega_physics (kind 0) / SMP_beta_physics (1) / SMP_omega_pairgraphs (10) /
    SMP_beta_pairgraphs (11), weights drawn by the constructor after srand(seed)."""
    ref = reference()
    if ref is None:
        return None

    def pack(gs):
        nV = np.array([len(a) for a, _ in gs], dtype=np.int32)
        adj = np.concatenate([np.asarray(a, dtype=np.int32).ravel() for a, _ in gs])
        feat = np.concatenate([np.asarray(f, dtype=np.float64).ravel() for _, f in gs])
        return nV, adj, feat
    a = pack(graphs1)
    b = pack(graphs2 if graphs2 is not None else graphs1)
    tg = np.ascontiguousarray(targets, dtype=np.float64)
    p0, p1, losses = np.zeros(n_params), np.zeros(n_params), np.zeros((nIter, 2))
    f = ref.lib.ref_smp_model_batchlearn
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    f.argtypes = [_i] * 8 + [ip, ip, _dp, ip, ip, _dp, _dp, _i, _i, C_double, _dp, _dp, _dp]
    f.restype = _i
    n = f(kind, maxV, cap, nLevels, C_, graphs1[0][1].shape[1], (graphs2 or graphs1)[0][1].shape[1], len(graphs1), a[0], a[1], a[2], b[0], b[1],
          b[2], tg, int(seed), nIter, float(learning_rate), p0, losses, p1)
    assert n == n_params, (n, n_params)
    return {"params0": p0, "params": p1, "losses": losses}
