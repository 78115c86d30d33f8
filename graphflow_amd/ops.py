"""Device-pointer ("mode B") entry points of the C ABI, taking torch CUDA tensors for storage only."""
import ctypes as C

import torch

from . import _lib


class GraphFlowHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("gf_hip status %d: %s" % (status, message))
        self.status = status


class Context:
    """gf_ctx wrapper.  By default it runs on torch's current stream of `device`, so torch ops and gf kernels
    order naturally; pass own_stream=True for a private non-blocking stream (gf_ctx_use_private_stream) -- the caller
    must then order it against torch's stream itself."""

    def __init__(self, device=0, own_stream=False):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise GraphFlowHipError(_lib.GF_ERR_HIP, "no HIP device visible to torch; graphflow_amd has no CPU fallback")
        self.device = torch.device("cuda", device)
        handle = C.c_void_p()
        # torch's default stream has handle 0 == HIP's null stream, which is also gf_ctx_create's default
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        st = self.lib.gf_ctx_create(C.byref(handle), device, stream)
        if st != _lib.GF_OK:
            raise GraphFlowHipError(st, self.lib.gf_last_error(None).decode())
        self.handle = handle
        if own_stream:
            self.check(self.lib.gf_ctx_use_private_stream(handle))

    def check(self, st):
        if st != _lib.GF_OK:
            raise GraphFlowHipError(st, self.lib.gf_last_error(self.handle).decode())

    def use_torch_stream(self, stream=None):
        stream = stream or torch.cuda.current_stream(self.device)
        self.check(self.lib.gf_ctx_set_stream(self.handle, C.c_void_p(stream.cuda_stream)))

    @property
    def stream_handle(self):
        return self.lib.gf_ctx_get_stream(self.handle)

    def synchronize(self):
        self.check(self.lib.gf_ctx_synchronize(self.handle))

    def reserve(self, nbytes):
        self.check(self.lib.gf_ctx_reserve(self.handle, int(nbytes)))

    def hbm_copy_probe(self, dst, src, mode=0, iters=5):
        """gf_hbm_copy_probe_f32: GB/s (read + written bytes) of a hand-written float4 copy src -> dst (mode 1: non-temporal)."""
        import ctypes as C_
        assert dst.numel() == src.numel() and src.numel() % 4 == 0
        ms = C_.c_double(0.0)
        self.check(self.lib.gf_hbm_copy_probe_f32(self.handle, C_.c_void_p(dst.data_ptr()), C_.c_void_p(src.data_ptr()), src.numel(), int(mode),
                                                  int(iters), C_.byref(ms)))
        return 2 * 4 * src.numel() / (ms.value * 1e-3) / 1e9

    def set_timing(self, enable=True):
        """Per-kernel HIP-event timing on the context stream (gf_ctx_set_timing)."""
        self.check(self.lib.gf_ctx_set_timing(self.handle, 1 if enable else 0))

    def set_timing_filter(self, kernel_name=None):
        """Only launches named `kernel_name` are timed (None: all) -- see gf_ctx_set_timing_filter."""
        self.check(self.lib.gf_ctx_set_timing_filter(self.handle, kernel_name.encode() if kernel_name else None))

    def set_option(self, option, value):
        """gf_ctx_set_option, e.g. (_lib.GF_OPT_R18_GENERIC_KERNELS, 1)."""
        self.check(self.lib.gf_ctx_set_option(self.handle, int(option), int(value)))

    # -- data parallelism: the context's RCCL communicator (gf_dist_*) ---------------------------------------------
    def dist_unique_id(self):
        """Rank 0: GF_DIST_ID_BYTES bytes to hand to every rank (any host channel) before dist_init."""
        buf = C.create_string_buffer(_lib.GF_DIST_ID_BYTES)
        self.check(self.lib.gf_dist_unique_id(self.handle, buf))
        return buf.raw

    def dist_init(self, unique_id, rank, world):
        if len(unique_id) != _lib.GF_DIST_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % _lib.GF_DIST_ID_BYTES)
        self.check(self.lib.gf_dist_init(self.handle, C.create_string_buffer(bytes(unique_id), _lib.GF_DIST_ID_BYTES), int(rank), int(world)))

    def dist_finalize(self):
        self.check(self.lib.gf_dist_finalize(self.handle))

    @property
    def dist_rank(self):
        return self.lib.gf_dist_rank(self.handle)

    @property
    def dist_world(self):
        return self.lib.gf_dist_world(self.handle)

    def dist_quiesce(self):
        """Bounded wait (GF_DIST_TIMEOUT_S) until every collective issued so far has completed; raises with the rank, the world and
        the exchange when a peer never joined -- call it before a blocking synchronize of a multi-rank run."""
        self.check(self.lib.gf_dist_quiesce(self.handle))

    def allreduce_sum_(self, t):
        """In-place sum over ranks of a contiguous float32 CUDA tensor, ordered on the context's stream."""
        self.check(self.lib.gf_dist_allreduce_sum_f32(self.handle, _dev_f32(t, "tensor"), t.numel()))
        return t

    def broadcast_(self, t, root=0):
        self.check(self.lib.gf_dist_broadcast_f32(self.handle, _dev_f32(t, "tensor"), t.numel(), int(root)))
        return t

    def timings(self):
        """{kernel name: (total_ms, launches)} since timing was enabled; synchronises the stream."""
        out = {}
        for i in range(self.lib.gf_ctx_timing_count(self.handle)):
            name, ms, n = C.c_char_p(), C.c_double(), C.c_longlong()
            self.check(self.lib.gf_ctx_timing_get(self.handle, i, C.byref(name), C.byref(ms), C.byref(n)))
            out[name.value.decode()] = (ms.value, n.value)
        return out

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gf_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = {}


def default_context(device=0):
    if device not in _default:
        _default[device] = Context(device)
    return _default[device]


def _dev_f32(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise TypeError("%s must be a contiguous float32 CUDA tensor" % name)
    return C.c_void_p(t.data_ptr())


def contract_workspace_bytes(K, N, C_, batch):
    return _lib.load().gf_contract_workspace_bytes(K, N, C_, batch)


def contract_forward(P, A, K=18, out=None, ctx=None):
    """P [B,N,N,N,C], A [B,N,N] (ignored for K=4) -> Out [B,N,N,K,C].  RisiContraction_K::forward."""
    B, N, _, _, C_ = P.shape
    ctx = ctx or default_context(P.device.index or 0)
    if out is None:
        out = torch.empty((B, N, N, K, C_), dtype=torch.float32, device=P.device)
    a_ptr = _dev_f32(A, "A") if A is not None else None
    ctx.check(ctx.lib.gf_contract_forward_f32(ctx.handle, K, _dev_f32(P, "P"), a_ptr, _dev_f32(out, "out"), N, C_, B))
    return out


def contract_backward(G, A, K=18, dP=None, accumulate=False, ctx=None):
    """G [B,N,N,K,C] -> dP [B,N,N,N,C] (+= when accumulate).  RisiContraction_K::backward; A gets no gradient."""
    B, N, _, _, C_ = G.shape
    ctx = ctx or default_context(G.device.index or 0)
    if dP is None:
        if accumulate:
            raise ValueError("accumulate=True needs the dP tensor to accumulate into")
        dP = torch.empty((B, N, N, N, C_), dtype=torch.float32, device=G.device)
    a_ptr = _dev_f32(A, "A") if A is not None else None
    ctx.check(ctx.lib.gf_contract_backward_f32(ctx.handle, K, _dev_f32(G, "G"), a_ptr, _dev_f32(dP, "dP"), N, C_, B,
                                               1 if accumulate else 0))
    return dP


def _keep_mask(use):
    m = 0
    for k, u in enumerate(use):
        if u:
            m |= 1 << k
    return m


def contract18_dropout_forward(P, A, use, train=True, nKept=None, out=None, ctx=None):
    """RisiContraction_18_dropout::forward.  use[18]: the kept slices (train mode); in test mode every slice is used and
    the value is scaled by nKept/18 (RisiContraction_18_dropout.h:465-471)."""
    B, N, _, _, C_ = P.shape
    ctx = ctx or default_context(P.device.index or 0)
    if out is None:
        out = torch.empty((B, N, N, 18, C_), dtype=torch.float32, device=P.device)
    mask, scale = (_keep_mask(use), 1.0) if train else (0x3ffff, float(nKept) / 18.0)
    ctx.check(ctx.lib.gf_contract18_dropout_forward_f32(ctx.handle, mask, scale, _dev_f32(P, "P"), _dev_f32(A, "A"),
                                                        _dev_f32(out, "out"), N, C_, B))
    return out


def contract18_dropout_backward(G, A, use, dP=None, accumulate=False, ctx=None):
    """RisiContraction_18_dropout::backward (train mode only, like the reference's assert at :484)."""
    B, N, _, _, C_ = G.shape
    ctx = ctx or default_context(G.device.index or 0)
    if dP is None:
        if accumulate:
            raise ValueError("accumulate=True needs the dP tensor to accumulate into")
        dP = torch.empty((B, N, N, N, C_), dtype=torch.float32, device=G.device)
    ctx.check(ctx.lib.gf_contract18_dropout_backward_f32(ctx.handle, _keep_mask(use), _dev_f32(G, "G"), _dev_f32(A, "A"),
                                                         _dev_f32(dP, "dP"), N, C_, B, 1 if accumulate else 0))
    return dP


def _opt(t, name):
    return _dev_f32(t, name) if t is not None else None


def matmul_forward(A, B, out=None, ctx=None):
    """C[M,N] = A[M,K] B[K,N]  (MatMul::forward)."""
    M, K = A.shape
    N = B.shape[1]
    ctx = ctx or default_context(A.device.index or 0)
    out = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=A.device)
    ctx.check(ctx.lib.gf_matmul_forward_f32(ctx.handle, _dev_f32(A, "A"), _dev_f32(B, "B"), _dev_f32(out, "C"), M, K, N))
    return out


def matmul_backward(dC, A, B, dA=None, dB=None, accumulate=False, ctx=None):
    """dA (+)= dC B^T, dB (+)= A^T dC  (MatMul::backward); pass dA/dB tensors to select what is computed."""
    M, K = A.shape
    N = B.shape[1]
    ctx = ctx or default_context(A.device.index or 0)
    ctx.check(ctx.lib.gf_matmul_backward_f32(ctx.handle, _dev_f32(dC, "dC"), _dev_f32(A, "A"), _dev_f32(B, "B"),
                                             _opt(dA, "dA"), _opt(dB, "dB"), M, K, N, 1 if accumulate else 0))
    return dA, dB


def mattensormul_forward(X, F, ctx=None):
    R, Kd = X.shape
    _, J, D = F.shape
    ctx = ctx or default_context(X.device.index or 0)
    out = torch.empty((R, J, D), dtype=torch.float32, device=X.device)
    ctx.check(ctx.lib.gf_mattensormul_forward_f32(ctx.handle, _dev_f32(X, "X"), _dev_f32(F, "F"), _dev_f32(out, "Out"), R, Kd, J, D))
    return out


def mattensormul_backward(G, X, F, dX=None, dF=None, accumulate=False, ctx=None):
    R, Kd = X.shape
    _, J, D = F.shape
    ctx = ctx or default_context(X.device.index or 0)
    ctx.check(ctx.lib.gf_mattensormul_backward_f32(ctx.handle, _dev_f32(G, "G"), _dev_f32(X, "X"), _dev_f32(F, "F"),
                                                   _opt(dX, "dX"), _opt(dF, "dF"), R, Kd, J, D, 1 if accumulate else 0))
    return dX, dF


def custommatmultensor_forward(W, T, ctx=None):
    """CustomMatMulTensor::forward (CustomMatMulTensor.h:47-68): Out[i,j,k] = sum_v W[k,v] T[i,j,v]."""
    I, J, V = T.shape
    Kout = W.shape[0]
    ctx = ctx or default_context(T.device.index or 0)
    out = torch.empty((I, J, Kout), dtype=torch.float32, device=T.device)
    ctx.check(ctx.lib.gf_custommatmultensor_forward_f32(ctx.handle, _dev_f32(W, "W"), _dev_f32(T, "T"), _dev_f32(out, "Out"),
                                                        I * J, V, Kout))
    return out


def custommatmultensor_backward(G, W, T, dW=None, dT=None, accumulate=False, ctx=None):
    I, J, V = T.shape
    Kout = W.shape[0]
    ctx = ctx or default_context(T.device.index or 0)
    ctx.check(ctx.lib.gf_custommatmultensor_backward_f32(ctx.handle, _dev_f32(G, "G"), _dev_f32(W, "W"), _dev_f32(T, "T"),
                                                         _opt(dW, "dW"), _opt(dT, "dT"), I * J, V, Kout,
                                                         1 if accumulate else 0))
    return dW, dT


def tensormatmul_forward(F, Y, ctx=None):
    R, Kd, D = F.shape
    J = Y.shape[1]
    ctx = ctx or default_context(F.device.index or 0)
    out = torch.empty((R, J, D), dtype=torch.float32, device=F.device)
    ctx.check(ctx.lib.gf_tensormatmul_forward_f32(ctx.handle, _dev_f32(F, "F"), _dev_f32(Y, "Y"), _dev_f32(out, "Out"), R, Kd, J, D))
    return out


def tensormatmul_backward(G, F, Y, dF=None, dY=None, accumulate=False, ctx=None):
    R, Kd, D = F.shape
    J = Y.shape[1]
    ctx = ctx or default_context(F.device.index or 0)
    ctx.check(ctx.lib.gf_tensormatmul_backward_f32(ctx.handle, _dev_f32(G, "G"), _dev_f32(F, "F"), _dev_f32(Y, "Y"),
                                                   _opt(dF, "dF"), _opt(dY, "dY"), R, Kd, J, D, 1 if accumulate else 0))
    return dF, dY


def stack_forward(tensors, ctx=None):
    """StackTensor3D::forward on device: a list of equally sized CUDA tensors -> one contiguous [n, ...] tensor."""
    ctx = ctx or default_context(tensors[0].device.index or 0)
    per = tensors[0].numel()
    ptrs = torch.tensor([t.data_ptr() for t in tensors], dtype=torch.int64, device=tensors[0].device)
    out = torch.empty((len(tensors),) + tuple(tensors[0].shape), dtype=torch.float32, device=tensors[0].device)
    ctx.check(ctx.lib.gf_stack_forward_f32(ctx.handle, C.c_void_p(ptrs.data_ptr()), _dev_f32(out, "out"), len(tensors), per))
    return out


def stack_backward(G, grads, ctx=None):
    """StackTensor3D::backward on device: grads[r] += G[r]."""
    ctx = ctx or default_context(G.device.index or 0)
    per = grads[0].numel()
    ptrs = torch.tensor([t.data_ptr() for t in grads], dtype=torch.int64, device=G.device)
    ctx.check(ctx.lib.gf_stack_backward_f32(ctx.handle, _dev_f32(G, "G"), C.c_void_p(ptrs.data_ptr()), len(grads), per))
    return grads
