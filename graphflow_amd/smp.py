"""Python handle on the batched SMP_omega driver of the C ABI (gf_smp_*).  Plumbing only: torch owns the parameter,
gradient and output buffers; everything else lives in libgf_hip.so."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .ops import default_context


class SMPConfig(C.Structure):
    _fields_ = [("nLevels", C.c_int), ("nChanels", C.c_int), ("nFeatures", C.c_int), ("nDepth", C.c_int),
                ("max_receptive_field", C.c_int), ("has_WL_ordering", C.c_int), ("nContractions", C.c_int),
                ("custom_matmul", C.c_int), ("physics", C.c_int)]


class SMPOmega:
    """Batched SMP_omega (GraphFlow/SMP_omega.h).  Parameters/gradients are one flat fp32 tensor in the reference's
    registration order: H[C, F(D+1)], (K_l[18C, C], b_l[C]) for l = 1..L, W[C]."""

    def __init__(self, nLevels, nChanels, nFeatures, nDepth, max_receptive_field, has_WL_ordering=True, ctx=None,
                 nContractions=18, custom_matmul=False, physics=False):
        """nContractions / custom_matmul select the SMP_2D_ver6 (10, True) / ver7 (50, True) / ver8 (18, True) wirings;
        physics=True makes the handle one tower of the _physics / _pairgraphs models (see gf_smp_config.physics)."""
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        self.cfg = SMPConfig(nLevels, nChanels, nFeatures, nDepth, max_receptive_field, 1 if has_WL_ordering else 0,
                             nContractions, 1 if custom_matmul else 0, 1 if physics else 0)
        h = C.c_void_p()
        self.ctx.check(self.lib.gf_smp_create(self.ctx.handle, C.byref(self.cfg), C.byref(h)))
        self.handle = h
        self.n_params = self.lib.gf_smp_param_count(h)
        self.n_mol = 0

    @staticmethod
    def pack(molecules, coulomb=None):
        """The flat host arrays gf_smp_prepare takes (vertex counts, adjacency and feature matrices back to back): a data
        loader builds these once per batch; prepare() accepts the result in place of the molecule list."""
        nV = np.array([len(m[0]) for m in molecules], dtype=np.int32)
        adj = np.concatenate([np.ascontiguousarray(m[0], dtype=np.int32).ravel() for m in molecules])
        feat = np.concatenate([np.ascontiguousarray(m[1], dtype=np.float64).ravel() for m in molecules])
        cm = None
        if coulomb is not None:
            cm = np.concatenate([np.ascontiguousarray(c, dtype=np.float64).ravel() for c in coulomb])
            assert cm.size == adj.size
        return {"nV": nV, "adj": adj, "feature": feat, "coulomb": cm}

    def prepare(self, molecules, coulomb=None):
        """molecules: list of (adj int[V,V], feature float[V,F]) or the dict pack() returns; coulomb: optional list of
        float[V,V] Coulomb matrices (the use_coulomb variant of SMP_omega).  Host graph preparation + upload (blocking)."""
        pk = molecules if isinstance(molecules, dict) else self.pack(molecules, coulomb)
        nV, adj, feat, cm = pk["nV"], pk["adj"], pk["feature"], pk["coulomb"]
        molecules = nV
        self.ctx.check(self.lib.gf_smp_prepare_coulomb(self.handle, len(nV), nV.ctypes.data_as(C.POINTER(C.c_int)),
                                                       adj.ctypes.data_as(C.POINTER(C.c_int)),
                                                       feat.ctypes.data_as(C.POINTER(C.c_double)),
                                                       cm.ctypes.data_as(C.POINTER(C.c_double)) if cm is not None else None))
        self.n_mol = len(molecules)
        dev = self.ctx.device
        self.predict = torch.empty(self.n_mol, dtype=torch.float32, device=dev)
        self.loss = torch.empty(self.n_mol, dtype=torch.float32, device=dev)
        self.feature = torch.empty((self.n_mol, int(self.lib.gf_smp_feature_width(self.handle))), dtype=torch.float32, device=dev)

    def _flat(self, t, name):
        """params / grads must be exactly the flat model: float32, contiguous, on the context's device, n_params long."""
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                and t.numel() == self.n_params and t.device == self.ctx.device):
            raise TypeError("%s must be a contiguous float32 tensor of %d elements on %s" % (name, self.n_params, self.ctx.device))
        return C.c_void_p(t.data_ptr())

    def backward_features(self, params, grads, d_feature, accumulate=False):
        """Physics tower: the reverse sweep from the gradient of the feature rows (what the head's backward returns)."""
        self._flat(params, "params")
        self._flat(grads, "grads")
        if not (d_feature.is_cuda and d_feature.dtype == torch.float32 and d_feature.is_contiguous() and d_feature.shape == self.feature.shape):
            raise TypeError("d_feature must be a contiguous float32 CUDA tensor shaped like the feature rows")
        self.ctx.check(self.lib.gf_smp_backward_features(self.handle, C.c_void_p(params.data_ptr()), C.c_void_p(grads.data_ptr()),
                                                         C.c_void_p(d_feature.data_ptr()), 1 if accumulate else 0))
        return grads

    def forward(self, params, targets=None):
        self._flat(params, "params")
        if self.cfg.physics:
            if targets is not None:
                raise TypeError("a physics tower has no loss of its own: targets go to the head")
            self.ctx.check(self.lib.gf_smp_forward(self.handle, C.c_void_p(params.data_ptr()), None, None, None,
                                                   C.c_void_p(self.feature.data_ptr())))
            return self.feature
        if targets is not None and not (targets.is_cuda and targets.dtype == torch.float32 and targets.is_contiguous()
                                        and targets.numel() == self.n_mol):
            raise TypeError("targets must be a contiguous float32 CUDA tensor with one value per molecule (%d)" % self.n_mol)
        t = C.c_void_p(targets.data_ptr()) if targets is not None else None
        self.ctx.check(self.lib.gf_smp_forward(self.handle, C.c_void_p(params.data_ptr()), t,
                                               C.c_void_p(self.predict.data_ptr()), C.c_void_p(self.loss.data_ptr()),
                                               C.c_void_p(self.feature.data_ptr())))
        return self.predict, self.loss, self.feature

    def backward(self, params, grads, accumulate=False):
        self._flat(params, "params")
        self._flat(grads, "grads")
        self.ctx.check(self.lib.gf_smp_backward(self.handle, C.c_void_p(params.data_ptr()), C.c_void_p(grads.data_ptr()),
                                                1 if accumulate else 0))
        return grads

    def adam_step(self, params, grads, learning_rate, nBatch):
        """Adam::Learn(learning_rate, nBatch) as SMP_omega::BatchLearn applies it (SMP_omega.h:820-821); grads = batch sum."""
        self._flat(params, "params")
        self._flat(grads, "grads")
        self.ctx.check(self.lib.gf_smp_adam_step(self.handle, C.c_void_p(params.data_ptr()), C.c_void_p(grads.data_ptr()),
                                                 float(learning_rate), int(nBatch)))
        return params

    def momentum_step(self, params, grads, learning_rate, nBatch, gamma=0.9):
        """Momentum::Learn(learning_rate, nBatch) (Momentum.h:64-71): the optimiser of SMP_2D_ver6-8."""
        self._flat(params, "params")
        self._flat(grads, "grads")
        self.ctx.check(self.lib.gf_smp_momentum_step(self.handle, C.c_void_p(params.data_ptr()), C.c_void_p(grads.data_ptr()),
                                                     float(learning_rate), int(nBatch), float(gamma)))
        return params

    def adam_reset(self):
        self.ctx.check(self.lib.gf_smp_adam_reset(self.handle))

    def uniform_init(self):
        """Initial weights exactly as SMP_omega's constructor draws them from rand() (host numpy array; call srand first)."""
        out = np.zeros(self.n_params, dtype=np.float32)
        st = self.lib.gf_smp_uniform_init_host(C.byref(self.cfg), out.ctypes.data_as(C.POINTER(C.c_float)))
        if st != 0:
            raise RuntimeError("gf_smp_uniform_init_host failed")
        return out

    def save_model(self, params, path):
        """SMP_omega::save_model (SMP_omega.h:1033-1042): text checkpoint the reference's load_model reads."""
        self.ctx.check(self.lib.gf_smp_save_model(self.handle, C.c_void_p(params.data_ptr()), str(path).encode()))

    def load_model(self, params, path):
        """SMP_omega::load_model (SMP_omega.h:1044-1055) into the flat device parameter buffer."""
        self.ctx.check(self.lib.gf_smp_load_model(self.handle, C.c_void_p(params.data_ptr()), str(path).encode()))
        return params

    def set_fused(self, on=True):
        """Fused level kernels (default) vs the op-by-op pipeline; both give the same results within fp32 rounding."""
        self.ctx.check(self.lib.gf_smp_set_fused(self.handle, 1 if on else 0))

    def device_bytes(self):
        """(bytes held by the current batch, bytes the handle's pool keeps in total)."""
        used, res = C.c_size_t(0), C.c_size_t(0)
        self.ctx.check(self.lib.gf_smp_device_bytes(self.handle, C.byref(used), C.byref(res)))
        return int(used.value), int(res.value)

    def receptive_field(self, mol, level, v):
        buf = (C.c_int * 4096)()
        n = self.lib.gf_smp_receptive_field(self.handle, mol, level, v, buf, 4096)
        return list(buf[:n])

    def activation(self, mol, level, v):
        """f_level[v] of molecule `mol` after forward(): numpy [s, s, C] (level[l]->f[v]->value in the reference)."""
        s = len(self.receptive_field(mol, level, v))
        ch = self.cfg.nChanels
        if self.cfg.physics:   # a physics tower halves its channels from level to level (SMP_omega_physics.h:141-156)
            for _ in range(level):
                ch = max(1, ch // 2)
        out = np.empty((s, s, ch), dtype=np.float32)
        n = self.lib.gf_smp_read_activation(self.handle, mol, level, v, out.ctypes.data_as(C.c_void_p), out.size)
        if n != out.size:
            raise RuntimeError("gf_smp_read_activation(%d, %d, %d) returned %d" % (mol, level, v, n))
        return out

    def reduced_adjacency(self, mol, level, v):
        """Reduced adjacency [s, s] of f_level[v] (level[l]->adj[v] in the reference), level >= 1."""
        s = len(self.receptive_field(mol, level, v))
        out = np.empty((s, s), dtype=np.float32)
        n = self.lib.gf_smp_read_reduced_adjacency(self.handle, mol, level, v, out.ctypes.data_as(C.c_void_p), out.size)
        if n != out.size:
            raise RuntimeError("gf_smp_read_reduced_adjacency(%d, %d, %d) returned %d" % (mol, level, v, n))
        return out

    def set_grad_allreduce(self, on=True):
        """Data-parallel runs (ctx.dist_init): backward() leaves the gradient summed over all ranks (default) or local."""
        self.ctx.check(self.lib.gf_smp_set_grad_allreduce(self.handle, 1 if on else 0))

    def level_sizes(self, level):
        a, b, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
        self.ctx.check(self.lib.gf_smp_level_sizes(self.handle, level, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def level_present_rows(self, level):
        """rows (a, b) of the level whose slab row is not structurally zero (gf_smp_level_present_rows)"""
        return int(self.lib.gf_smp_level_present_rows(self.handle, level))

    def level_pairs(self, level):
        """(node, neighbour) pairs of the level = sum of its receptive-field sizes (gf_smp_level_pairs)"""
        return int(self.lib.gf_smp_level_pairs(self.handle, level))

    def level_covered_rows(self, level):
        """rows (b, c) of the level that some source covers (gf_smp_level_covered_rows)"""
        return int(self.lib.gf_smp_level_covered_rows(self.handle, level))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gf_smp_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SMPModelConfig(C.Structure):
    _fields_ = [("nTowers", C.c_int), ("nLevels", C.c_int), ("nChanels", C.c_int), ("max_receptive_field", C.c_int),
                ("nFeatures", C.c_int * 2), ("nKept", C.c_int)]


class SMPModel:
    """The _physics (one tower) / _pairgraphs (two towers, nKept > 0: SMP_sigma_pairgraphs) models of GraphFlow through
    gf_smp_model_*.  Parameters / gradients: one flat fp32 tensor in the class's registration order."""

    def __init__(self, nLevels, nChanels, max_receptive_field, nFeatures, nKept=0, ctx=None):
        feats = list(nFeatures) if isinstance(nFeatures, (list, tuple)) else [nFeatures]
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        self.cfg = SMPModelConfig(len(feats), nLevels, nChanels, max_receptive_field, (C.c_int * 2)(*(feats + [0])[:2]), nKept)
        h = C.c_void_p()
        self.ctx.check(self.lib.gf_smp_model_create(self.ctx.handle, C.byref(self.cfg), C.byref(h)))
        self.handle = h
        self.n_params = self.lib.gf_smp_model_param_count(h)
        self.n_mol = 0

    @staticmethod
    def _pack(graphs):
        nV = np.array([len(g[0]) for g in graphs], dtype=np.int32)
        adj = np.concatenate([np.ascontiguousarray(g[0], dtype=np.int32).ravel() for g in graphs])
        feat = np.concatenate([np.ascontiguousarray(g[1], dtype=np.float64).ravel() for g in graphs])
        return nV, adj, feat

    def prepare(self, graphs1, graphs2=None):
        a = self._pack(graphs1)
        b = self._pack(graphs2) if graphs2 is not None else (None, None, None)
        ptr = lambda x: x.ctypes.data_as(C.c_void_p) if x is not None else None  # noqa: E731
        self.ctx.check(self.lib.gf_smp_model_prepare(self.handle, len(graphs1), ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(b[0]), ptr(b[1]), ptr(b[2])))
        self.n_mol = len(graphs1)
        dev = self.ctx.device
        self.predict = torch.empty(self.n_mol, dtype=torch.float32, device=dev)
        self.loss = torch.empty(self.n_mol, dtype=torch.float32, device=dev)

    def set_mode(self, train=True):
        self.ctx.check(self.lib.gf_smp_model_set_mode(self.handle, 1 if train else 0))

    def forward(self, params, targets=None):
        t = C.c_void_p(targets.data_ptr()) if targets is not None else None
        self.ctx.check(self.lib.gf_smp_model_forward(self.handle, C.c_void_p(params.data_ptr()), t, C.c_void_p(self.predict.data_ptr()),
                                                     C.c_void_p(self.loss.data_ptr())))
        return self.predict, self.loss

    def backward(self, params, grads, accumulate=False):
        self.ctx.check(self.lib.gf_smp_model_backward(self.handle, C.c_void_p(params.data_ptr()), C.c_void_p(grads.data_ptr()),
                                                      1 if accumulate else 0))
        return grads

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gf_smp_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
