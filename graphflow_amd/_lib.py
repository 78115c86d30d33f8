"""Loads graphflow_amd/csrc/libgf_hip.so (the C-ABI of include/gf_hip.h) and declares its prototypes.

There is NO fallback: if the shared library is missing or a symbol is absent this module raises, and every op in the
package fails with it.  Building happens in-tree (`make -C graphflow_amd/csrc`, driven by __graft_entry__.build()).
"""
import ctypes as C
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("GF_HIP_LIBRARY") or os.path.join(_CSRC, "libgf_hip.so")   # (GF_HIP_LIBRARY: an experimental build, tools/ab_lib.sh)

GF_OK, GF_ERR_INVALID, GF_ERR_HIP, GF_ERR_NOMEM, GF_ERR_UNSUPPORTED, GF_ERR_TIMEOUT = range(6)
GF_OPT_R18_GENERIC_KERNELS = 1
GF_OPT_SMP_FP32_PRODUCTS = 2
GF_DIST_ID_BYTES = 128

_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/gf_hip.h one to one (tests/test_abi.py checks the two agree)
PROTOTYPES = {
    "gf_ctx_create": (C.c_int, [C.POINTER(_vp), C.c_int, _vp]),
    "gf_ctx_use_private_stream": (C.c_int, [_vp]),
    "gf_ctx_destroy": (C.c_int, [_vp]),
    "gf_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "gf_ctx_get_stream": (_vp, [_vp]),
    "gf_ctx_synchronize": (C.c_int, [_vp]),
    "gf_ctx_reserve": (C.c_int, [_vp, C.c_size_t]),
    "gf_ctx_set_timing": (C.c_int, [_vp, C.c_int]),
    "gf_ctx_timing_count": (C.c_int, [_vp]),
    "gf_ctx_timing_get": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "gf_hbm_copy_probe_f32": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "gf_last_error": (C.c_char_p, [_vp]),
    "gf_version": (C.c_char_p, []),
    "gf_ctx_set_option": (C.c_int, [_vp, C.c_int, C.c_int]),
    "gf_dist_unique_id": (C.c_int, [_vp, _vp]),
    "gf_dist_init": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "gf_dist_finalize": (C.c_int, [_vp]),
    "gf_dist_rank": (C.c_int, [_vp]),
    "gf_dist_world": (C.c_int, [_vp]),
    "gf_dist_allreduce_sum_f32": (C.c_int, [_vp, _vp, C.c_size_t]),
    "gf_dist_broadcast_f32": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int]),
    "gf_dist_quiesce": (C.c_int, [_vp]),
    "gf_contract_forward_f32": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    "gf_contract_backward_f32": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gf_contract_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "gf_contract_forward_host_f64": (C.c_int, [_vp, C.c_int, C.POINTER(_dp), _dp, _dp, C.c_int, C.c_int]),
    "gf_contract_backward_host_f64": (C.c_int, [_vp, C.c_int, _dp, _dp, C.POINTER(_dp), C.c_int, C.c_int]),
    "gf_contract_forward_host_f32": (C.c_int, [_vp, C.c_int, C.POINTER(_fp), _fp, _fp, C.c_int, C.c_int]),
    "gf_contract_backward_host_f32": (C.c_int, [_vp, C.c_int, _fp, _fp, C.POINTER(_fp), C.c_int, C.c_int]),
}

_i = C.c_int
PROTOTYPES.update({
    "gf_contract18_dropout_forward_f32": (_i, [_vp, C.c_uint, C.c_float, _vp, _vp, _vp, _i, _i, _i]),
    "gf_contract18_dropout_backward_f32": (_i, [_vp, C.c_uint, _vp, _vp, _vp, _i, _i, _i, _i]),
    "gf_contract18_dropout_forward_host_f64": (_i, [_vp, C.c_uint, C.c_double, C.POINTER(_dp), _dp, _dp, _i, _i]),
    "gf_contract18_dropout_backward_host_f64": (_i, [_vp, C.c_uint, _dp, _dp, C.POINTER(_dp), _i, _i]),
    "gf_contract18_dropout_forward_host_f32": (_i, [_vp, C.c_uint, C.c_double, C.POINTER(_fp), _fp, _fp, _i, _i]),
    "gf_contract18_dropout_backward_host_f32": (_i, [_vp, C.c_uint, _fp, _fp, C.POINTER(_fp), _i, _i]),
})
for _sfx, _t in (("f64", _dp), ("f32", _fp)):
    PROTOTYPES["gf_matmul_forward_host_" + _sfx] = (_i, [_vp, _t, _t, _t, _i, _i, _i])
    PROTOTYPES["gf_stack_forward_host_" + _sfx] = (_i, [_vp, C.POINTER(_t), _t, _i, C.c_size_t])
    PROTOTYPES["gf_stack_backward_host_" + _sfx] = (_i, [_vp, _t, C.POINTER(_t), _i, C.c_size_t])
    PROTOTYPES["gf_matmul_backward_host_" + _sfx] = (_i, [_vp, _t, _t, _t, _t, _t, _i, _i, _i])
    PROTOTYPES["gf_custommatmultensor_forward_host_" + _sfx] = (_i, [_vp, _t, _t, _t, C.c_longlong, _i, _i])
    PROTOTYPES["gf_custommatmultensor_backward_host_" + _sfx] = (_i, [_vp, _t, _t, _t, _t, _t, C.c_longlong, _i, _i])
    for _op in ("mattensormul", "tensormatmul"):
        PROTOTYPES["gf_%s_forward_host_%s" % (_op, _sfx)] = (_i, [_vp, _t, _t, _t, _i, _i, _i, _i])
        PROTOTYPES["gf_%s_backward_host_%s" % (_op, _sfx)] = (_i, [_vp, _t, _t, _t, _t, _t, _i, _i, _i, _i])
PROTOTYPES.update({
    "gf_matmul_forward_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i]),
    "gf_matmul_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "gf_mattensormul_forward_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "gf_mattensormul_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "gf_tensormatmul_forward_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "gf_tensormatmul_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "gf_custommatmultensor_forward_f32": (_i, [_vp, _vp, _vp, _vp, C.c_longlong, _i, _i]),
    "gf_custommatmultensor_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_longlong, _i, _i, _i]),
    "gf_smp_create": (_i, [_vp, _vp, C.POINTER(_vp)]),
    "gf_smp_destroy": (_i, [_vp]),
    "gf_smp_param_count": (C.c_size_t, [_vp]),
    "gf_smp_prepare": (_i, [_vp, _i, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp]),
    "gf_smp_prepare_coulomb": (_i, [_vp, _i, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, _dp]),
    "gf_smp_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "gf_smp_backward": (_i, [_vp, _vp, _vp, _i]),
    "gf_smp_set_grad_allreduce": (_i, [_vp, _i]),
    "gf_smp_feature_width": (C.c_size_t, [_vp]),
    "gf_smp_backward_features": (_i, [_vp, _vp, _vp, _vp, _i]),
    "gf_head_param_count": (C.c_size_t, [_i, C.POINTER(C.c_int)]),
    "gf_head_work_floats": (C.c_size_t, [_i, C.POINTER(C.c_int), _i]),
    "gf_head_forward_f32": (_i, [_vp, _i, C.POINTER(C.c_int), _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "gf_head_backward_f32": (_i, [_vp, _i, C.POINTER(C.c_int), _vp, _i, _vp, _vp, _vp, _vp]),
    "gf_ctx_set_timing_filter": (_i, [_vp, C.c_char_p]),
    "gf_smp_parameters_upload": (_i, [_vp, _fp]),
    "gf_smp_parameters_download": (_i, [_vp, _fp, _fp]),
    "gf_smp_forward_host": (_i, [_vp, _dp, _dp, _dp, _dp]),
    "gf_smp_adam_step": (_i, [_vp, _vp, _vp, C.c_double, _i]),
    "gf_smp_adam_reset": (_i, [_vp]),
    "gf_smp_momentum_step": (_i, [_vp, _vp, _vp, C.c_double, _i, C.c_double]),
    "gf_smp_uniform_init_host": (_i, [_vp, _fp]),
    "gf_smp_save_model": (_i, [_vp, _vp, C.c_char_p]),
    "gf_smp_load_model": (_i, [_vp, _vp, C.c_char_p]),
    "gf_smp_set_fused": (_i, [_vp, _i]),
    "gf_smp_device_bytes": (_i, [_vp, _vp, _vp]),
    "gf_smp_level_products_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "gf_smp_level_wgrad_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "gf_smp_prepare_molecule_host": (_i, [_vp, _i, C.POINTER(C.c_int), _dp, C.POINTER(C.c_int), _dp]),
    "gf_smp_receptive_field": (_i, [_vp, _i, _i, _i, C.POINTER(C.c_int), _i]),
    "gf_smp_read_activation": (C.c_longlong, [_vp, _i, _i, _i, _vp, C.c_size_t]),
    "gf_smp_read_reduced_adjacency": (C.c_longlong, [_vp, _i, _i, _i, _vp, C.c_size_t]),
    "gf_smp_level_sizes": (_i, [_vp, _i, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "gf_smp_level_present_rows": (C.c_longlong, [_vp, _i]),
    "gf_smp_level_pairs": (C.c_longlong, [_vp, _i]),
    "gf_smp_level_covered_rows": (C.c_longlong, [_vp, _i]),
    "gf_stack_forward_f32": (_i, [_vp, _vp, _vp, _i, C.c_size_t]),
    "gf_stack_backward_f32": (_i, [_vp, _vp, _vp, _i, C.c_size_t]),
})

PROTOTYPES.update({
    "gf_smp_dropout_masks": (_i, [_vp, _vp, C.c_float]),
    "gf_adam_step_f32": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_size_t, C.c_double, _i, C.c_ulonglong]),
    "gf_smp_model_create": (_i, [_vp, _vp, C.POINTER(_vp)]),
    "gf_smp_model_destroy": (_i, [_vp]),
    "gf_smp_model_param_count": (C.c_size_t, [_vp]),
    "gf_smp_model_set_mode": (_i, [_vp, _i]),
    "gf_smp_model_prepare": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gf_smp_model_forward": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "gf_smp_model_backward": (_i, [_vp, _vp, _vp, _i]),
    "gf_smp_model_uniform_init_host": (_i, [_vp, _fp]),
    "gf_smp_model_parameters_upload": (_i, [_vp, _fp]),
    "gf_smp_model_parameters_download": (_i, [_vp, _fp, _fp]),
    "gf_smp_model_forward_host": (_i, [_vp, _dp, _dp, _dp]),
    "gf_smp_model_adam_step": (_i, [_vp, C.c_double, _i]),
})

_lib = None


def build(verbose=False):
    """hipcc --offload-arch=gfx950 build of the shared library (cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", _CSRC, "-j4"], stdout=out)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "graphflow_amd: %s is missing -- build it with `make -C graphflow_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and this table drift apart
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
