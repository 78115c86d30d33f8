"""CPU suite: the C-ABI shared library loads and exports exactly what include/gf_hip.h declares.  No compute calls."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "gf_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gf_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for must in ("gf_ctx_create", "gf_contract_forward_f32", "gf_contract_backward_f32", "gf_last_error",
                 "gf_contract_forward_host_f64", "gf_contract_backward_host_f64"):
        assert must in syms


def test_library_exports_every_declared_symbol(gf):
    from graphflow_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), "libgf_hip.so does not export %s" % name


def test_library_exports_nothing_the_header_does_not_declare(gf):
    """No undeclared gf_* entry points (test hooks, leftovers) in the product ABI: dynamic symbol table == header."""
    import subprocess
    from graphflow_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if re.search(r" [TW] gf_[a-z0-9_]+$", ln)})
    assert exported == declared_symbols(), sorted(set(exported) ^ set(declared_symbols()))


def test_python_prototype_table_matches_header():
    from graphflow_amd import _lib
    assert sorted(_lib.PROTOTYPES) == declared_symbols()


def test_version_string(gf):
    from graphflow_amd import _lib
    assert b"gfx950" in _lib.load().gf_version()


def test_no_cpu_fallback_without_device(gf):
    """On a box without a GPU the context must refuse to exist (and say why), not fall back to host code."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible; the refusal path is exercised on CPU-only boxes")
    from graphflow_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    st = lib.gf_ctx_create(C.byref(h), 0, None)
    assert st == _lib.GF_ERR_HIP and not h.value
    assert b"no CPU fallback" in lib.gf_last_error(None)
    with pytest.raises(gf.GraphFlowHipError):
        gf.Context(0)


def test_bad_arguments_are_rejected_without_device(gf):
    from graphflow_amd import _lib
    lib = _lib.load()
    assert lib.gf_contract_forward_f32(None, 18, None, None, None, 4, 4, 1) == _lib.GF_ERR_INVALID
    assert lib.gf_contract_workspace_bytes(18, 32, 64, 256) > 0
    assert lib.gf_contract_workspace_bytes(18, 0, 64, 256) == 0
    # data-parallel entry points: declared, exported, and refusing a null context rather than crashing
    assert lib.gf_dist_init(None, None, 0, 1) == _lib.GF_ERR_INVALID
    assert lib.gf_dist_allreduce_sum_f32(None, None, 0) == _lib.GF_ERR_INVALID
    assert lib.gf_dist_rank(None) == 0 and lib.gf_dist_world(None) == 1
    assert lib.gf_ctx_set_option(None, _lib.GF_OPT_R18_GENERIC_KERNELS, 1) == _lib.GF_ERR_INVALID


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure; nothing under graphflow_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "graphflow_amd")):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "pyoracle" not in src and "gf_oracle" not in src and "libgf_ref" not in src, fn
