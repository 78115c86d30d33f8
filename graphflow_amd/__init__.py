"""graphflow_amd -- MI355X-native kernels for GraphFlow's second-order CCN/SMP hot path.

The product is graphflow_amd/csrc/libgf_hip.so (C ABI in include/gf_hip.h) plus the header-only C++ op classes in
graphflow_amd/host/.  This Python package is plumbing for tests and benchmarks: torch supplies device memory and
streams, every op call goes straight through the C ABI.  Nothing here computes on the CPU.
"""
from . import _lib
from .ops import (Context, GraphFlowHipError, contract18_dropout_backward, contract18_dropout_forward,  # noqa: F401
                  contract_backward, contract_forward, contract_workspace_bytes, custommatmultensor_backward,
                  custommatmultensor_forward, default_context, matmul_backward, matmul_forward,
                  mattensormul_backward, mattensormul_forward, stack_backward, stack_forward,
                  tensormatmul_backward, tensormatmul_forward)

__all__ = ["Context", "GraphFlowHipError", "contract_forward", "contract_backward", "contract_workspace_bytes",
           "default_context", "build"]


def build(verbose=False):
    return _lib.build(verbose)
