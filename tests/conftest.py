import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """fp64 CPU restatement (oracle/gf_oracle.c) -- the checker, never the product."""
    from oracle import pyoracle
    return pyoracle.oracle()


@pytest.fixture(scope="session")
def reference():
    """The real reference build (oracle/_ref/libgf_ref.so) or None when it was not shipped."""
    from oracle import pyoracle
    return pyoracle.reference()


@pytest.fixture(scope="session")
def golden():
    g = {}
    for name in ("contractions", "contractions_wide", "contractions_16x8", "stack", "mixers", "smp", "smp_big", "dropout"):
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        with np.load(path) as z:
            g.update({k: z[k] for k in z.files})
    return g


@pytest.fixture(scope="session")
def gf():
    """The product package; importing it loads libgf_hip.so and fails loudly if it is missing."""
    import graphflow_amd
    from graphflow_amd import _lib
    _lib.load()
    return graphflow_amd
