"""CPU suite for the SMP_omega row (SURVEY 8 a-H): the oracle restatement and the library's host graph preparation,
both pinned to golden vectors captured from the real reference SMP_omega (tests/golden/smp.npz)."""
import ctypes as C

import numpy as np
import pytest

from inputs import synthetic_molecule, smp_params
from util import golden_cases


def cases(golden):
    return golden_cases(golden, "smp_")


def fields_of(phi):
    L1, V, _ = phi.shape
    return [[list(phi[l, v, 1:1 + phi[l, v, 0]]) for v in range(V)] for l in range(L1)]


def test_oracle_matches_reference_goldens(golden):
    from oracle import smp_oracle
    cs = cases(golden)
    assert len(cs) >= 8
    for tag, c in cs.items():
        L, Cn, D, cap, wl = (int(x) for x in c["cfg"])
        o = smp_oracle.run(c["adj"], c["feature"], float(c["target"][0]), c["params"].astype(np.float64), L, Cn, D, cap, bool(wl))
        assert [[list(map(int, f)) for f in lv] for lv in o["phi"]] == fields_of(c["phi"]), tag
        assert abs(o["predict"] - c["predict"][0]) <= 1e-10 * max(1, abs(c["predict"][0])), tag
        assert abs(o["loss"] - c["loss"][0]) <= 1e-10 * max(1, abs(c["loss"][0])), tag
        assert np.abs(o["graph_feature"] - c["graph_feature"]).max() <= 1e-10 * max(1, np.abs(c["graph_feature"]).max()), tag
        assert np.abs(o["grads"] - c["grads"]).max() <= 1e-9 * max(1, np.abs(c["grads"]).max()), tag


def test_library_host_preparation_matches_reference_goldens(gf, golden):
    """gf_smp_prepare_molecule_host is pure host code: receptive fields must equal the reference's, list for list."""
    from graphflow_amd import _lib
    from graphflow_amd.smp import SMPConfig
    lib = _lib.load()
    for tag, c in cases(golden).items():
        L, Cn, D, cap, wl = (int(x) for x in c["cfg"])
        adj = np.ascontiguousarray(c["adj"], dtype=np.int32)
        feat = np.ascontiguousarray(c["feature"], dtype=np.float64)
        V, F = feat.shape
        cfg = SMPConfig(L, Cn, F, D, cap, wl)
        phi = np.zeros((L + 1, V, cap + 1), dtype=np.int32)
        wlf = np.zeros((V, F * (D + 1)))
        st = lib.gf_smp_prepare_molecule_host(C.byref(cfg), V, adj.ctypes.data_as(C.POINTER(C.c_int)),
                                              feat.ctypes.data_as(C.POINTER(C.c_double)),
                                              phi.ctypes.data_as(C.POINTER(C.c_int)), wlf.ctypes.data_as(C.POINTER(C.c_double)))
        assert st == 0
        assert fields_of(phi) == fields_of(c["phi"]), tag
        # WL features: row sums count the vertices within nDepth hops, first block is the raw feature
        assert np.array_equal(wlf[:, :F], feat)


def test_cap_drops_whole_hop_shells(golden):
    c = cases(golden)["smp_syn17_cap6"]
    phi = fields_of(c["phi"])
    assert max(len(f) for f in phi[3]) <= 6
    for v, f in enumerate(phi[3]):
        assert v in f  # the centre (hop 0) always survives the cap (assert A[0] == v in the reference, before re-sorting)


def test_oracle_against_live_reference_random_molecules():
    from oracle import pyoracle, smp_oracle
    if pyoracle.reference() is None:
        pytest.skip("oracle/_ref/libgf_ref.so not present")
    for seed in (11, 12, 13):
        adj, feat, tgt = synthetic_molecule(seed, nV=5 + seed % 7)
        params = smp_params(4, 5, 2, 2, seed)
        r = pyoracle.reference_smp_omega(adj, feat, tgt, params, 2, 4, 2, 8)
        o = smp_oracle.run(adj, feat, tgt, params, 2, 4, 2, 8)
        assert r["phi"] == [[list(map(int, f)) for f in lv] for lv in o["phi"]]
        assert abs(r["predict"] - o["predict"]) <= 1e-10 * max(1, abs(r["predict"]))
        assert np.abs(r["grads"] - o["grads"]).max() <= 1e-9 * max(1, np.abs(r["grads"]).max())


def test_checkpoint_fixture_holds_the_golden_parameters(golden):
    """The reference-written text checkpoint (SMP_omega::save_model, 6 significant digits) is the syn12 parameter set in
    registration order H, (K_l, b_l)..., W -- the order the flat device buffer uses."""
    import os
    c = golden_cases(golden, "smp_syn12")["smp_syn12"]
    raw = open(os.path.join(os.path.dirname(__file__), "golden", "smp_syn12_checkpoint.txt")).read()
    assert raw.endswith(" ") and "\n" not in raw
    text = np.array(raw.split(), dtype=np.float64)
    assert text.size == c["params"].size
    assert np.max(np.abs(text - c["params"]) / np.maximum(np.abs(c["params"]), 1e-30)) < 1e-5
