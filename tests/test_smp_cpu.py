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
    assert len(cs) >= 9 and any("coulomb" in c for c in cs.values())
    for tag, c in cs.items():
        L, Cn, D, cap, wl = (int(x) for x in c["cfg"])
        nK, custom = (int(x) for x in c["wiring"]) if "wiring" in c else (18, 0)
        o = smp_oracle.run(c["adj"], c["feature"], float(c["target"][0]), c["params"].astype(np.float64), L, Cn, D, cap, bool(wl),
                           coulomb=c.get("coulomb"), nK=nK, custom=bool(custom))
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


def _train_case():
    import os
    from inputs import toy_molecules
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "smp_train.npz"))
    L, Cn, D, cap, maxV, seed, nIter = (int(x) for x in z["train__cfg"])
    mols = [(adj, feat) for _, adj, feat, _ in toy_molecules()]
    return z, (L, Cn, D, cap, maxV, seed, nIter), mols


def test_uniform_init_draws_the_reference_weights(gf):
    """gf_smp_uniform_init_host after srand(7) == the weights SMP_omega's constructor drew after srand(7) in the reference
    (SMP_omega.h:334-338, GraphFlow.h:1297-1306); fixture tests/golden/smp_train.npz."""
    from graphflow_amd import _lib
    from graphflow_amd.smp import SMPConfig
    z, (L, Cn, D, cap, maxV, seed, nIter), mols = _train_case()
    lib = _lib.load()
    cfg = SMPConfig(L, Cn, mols[0][1].shape[1], D, cap, 1)
    out = np.zeros(z["train__params0"].size, dtype=np.float32)
    C.CDLL(None).srand(seed)
    assert lib.gf_smp_uniform_init_host(C.byref(cfg), out.ctypes.data_as(C.POINTER(C.c_float))) == 0
    assert np.array_equal(out, z["train__params0"].astype(np.float32))


def test_oracle_batchlearn_matches_reference():
    """Three BatchLearn steps (summed gradients + Adam::Learn(alpha, nBatch) with its per-element beta powers) restated
    with the oracle reproduce the real reference's parameters and before/after losses."""
    from oracle import smp_oracle
    z, (L, Cn, D, cap, maxV, seed, nIter), mols = _train_case()
    tg = z["train__targets"]
    p = z["train__params0"].copy()
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    n0 = 0

    def total_loss(p):
        return sum(smp_oracle.run(a, f, float(t), p, L, Cn, D, cap, True, want_grads=False)["loss"] for (a, f), t in zip(mols, tg))

    for it in range(nIter):
        before = total_loss(p)
        g = sum(smp_oracle.run(a, f, float(t), p, L, Cn, D, cap, True)["grads"] for (a, f), t in zip(mols, tg))
        p, m, v, n0 = smp_oracle.adam_learn(p, g, m, v, n0, float(z["train__lr"][0]), len(mols))
        after = total_loss(p)
        assert abs(before - z["train__losses"][it, 0]) <= 1e-9 * max(1, before)
        assert abs(after - z["train__losses"][it, 1]) <= 1e-7 * max(1, after)
    assert np.abs(p - z["train__params"]).max() <= 1e-9


def test_oracle_against_live_reference_edge_molecules():
    """Single atom, bonded pair, disconnected fragments, isolated vertex, clique, long path (tests/golden/inputs.py)."""
    from inputs import edge_molecules
    from oracle import pyoracle, smp_oracle
    if pyoracle.reference() is None:
        pytest.skip("oracle/_ref/libgf_ref.so not present")
    for name, adj, feat, tgt in edge_molecules():
        params = smp_params(4, 5, 2, 3, 21)
        r = pyoracle.reference_smp_omega(adj, feat, tgt, params, 3, 4, 2, 6, max_nVertices=12)
        o = smp_oracle.run(adj, feat, tgt, params, 3, 4, 2, 6)
        assert r["phi"] == [[list(map(int, f)) for f in lv] for lv in o["phi"]], name
        assert abs(r["predict"] - o["predict"]) <= 1e-10 * max(1, abs(r["predict"])), name
        assert np.abs(r["grads"] - o["grads"]).max() <= 1e-9 * max(1, np.abs(r["grads"]).max()), name


def test_c_port_matches_reference_goldens(golden):
    """oracle/smp_port.c (the C restatement of the step body; also the "port"-kind CPU baseline of bench.py) against the
    real-reference goldens: SMP_omega cases with the 18-contraction wiring, Coulomb adjacency included."""
    from oracle import pyoracle
    n = 0
    for tag, c in cases(golden).items():
        if "wiring" in c or "beta" in tag:
            continue
        L, Cn, D, cap, wl = (int(x) for x in c["cfg"])
        o = pyoracle.port_smp_molecule(c["adj"], c["feature"], float(c["target"][0]), c["params"].astype(np.float64), L, Cn, D, cap,
                                       bool(wl), coulomb=c.get("coulomb"))
        assert abs(o["predict"] - c["predict"][0]) <= 1e-10 * max(1, abs(c["predict"][0])), tag
        assert np.abs(o["graph_feature"] - c["graph_feature"]).max() <= 1e-10 * max(1, np.abs(c["graph_feature"]).max()), tag
        assert np.abs(o["grads"] - c["grads"]).max() <= 1e-9 * max(1, np.abs(c["grads"]).max()), tag
        n += 1
    assert n >= 8


def test_c_port_batch_driver_is_thread_count_independent():
    """gfo_smp_batch (waves of nThreads molecules, the shape of Threaded_BatchLearn) sums the same gradients for any thread
    count, and they equal the numpy restatement's."""
    from oracle import pyoracle, smp_oracle
    L, Cn, D, cap = 2, 4, 2, 8
    params = smp_params(Cn, 5, D, L, 3)
    mols = [synthetic_molecule(s, 6 + s)[:2] for s in range(5)]
    tg = [6.0 + s for s in range(5)]
    _, p1, l1, g1 = pyoracle.port_smp_batch(mols, tg, params, L, Cn, D, cap, 1)
    _, p3, l3, g3 = pyoracle.port_smp_batch(mols, tg, params, L, Cn, D, cap, 3)
    assert np.array_equal(p1, p3) and np.array_equal(l1, l3)
    assert np.abs(g1 - g3).max() <= 1e-12 * max(1, np.abs(g1).max())   # (the order of the serial add differs with the wave size)
    ref = [smp_oracle.run(a, f, t, params, L, Cn, D, cap) for (a, f), t in zip(mols, tg)]
    assert np.abs(g1 - sum(r["grads"] for r in ref)).max() <= 1e-9 * max(1, np.abs(g1).max())
    assert np.abs(p1 - np.array([r["predict"] for r in ref])).max() <= 1e-10 * max(1, np.abs(p1).max())


def headline_golden():
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "smp_headline.npz"))
    c = {k.split("__")[1]: z[k] for k in z.files}
    L, Cn, D, cap, wl, seed_p = (int(x) for x in c["cfg"])
    params = smp_params(Cn, c["feature"].shape[1], D, L, seed_p)
    chk = np.array([params.sum(), np.abs(params).sum(), (params * np.arange(params.size)).sum()])
    assert np.array_equal(chk, c["params_checksum"]), "smp_params(seed) no longer reproduces the fixture's parameters"
    return c, (L, Cn, D, cap), params


def activation_digest(f):
    s = f.shape[0]
    i, j = np.meshgrid(np.arange(s), np.arange(s), indexing="ij")
    w = ((3 * i + 5 * j) % 7 - 3).astype(np.float64) / 4.0
    return np.stack([f.sum(axis=(0, 1)), (w[:, :, None] * f).sum(axis=(0, 1))])


def test_c_port_at_the_headline_shape(golden):
    """BASELINE configs[2]'s own shape -- L = 3, cap 29, C = 64, a 29-atom molecule -- where the numpy restatement is out of
    reach: the C port against the real reference's Feature / predict / loss / all 223,360 gradients and one full level-3
    activation (tests/golden/smp_headline.npz, about 8 s of CPU)."""
    from oracle import pyoracle
    c, (L, Cn, D, cap), params = headline_golden()
    l, v = (int(x) for x in c["act_picks"][2])   # the largest level-3 field
    o = pyoracle.port_smp_molecule(c["adj"], c["feature"], float(c["target"][0]), params, L, Cn, D, cap, True, activation=(l, v))
    assert abs(o["predict"] - c["predict"][0]) <= 1e-10 * abs(c["predict"][0])
    assert abs(o["loss"] - c["loss"][0]) <= 1e-10 * abs(c["loss"][0])
    assert np.abs(o["graph_feature"] - c["graph_feature"]).max() <= 1e-10 * np.abs(c["graph_feature"]).max()
    assert np.abs(o["grads"] - c["grads"]).max() <= 1e-6 * np.abs(c["grads"]).max()   # the fixture keeps gradients in fp32
    act = c["act_l%d_v%d" % (l, v)].astype(np.float64)
    assert np.abs(o["f"] - act).max() <= 1e-6 * np.abs(act).max()
    assert np.abs(activation_digest(o["f"]) - c["act_digest"][l, v]).max() <= 1e-10 * np.abs(c["act_digest"][l, v]).max()


def test_physics_host_preparation_matches_reference_goldens(gf):
    """The `_physics` / `_pairgraphs` receptive fields (no WL ordering, cap ordered by hop distance only:
    SMP_omega_physics.h:436-478) from gf_smp_prepare_molecule_host with cfg.physics = 1, list for list."""
    import os
    from graphflow_amd import _lib
    from graphflow_amd.smp import SMPConfig
    lib = _lib.load()
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "smp_physics.npz")) as z:
        cs = golden_cases({k: z[k] for k in z.files}, "physics_")
    checked = 0
    for tag, c in cs.items():
        towers, L, Cn, cap, _, _ = (int(x) for x in c["cfg"])
        for adj_k, feat_k, phi_k in (("adj", "feature", "phi"), ("adj2", "feature2", "phi2"))[:towers]:
            adj = np.ascontiguousarray(c[adj_k], dtype=np.int32)
            feat = np.ascontiguousarray(c[feat_k], dtype=np.float64)
            V, F = feat.shape
            cfg = SMPConfig(L, Cn, F, 0, min(cap, V) if "beta" in tag else cap, 0, 18, 0, 1)
            stride = cfg.max_receptive_field + 1
            phi = np.zeros((L + 1, V, stride), dtype=np.int32)
            st = lib.gf_smp_prepare_molecule_host(C.byref(cfg), V, adj.ctypes.data_as(C.POINTER(C.c_int)),
                                                  feat.ctypes.data_as(C.POINTER(C.c_double)), phi.ctypes.data_as(C.POINTER(C.c_int)), None)
            assert st == 0
            assert fields_of(phi) == fields_of(c[phi_k]), (tag, adj_k)
            checked += 1
    assert checked >= 10
