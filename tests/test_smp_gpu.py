"""GPU parity suite for the batched SMP_omega driver (gf_smp_*) vs reference goldens and the fp64 oracle."""
import os

import numpy as np
import pytest

from inputs import f32exact, smp_params, synthetic_molecule, toy_molecules
from util import golden_cases, rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

# End-to-end tolerance = north_star's 1e-5 (relative to the largest reference magnitude, tests/util.py:rel_err), forward
# quantities and parameter gradients alike.  Measured maxima over the whole suite (printed by every test below, see
# profiles/r02_parity_margins.txt): about 1e-6 forward, 3e-6 gradients -- the bound holds with a 3x margin.  The loss is
# (y - t)^2 / 2 with |y - t| up to 1e4 x |t|: its relative error is twice the prediction's, hence 2 x.
TOL_FWD, TOL_GRAD = 1e-5, 1e-5
MARGINS = {}


def note(name, **errs):
    """Record the measured maxima (printed with -s and summarised by test_zz_print_margins)."""
    for k, v in errs.items():
        MARGINS[name + "." + k] = max(MARGINS.get(name + "." + k, 0.0), float(v))


def dev(x, dtype=np.float32):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=dtype)).cuda()


def run_batch(gf, mols, targets, params, L, C, F, D, cap, wl=True, fused=True, coulomb=None, wiring=(18, False)):
    from graphflow_amd.smp import SMPOmega
    net = SMPOmega(L, C, F, D, cap, wl, nContractions=wiring[0], custom_matmul=wiring[1])
    net.set_fused(fused)
    net.prepare(mols, coulomb=coulomb)
    p = dev(params)
    pred, loss, feat = net.forward(p, dev(targets))
    grads = torch.empty(net.n_params, device="cuda")
    net.backward(p, grads)
    out = (pred.cpu().numpy().astype(np.float64), loss.cpu().numpy().astype(np.float64),
           feat.cpu().numpy().astype(np.float64), grads.cpu().numpy().astype(np.float64), net)
    return out


@pytest.mark.parametrize("fused", [True, False])
def test_reference_goldens_one_molecule_at_a_time(gf, golden, fused):
    cs = golden_cases(golden, "smp_")
    for tag, c in cs.items():
        L, C, D, cap, wl = (int(x) for x in c["cfg"])
        F = c["feature"].shape[1]
        pred, loss, feat, grads, net = run_batch(gf, [(c["adj"], c["feature"])], c["target"], c["params"], L, C, F, D, cap, bool(wl), fused=fused,
                                                 coulomb=[c["coulomb"]] if "coulomb" in c else None,
                                                 wiring=(int(c["wiring"][0]), bool(c["wiring"][1])) if "wiring" in c else (18, False))
        V = len(c["adj"])
        for l in range(L + 1):
            for v in range(V):
                n = c["phi"][l, v, 0]
                assert net.receptive_field(0, l, v) == list(c["phi"][l, v, 1:1 + n]), tag
        note("goldens_fused" if fused else "goldens_unfused", feat=rel_err(feat[0], c["graph_feature"]), pred=rel_err(pred, c["predict"]),
             loss=rel_err(loss, c["loss"]), grads=rel_err(grads, c["grads"]))
        assert rel_err(feat[0], c["graph_feature"]) <= TOL_FWD, tag
        assert rel_err(pred, c["predict"]) <= TOL_FWD, tag
        assert rel_err(loss, c["loss"]) <= 2 * TOL_FWD, tag
        assert rel_err(grads, c["grads"]) <= TOL_GRAD, tag


def test_reference_goldens_with_host_built_level_tables(gf, golden, monkeypatch):
    """The same goldens with the rows-sized level tables built by the host (GF_PREP_DEVICE_TABLES=0).  The C = 10 models are computed at
    32 padded channels, whose weight-gradient kernel then takes its column exponents from exact column maxima over T and dO (the device
    builder's statistics words are absent): maxima over ALL of T, so the structurally absent blocks -- which nobody writes while every
    reader masks them -- must have their zeros first.  (Found in round 4's second session: gradients off by 0.5 on the toy molecules.)"""
    monkeypatch.setenv("GF_PREP_DEVICE_TABLES", "0")
    test_reference_goldens_one_molecule_at_a_time(gf, golden, True)


def test_batch_equals_sum_of_molecules(gf, golden):
    """The four toy molecules of tests/test_SMP_omega.cpp as ONE batch: per-molecule outputs unchanged, gradient = sum
    (what sum_gradients accumulates in BatchLearn, SMP_omega.h:808-820)."""
    cs = {k: v for k, v in golden_cases(golden, "smp_toy").items()}
    names = sorted(cs)
    c0 = cs[names[0]]
    L, C, D, cap, wl = (int(x) for x in c0["cfg"])
    params = c0["params"]  # any one parameter set for all four
    from oracle import smp_oracle
    mols = [(cs[n]["adj"], cs[n]["feature"]) for n in names]
    tg = np.array([cs[n]["target"][0] for n in names])
    pred, loss, feat, grads, _ = run_batch(gf, mols, tg, params, L, C, 4, D, cap)
    ref = [smp_oracle.run(cs[n]["adj"], cs[n]["feature"], float(cs[n]["target"][0]), params.astype(np.float64), L, C, D, cap)
           for n in names]
    assert rel_err(pred, np.array([r["predict"] for r in ref])) <= TOL_FWD
    assert rel_err(loss, np.array([r["loss"] for r in ref])) <= 2 * TOL_FWD
    assert rel_err(feat, np.stack([r["graph_feature"] for r in ref])) <= TOL_FWD
    assert rel_err(grads, sum(r["grads"] for r in ref)) <= TOL_GRAD


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("C,L,cap", [(8, 2, 6), (16, 3, 8), (20, 2, 8), (32, 3, 10), (64, 2, 12), (128, 2, 8)])
def test_synthetic_batch_vs_oracle(gf, C, L, cap, fused):
    from oracle import smp_oracle
    F, D = 5, 2
    mols, tg = [], []
    for seed in range(6):
        adj, feat, t = synthetic_molecule(100 + seed, nV=3 + 2 * seed)
        mols.append((adj, feat))
        tg.append(t)
    params = smp_params(C, F, D, L, 7)
    pred, loss, feat, grads, _ = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap, fused=fused)
    ref = [smp_oracle.run(a, f, t, params, L, C, D, cap) for (a, f), t in zip(mols, tg)]
    note("synthetic_vs_oracle", pred=rel_err(pred, np.array([r["predict"] for r in ref])),
         feat=rel_err(feat, np.stack([r["graph_feature"] for r in ref])), grads=rel_err(grads, sum(r["grads"] for r in ref)))
    assert rel_err(pred, np.array([r["predict"] for r in ref])) <= TOL_FWD
    assert rel_err(feat, np.stack([r["graph_feature"] for r in ref])) <= TOL_FWD
    assert rel_err(grads, sum(r["grads"] for r in ref)) <= TOL_GRAD


def _headline():
    import os
    from test_smp_cpu import activation_digest, headline_golden
    c, cfg, params = headline_golden()
    return c, cfg, params, activation_digest


# LeakyReLU's kink.  At this size a level holds 2.4e5 pre-activations spread over +-3.6e3, and a handful of them (4 at level 3
# of the fixture, |z| = 7e-5 = 2e-8 of the level's largest) lie closer to 0 than fp32 can resolve the sum that produced them
# (the forward itself agrees with fp64 to 8e-7 of the level's largest value).  An fp32 implementation -- any summation order --
# lands such an element on either side of 0 and takes the other slope, 1 vs 0.01: either is a valid subgradient at the kink,
# but the two parameter gradients differ by a visible 1e-4..1e-3 of the largest gradient.  So the gradient is held to 1e-5
# against the fp64 C port (bit-identical to the real reference on this fixture, tests/test_smp_cpu.py) evaluated with the
# slope the device saw at elements within KINK_TOL of the kink and ITS OWN slope everywhere else; a device sign that differs
# from fp64 outside that tolerance is a forward error and fails the test (n_conflict).  Against the real reference's own
# gradient (fp64's choice at every kink) the bound is the kink-limited KINK_GRAD.
KINK_TOL, KINK_GRAD = 1e-7, 2e-3


def kink_aware_reference(c, params, cfg, net, mol):
    from oracle import pyoracle
    L, C, D, cap = cfg
    V = len(c["adj"])
    signs = [[net.activation(mol, l, v) for v in range(V)] for l in range(L + 1)]
    o = pyoracle.port_smp_molecule(c["adj"], c["feature"], float(c["target"][0]), params, L, C, D, cap, True, ext_sign=signs,
                                   kink_tol=KINK_TOL)
    assert o["n_conflict"] == 0, "%d activations have the wrong sign outside the kink tolerance" % o["n_conflict"]
    assert o["n_override"] <= 1e-4 * o["n_elements"], (o["n_override"], o["n_elements"])
    return o


def slope_flips(n1, n0, mols, L):
    """LeakyReLU slopes two implementations of the same batch took differently: (count, largest |f| among them relative to the
    level's largest activation).  f = LeakyReLU(z) has the sign of z, so the signs of the stored activations ARE the slopes."""
    flips, near = 0, 0.0
    for l in range(L + 1):
        diffs, top = [], 0.0
        for m, (adj, _) in enumerate(mols):
            for v in range(len(adj)):
                a1, a0 = n1.activation(m, l, v), n0.activation(m, l, v)
                top = max(top, float(np.abs(a0).max()))
                d = (a1 > 0) != (a0 > 0)
                if d.any():
                    flips += int(d.sum())
                    diffs.append(max(float(np.abs(a1[d]).max()), float(np.abs(a0[d]).max())))
        if diffs:
            near = max(near, max(diffs) / top)
    return flips, near


TOL_SELF = 2e-5   # two fp32 implementations of the same sums, each within TOL_GRAD of the truth


def assert_grads_agree_kink_aware(name, g1, g0, n1, n0, mols, L):
    """Two implementations of the same batch (round-3 review, weak #2: these comparisons used the kink-limited 2e-3 bound
    unconditionally, which would pass a real 1e-4 bug).  The gradients must agree to TOL_SELF unless the two forwards took a
    LeakyReLU slope differently; then every such pre-activation must sit within KINK_TOL of the kink (relative to its level's
    largest activation; a flipped negative value is stored times 0.01, hence the factor) and only then does the kink-limited
    bound apply."""
    e = rel_err(g1, g0)
    flips, near = slope_flips(n1, n0, mols, L)
    note(name, grads=e, slope_flips=flips)
    print("%s: gradients differ by %.2e, %d slopes taken differently (|f| <= %.1e of the level's largest)" % (name, e, flips, near))
    if flips == 0:
        assert e <= TOL_SELF, e
    else:
        assert near <= KINK_TOL and e <= KINK_GRAD, (flips, near, e)


@pytest.mark.parametrize("fused", [True, False])
def test_headline_shape_against_the_real_reference(gf, fused):
    """BASELINE configs[2]'s own shape pinned to the REAL reference (tests/golden/smp_headline.npz, generated by
    tests/golden/make_golden.py from GraphFlow/SMP_omega.h:584-693): a 29-atom molecule at L = 3, cap 29, C = 64, F = 5, D = 5
    (fields up to 20: the s <= 32 size classes of tables-forward, the 16-accumulator class of the folded gather, ragged 32-row
    panels).  Receptive fields, reduced adjacencies, every level's activations (digests for all vertices, full tensors for
    levels 0-1 and four picked nodes of levels 2-3), Feature(), predict, loss to 1e-5 of the real reference; all 223,360
    parameter gradients to 1e-5 of the fp64 port with the device's slope at the LeakyReLU kinks (see KINK_TOL above)."""
    c, (L, C, D, cap), params, digest = _headline()
    F = c["feature"].shape[1]
    pred, loss, feat, grads, net = run_batch(gf, [(c["adj"], c["feature"])], c["target"], params, L, C, F, D, cap, True, fused=fused)
    V = len(c["adj"])
    worst_act = 0.0
    for l in range(L + 1):
        scale = max(np.abs(c["act_digest"][l]).max(), 1.0)
        for v in range(V):
            n = int(c["phi"][l, v, 0])
            assert net.receptive_field(0, l, v) == list(c["phi"][l, v, 1:1 + n]), (l, v)
            if l > 0:
                assert np.array_equal(net.reduced_adjacency(0, l, v), c["reduced_adj"][l, v, :n, :n].astype(np.float32)), (l, v)
            d = digest(net.activation(0, l, v).astype(np.float64))
            worst_act = max(worst_act, np.abs(d - c["act_digest"][l, v]).max() / scale)
    for l in (0, 1):
        got = np.concatenate([net.activation(0, l, v).ravel() for v in range(V)]).astype(np.float64)
        worst_act = max(worst_act, rel_err(got, c["act_level%d" % l].astype(np.float64)))
    for l, v in c["act_picks"]:
        worst_act = max(worst_act, rel_err(net.activation(0, int(l), int(v)).astype(np.float64), c["act_l%d_v%d" % (l, v)].astype(np.float64)))
    o = kink_aware_reference(c, params, (L, C, D, cap), net, 0)
    e = dict(act=worst_act, feat=rel_err(feat[0], c["graph_feature"]), pred=rel_err(pred, c["predict"]), loss=rel_err(loss, c["loss"]),
             grads=rel_err(grads, o["grads"]), grads_vs_fp64_kinks=rel_err(grads, c["grads"].astype(np.float64)))
    note("headline_fused" if fused else "headline_unfused", **e)
    print("headline shape (%s): max rel err %s; %d of %d slopes taken from the device at the kink" %
          ("fused" if fused else "op-by-op", {k: "%.2e" % v for k, v in e.items()}, o["n_override"], o["n_elements"]))
    assert e["act"] <= TOL_FWD and e["feat"] <= TOL_FWD and e["pred"] <= TOL_FWD and e["loss"] <= 2 * TOL_FWD
    assert e["grads"] <= TOL_GRAD
    assert e["grads_vs_fp64_kinks"] <= KINK_GRAD


# fused variants of the level at C = 64: what runs the block products.  Each gets its own LeakyReLU-kink budget (needed_tol = the
# largest |z| / max|z| of its level among the pre-activations whose sign the device took differently from fp64).
VARIANTS = {
    "split": {},                                                   # default: f16 matrix pipe, two-half operands
    "fp32_pipe": {"GF_SMP_SPLIT": "0"},                            # row-panel kernels on the fp32 matrix pipe
    "tiled_gemm": {"GF_SMP_ROWPANEL": "0"},                        # grouped tiled fp32 GEMMs, three-block projection (what every other channel count runs)
    "op_by_op": None,                                              # the unfused pipeline
}
# measured on the fixture (profiles/r04_parity_margins.txt; r03: the same): NO variant flips a slope -- split operands, fp32-pipe row
# panels, tiled fp32 GEMMs and the op-by-op pipeline all take fp64's side at every one of the 339,712 pre-activations, and their raw
# gradients are within 3e-7 of the real reference's (the test then asserts the raw 1e-5 bar, no kink involved).  Round 2's row-panel
# kernels flipped 2 / 4 slopes, all at |z| <= 1.93e-8 of their level's largest pre-activation (a third of fp32's epsilon of that
# value): summation order, not operand width.  KINK_TOL_VARIANT (1e-7, 1.6 x fp32 epsilon) is what a flip may be away from 0.
KINK_TOL_VARIANT = {"split": 1e-7, "fp32_pipe": 1e-7, "tiled_gemm": 1e-7, "op_by_op": 1e-7}


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_headline_slope_flip_accounting(gf, monkeypatch, variant):
    """For each implementation of the level: how many LeakyReLU slopes of the headline fixture the device takes differently from
    the fp64 reference, how close to the kink those pre-activations are (needed_tol), the RAW gradient error against the REAL
    reference's gradient, and the error against the fp64 port evaluated with the device's slopes at exactly those elements
    (kink_tol = needed_tol: nothing else is overridden).  Asserts needed_tol <= KINK_TOL_VARIANT and the kink-aware gradient to 1e-5."""
    from oracle import pyoracle
    c, (L, C, D, cap), params, _ = _headline()
    F = c["feature"].shape[1]
    env = VARIANTS[variant]
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    pred, loss, feat, grads, net = run_batch(gf, [(c["adj"], c["feature"])], c["target"], params, L, C, F, D, cap, True, fused=env is not None)
    V = len(c["adj"])
    ref = pyoracle.port_smp_molecule(c["adj"], c["feature"], float(c["target"][0]), params, L, C, D, cap, True, want_acts=True)
    dev_acts = [[net.activation(0, l, v) for v in range(V)] for l in range(L + 1)]
    flips, needed = 0, 0.0
    for l in range(L + 1):
        # f = LeakyReLU(z): |z| = |f| for f > 0, |f| / 0.01 for f < 0
        zabs = [np.where(a > 0, a, -a / 0.01) for a in ref["acts"][l]]
        zmax = max(float(z.max()) for z in zabs)
        for v in range(V):
            diff = (np.asarray(dev_acts[l][v]) > 0) != (ref["acts"][l][v] > 0)
            flips += int(diff.sum())
            if diff.any():
                needed = max(needed, float(zabs[v][diff].max()) / zmax)
    raw = rel_err(grads, c["grads"].astype(np.float64))
    tol = needed * (1 + 1e-9) if flips else 0.0
    o = pyoracle.port_smp_molecule(c["adj"], c["feature"], float(c["target"][0]), params, L, C, D, cap, True, ext_sign=dev_acts, kink_tol=tol)
    assert o["n_conflict"] == 0 and o["n_override"] == flips, (o["n_conflict"], o["n_override"], flips)
    aware = rel_err(grads, o["grads"])
    note("headline_flips_" + variant, flips=flips, needed_kink_tol=needed, grads_raw_vs_real_reference=raw, grads_kink_aware=aware,
         feat=rel_err(feat[0], c["graph_feature"]))
    print("headline fixture, %s: %d of %d slopes flipped (|z| <= %.2e of the level max); gradient vs real reference raw %.2e, "
          "with the device's slopes at those elements %.2e" % (variant, flips, o["n_elements"], needed, raw, aware))
    assert needed <= KINK_TOL_VARIANT[variant], (variant, flips, needed)
    assert aware <= TOL_GRAD
    if flips == 0:
        assert raw <= TOL_GRAD   # no kink involved: the raw comparison IS the 1e-5 bar


def test_headline_molecule_inside_a_batch(gf):
    """The same molecule at position 17 of a 64-molecule batch (its nodes interleaved with the others' in every size class):
    same Feature / predict / activations as alone, and the batch gradient minus the gradient of the other 63 run without it
    = this molecule's gradient (fp64 port, device slopes at the kinks)."""
    c, (L, C, D, cap), params, digest = _headline()
    F = c["feature"].shape[1]
    others, tg = [], []
    for seed in range(63):
        a, f, t = synthetic_molecule(7000 + seed)
        others.append((a, f))
        tg.append(t)
    mols = others[:17] + [(c["adj"], c["feature"])] + others[17:]
    tgs = np.array(tg[:17] + [float(c["target"][0])] + tg[17:])
    pred, loss, feat, grads, net = run_batch(gf, mols, tgs, params, L, C, F, D, cap)
    l, v = (int(x) for x in c["act_picks"][2])
    e = dict(feat=rel_err(feat[17], c["graph_feature"]), pred=rel_err(pred[17:18], c["predict"]),
             act=rel_err(net.activation(17, l, v).astype(np.float64), c["act_l%d_v%d" % (l, v)].astype(np.float64)))
    o = kink_aware_reference(c, params, (L, C, D, cap), net, 17)
    p_o, _, f_o, g_others, _ = run_batch(gf, others, np.array(tg), params, L, C, F, D, cap)
    # a molecule's forward does not depend on its batch mates: the other 63 took the same slopes in both runs
    assert np.array_equal(np.delete(pred, 17), p_o) and np.array_equal(np.delete(feat, 17, axis=0), f_o)
    e["grads"] = rel_err(grads - g_others, o["grads"])
    note("headline_in_batch", **e)
    print("headline molecule inside a batch: %s" % {k: "%.2e" % x for k, x in e.items()})
    assert e["feat"] <= TOL_FWD and e["pred"] <= TOL_FWD and e["act"] <= TOL_FWD
    assert e["grads"] <= 2 * TOL_GRAD   # a difference of two fp32 batch sums, each within TOL_GRAD of its own truth


def test_backward_needs_a_forward_with_targets(gf):
    """A Predict / Feature forward (targets NULL) leaves no loss gradient: backward after it is refused, not run against 0."""
    from graphflow_amd.smp import SMPOmega
    net = SMPOmega(2, 8, 5, 2, 6, True)
    net.prepare([synthetic_molecule(1, nV=6)[:2]])
    p = dev(smp_params(8, 5, 2, 2, 1))
    g = torch.empty(net.n_params, device="cuda")
    net.forward(p)
    with pytest.raises(Exception, match="no targets"):
        net.backward(p, g)
    net.forward(p, dev(np.array([6.0])))
    net.backward(p, g)
    with pytest.raises(TypeError):
        net.backward(p, g[:-1])
    with pytest.raises(TypeError):
        net.forward(p.double())


def test_a_second_backward_after_op_by_op_levels_is_refused(gf):
    """An op-by-op level's reverse sweep overwrites its Q with dQ (MatMul::backward's first operand in place): a second gf_smp_backward
    without a new forward would differentiate garbage.  It is an error, not a wrong gradient; the fused levels keep their forward state
    and may be swept again (backward(accumulate) after backward)."""
    from graphflow_amd.smp import SMPOmega
    F, D, C, L, cap = 5, 2, 8, 2, 8
    mols = [synthetic_molecule(s, nV=7)[:2] for s in range(3)]
    p = dev(smp_params(C, F, D, L, 1))
    t = dev(np.arange(3))
    for fused in (False, True):
        net = SMPOmega(L, C, F, D, cap)
        net.set_fused(fused)
        net.prepare(mols)
        net.forward(p, t)
        g = torch.empty(net.n_params, device="cuda")
        net.backward(p, g)
        if fused:
            g2 = g.clone()
            net.backward(p, g2, accumulate=True)
            assert float((g2 - 2 * g).abs().max()) <= 1e-5 * float(g.abs().max())
        else:
            with pytest.raises(Exception, match="second reverse sweep"):
                net.backward(p, g)
            net.forward(p, t)
            net.backward(p, g)   # (fine again after a forward)
        net.close()


def test_vertex_permutation_invariance(gf):
    """tests/test_graph_permutation_invariant.cpp: Feature() of a graph and of a vertex-permuted copy agree."""
    F, D, C, L, cap = 5, 3, 16, 2, 10
    adj, feat, _ = synthetic_molecule(77, nV=14)
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(adj))
    adj2, feat2 = adj[np.ix_(perm, perm)], feat[perm]
    params = smp_params(C, F, D, L, 5)
    _, _, f, _, _ = run_batch(gf, [(adj, feat), (adj2, feat2)], np.zeros(2), params, L, C, F, D, cap)
    assert rel_err(f[0], f[1]) <= TOL_FWD


def test_forward_is_deterministic_and_backward_accumulates(gf):
    from graphflow_amd.smp import SMPOmega
    F, D, C, L, cap = 5, 2, 8, 2, 8
    mols = [synthetic_molecule(s, nV=9)[:2] for s in range(4)]
    net = SMPOmega(L, C, F, D, cap)
    net.prepare(mols)
    p = dev(smp_params(C, F, D, L, 1))
    t = dev(np.arange(4))
    a = net.forward(p, t)[0].clone()
    b = net.forward(p, t)[0].clone()
    assert torch.equal(a, b)
    g1 = torch.empty(net.n_params, device="cuda")
    net.backward(p, g1)
    g2 = g1.clone()
    net.forward(p, t)
    net.backward(p, g2, accumulate=True)
    assert float((g2 - 2 * g1).abs().max()) <= 1e-5 * float(g1.abs().max())


def test_fused_and_op_by_op_levels_agree_at_scale(gf):
    """64 channels, 3 levels, cap 29, 48 molecules up to 29 atoms: the fused level path against the op-by-op pipeline."""
    F, D, C, L, cap = 5, 5, 64, 3, 29
    mols, tg = [], []
    for seed in range(48):
        adj, feat, t = synthetic_molecule(500 + seed)
        mols.append((adj, feat))
        tg.append(t)
    params = smp_params(C, F, D, L, 9)
    p1, l1, f1, g1, _ = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap, fused=True)
    p0, l0, f0, g0, _ = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap, fused=False)
    assert rel_err(p1, p0) <= TOL_FWD and rel_err(f1, f0) <= TOL_FWD
    assert rel_err(g1, g0) <= TOL_GRAD


def test_folded_backward_gather_equals_the_two_kernel_path(gf, monkeypatch):
    """Fused levels evaluate dP inside the consumer gather (default: one launch per level, a wave per (source, 64-lane chunk) with
    scalar-loaded per-consumer records) or write it with tables-backward and gather it afterwards (GF_SMP_BWD_GATHER=0, also the
    route for receptive fields > 32): same expression (the gather adds the two diagonal terms of a row after the consumers instead
    of per consumer).  29-atom molecules reach the 12- and 16-accumulator paths and the two-half items of sources above 16."""
    F, D, C, L, cap = 5, 5, 64, 3, 29
    mols, tg = [], []
    for seed in range(24):
        adj, feat, t = synthetic_molecule(900 + seed, nV=29 if seed % 3 == 0 else None)
        mols.append((adj, feat))
        tg.append(t)
    params = smp_params(C, F, D, L, 5)
    g1 = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap)[3]
    monkeypatch.setenv("GF_SMP_BWD_GATHER", "0")   # read when the handle is created
    g0 = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap)[3]
    assert rel_err(g1, g0) <= 1e-6


def test_c64_level_kernels_equal_the_tiled_gemm_path(gf, monkeypatch):
    """At C = 64 the block products of a fused level run as dedicated kernels (weights resident in LDS with the rows in registers,
    compact two-block projection U = Z + Z'^T, combine-forward on row panels; output-stationary weight gradients).  GF_SMP_ROWPANEL=0
    selects what every other channel count runs: the grouped tiled GEMM launches, the three-block projection and the workgroup-per-
    (node, four x) combine -- same products, different summation order.  29-atom molecules: one-group panels (s >= 17), ragged last
    panels, up to eight groups per panel."""
    F, D, C, L, cap = 5, 5, 64, 3, 29
    mols, tg = [], []
    for seed in range(40):   # 40 molecules: the row counts are not multiples of the 32-row panels / 32-row slices
        adj, feat, t = synthetic_molecule(1300 + seed, nV=29 if seed % 4 == 0 else None)
        mols.append((adj, feat))
        tg.append(t)
    params = smp_params(C, F, D, L, 6)
    p1, _, f1, g1, n1 = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap)
    a1 = [n1.activation(0, l, 0) for l in (1, 2, 3)]
    monkeypatch.setenv("GF_SMP_ROWPANEL", "0")
    p0, _, f0, g0, n0 = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap)
    a0 = [n0.activation(0, l, 0) for l in (1, 2, 3)]
    assert not np.array_equal(f1, f0)   # (the switch switches something)
    for x, y in zip(a1, a0):
        assert rel_err(x.astype(np.float64), y.astype(np.float64)) <= 2e-6
    note("c64_kernels_vs_tiled_gemms", pred=rel_err(p1, p0), feat=rel_err(f1, f0))
    assert rel_err(p1, p0) <= 2e-6 and rel_err(f1, f0) <= 2e-6
    assert_grads_agree_kink_aware("c64_kernels_vs_tiled_gemms", g1, g0, n1, n0, mols, L)


def test_c32_row_panel_kernels_equal_the_tiled_gemm_path(gf, monkeypatch):
    """Round 4: at C = 32 the block products of a fused level run on the same split-operand row-panel kernels as at C = 64 (32 x 32
    blocks: one column half, two k-chunks per lane; compact two-block projection with the transposed-row gather; structural zeros
    read from the zero page) and the weight gradients on smp_wgrad_direct<32> (operands loaded straight into MFMA layout, exact
    column bounds).  GF_SMP_ROWPANEL=0 selects the grouped tiled fp32 GEMMs with the three-block projection that the other channel
    counts run.  29-atom molecules: ragged last panels and slices, rows without data in every block."""
    F, D, C, L, cap = 5, 5, 32, 3, 29
    mols, tg = [], []
    for seed in range(40):
        adj, feat, t = synthetic_molecule(1700 + seed, nV=29 if seed % 4 == 0 else None)
        mols.append((adj, feat))
        tg.append(t)
    params = smp_params(C, F, D, L, 6)
    p1, _, f1, g1, n1 = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap)
    a1 = [n1.activation(0, l, 0) for l in (1, 2, 3)]
    monkeypatch.setenv("GF_SMP_ROWPANEL", "0")
    p0, _, f0, g0, n0 = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap)
    a0 = [n0.activation(0, l, 0) for l in (1, 2, 3)]
    assert not np.array_equal(f1, f0)   # (the switch switches something)
    for x, y in zip(a1, a0):
        assert rel_err(x.astype(np.float64), y.astype(np.float64)) <= 2e-6
    note("c32_kernels_vs_tiled_gemms", pred=rel_err(p1, p0), feat=rel_err(f1, f0))
    assert rel_err(p1, p0) <= 2e-6 and rel_err(f1, f0) <= 2e-6
    assert_grads_agree_kink_aware("c32_kernels_vs_tiled_gemms", g1, g0, n1, n0, mols, L)
    # ... and with the structural-zero masking off (dense reads of the same tables)
    monkeypatch.delenv("GF_SMP_ROWPANEL")
    monkeypatch.setenv("GF_SMP_MASK_ZEROS", "0")
    p2, _, f2, g2, n2 = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap)
    assert rel_err(p2, p1) <= 2e-6 and rel_err(f2, f1) <= 2e-6
    assert_grads_agree_kink_aware("c32_masked_vs_dense", g2, g1, n2, n1, mols, L)


@pytest.mark.parametrize("C,wiring", [(10, (18, False)), (12, (18, False)), (20, (18, True)), (40, (18, False)), (66, (18, False))])
def test_padded_channels_equal_the_unpadded_model(gf, monkeypatch, C, wiring):
    """Round 4: a model whose nChanels is not 32 / 64 is COMPUTED with its channels zero-padded to 32 / 64 (above 64: to a multiple of
    4) so that it runs the dedicated level kernels (gf_smp_create).  The padded weights are zero and LeakyReLU(0) = 0, so the real
    channels are unchanged; parameters, gradients, features and activations keep the caller's layout at the C ABI.  Compared with
    GF_SMP_PAD_CHANNELS=0 (the level computed at nChanels: the generic fused level for C % 4 == 0, op by op otherwise) and with the
    fp64 oracle; SMP_2D_ver8's [C, 18 C] weights (custom_matmul) go through the same padding."""
    from oracle import smp_oracle
    F, D, L, cap = 5, 2, 3, 12
    mols, tg = [], []
    for seed in range(24):
        adj, feat, t = synthetic_molecule(2300 + seed, nV=4 + seed % 9)
        mols.append((adj, feat))
        tg.append(t)
    params = smp_params(C, F, D, L, 9)
    p1, l1, f1, g1, n1 = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap, wiring=wiring)
    assert f1.shape == (len(mols), C) and g1.shape == (n1.n_params,) == params.shape
    a1 = [n1.activation(3, l, 1) for l in (0, 1, 2, 3)]
    # backward(accumulate): the caller's gradient buffer gets += the cropped gradient
    g_acc = torch.full((n1.n_params,), 1.0, device="cuda")
    n1.backward(dev(params), g_acc, accumulate=True)
    assert rel_err(g_acc.cpu().numpy().astype(np.float64) - 1.0, g1) <= 1e-6
    monkeypatch.setenv("GF_SMP_PAD_CHANNELS", "0")
    p0, l0, f0, g0, n0 = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap, wiring=wiring)
    a0 = [n0.activation(3, l, 1) for l in (0, 1, 2, 3)]
    for x, y in zip(a1, a0):
        assert x.shape == y.shape and x.shape[2] == C
        assert rel_err(x.astype(np.float64), y.astype(np.float64)) <= 2e-6
    note("padded_vs_unpadded", pred=rel_err(p1, p0), feat=rel_err(f1, f0))
    assert rel_err(p1, p0) <= 2e-6 and rel_err(f1, f0) <= 2e-6 and rel_err(l1, l0) <= 4e-6
    assert_grads_agree_kink_aware("padded_vs_unpadded", g1, g0, n1, n0, mols, L)
    if not wiring[1]:
        ref = [smp_oracle.run(a, f, t, params, L, C, D, cap) for (a, f), t in zip(mols, tg)]
        assert rel_err(p1, np.array([r["predict"] for r in ref])) <= TOL_FWD
        assert rel_err(f1, np.stack([r["graph_feature"] for r in ref])) <= TOL_FWD
        assert rel_err(g1, sum(r["grads"] for r in ref)) <= TOL_GRAD


@pytest.mark.parametrize("C", [32, 64])
def test_a_dead_channel_has_zero_gradients_not_nan(gf, monkeypatch, C):
    """A channel whose weights are all zero stays zero through LeakyReLU: its columns of the weight-gradient operands are entirely
    zero.  The split-operand weight-gradient kernels fold a row factor into the column's power-of-two scale before the multiply;
    with the scale of a zero column at its ceiling that product was inf and inf x 0 = NaN (found by the padded models, round 4)."""
    monkeypatch.setenv("GF_SMP_PAD_CHANNELS", "0")
    F, D, L, cap = 5, 2, 2, 10
    mols, tg = [], []
    for seed in range(8):
        adj, feat, t = synthetic_molecule(2500 + seed, nV=5 + seed)
        mols.append((adj, feat))
        tg.append(t)
    FD = F * (D + 1)
    params = smp_params(C, F, D, L, 4).copy()
    dead, o = [3, C - 1], C * FD
    params[:o].reshape(C, FD)[dead] = 0
    for _ in range(L):
        K = params[o:o + 18 * C * C].reshape(18, C, C)
        K[:, dead, :] = 0
        K[:, :, dead] = 0
        o += 18 * C * C
        params[o:o + C][dead] = 0
        o += C
    params[o:][dead] = 0
    _, _, _, g, _ = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap)
    assert np.isfinite(g).all()
    o = C * FD
    for _ in range(L):   # dK[k][ci][dead co] = sum over rows of Q[.., ci] dz[.., dead] = 0
        K = g[o:o + 18 * C * C].reshape(18, C, C)
        assert not K[:, :, dead].any()
        o += 18 * C * C + C


def test_split_operand_products_equal_the_fp32_products(gf, monkeypatch):
    """The C = 64 level's three block-product kernels on the f16 matrix pipe with two-half fp32 operands (default,
    smp_level_c64_split.hip) against the same products on the fp32 pipe (GF_SMP_SPLIT=0): same sums, operands carried to 22
    bits and accumulated in fp32 -- the difference stays at fp32 rounding level (and at the kink-limited bound for gradients)."""
    F, D, C, L, cap = 5, 5, 64, 3, 29
    mols, tg = [], []
    for seed in range(40):
        adj, feat, t = synthetic_molecule(1300 + seed)
        mols.append((adj, feat))
        tg.append(t)
    params = smp_params(C, F, D, L, 6)
    p1, _, f1, g1, n1 = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap)
    monkeypatch.setenv("GF_SMP_SPLIT", "0")
    p0, _, f0, g0, n0 = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap)
    assert not np.array_equal(f1, f0)   # (the switch switches something)
    note("split_vs_fp32_products", pred=rel_err(p1, p0), feat=rel_err(f1, f0))
    assert rel_err(p1, p0) <= 2e-6 and rel_err(f1, f0) <= 2e-6
    assert_grads_agree_kink_aware("split_vs_fp32_products", g1, g0, n1, n0, mols, L)


@pytest.mark.parametrize("scales", [(1e-3,), (1e3,), (1e-4, 1.0, 1e4)])
def test_split_operand_products_over_input_scales(gf, scales):
    """The split products choose a power-of-two exponent per row (forward / backward products) and per block of the level
    (weight gradients).  Inputs three or four decades away from 1, and a batch that mixes molecules eight decades apart, against
    the fp64 oracle: every molecule's prediction and feature to 1e-5 of ITS OWN magnitude, the summed gradient to 1e-5 (uniform
    scales) or to the kink-limited bound (mixed)."""
    from oracle import smp_oracle
    F, D, C, L, cap = 5, 2, 64, 2, 12
    mols, tg = [], []
    for seed in range(6):
        adj, feat, t = synthetic_molecule(100 + seed, nV=3 + 2 * seed)
        k = scales[seed % len(scales)]
        mols.append((adj, (np.asarray(feat, dtype=np.float64) * k).astype(np.float32)))
        tg.append(t * k)
    params = smp_params(C, F, D, L, 7)
    pred, loss, feat, grads, net_fused = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap)
    ref = [smp_oracle.run(a, f, t, params, L, C, D, cap) for (a, f), t in zip(mols, tg)]

    def rel(x, r):   # (util.rel_err floors the denominator at 1: here the magnitudes are the point)
        x, r = np.asarray(x, dtype=np.float64), np.asarray(r, dtype=np.float64)
        return float(np.abs(x - r).max() / np.abs(r).max())

    for i, r in enumerate(ref):   # per molecule: relative to the molecule, not to the batch
        assert rel(feat[i], r["graph_feature"]) <= TOL_FWD, i
        assert abs(pred[i] - r["predict"]) <= TOL_FWD * max(abs(r["predict"]), np.abs(r["graph_feature"]).max()), i
    # gradients: the kink-limited bound (KINK_TOL above) -- in the mixed batch the all-fp32 op-by-op path sits at the same 5.2e-5
    # from the fp64 oracle as this one, the fp32 fused products at 1e-6: a LeakyReLU slope, not an operand width
    g_ref = sum(r["grads"] for r in ref)
    assert rel(grads, g_ref) <= (TOL_GRAD if len(scales) == 1 else KINK_GRAD)
    if len(scales) > 1:   # ... and against the op-by-op fp32 pipeline of the same batch, slope for slope
        _, _, _, g_ob, n_ob = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap, fused=False)
        assert_grads_agree_kink_aware("mixed_scales_fused_vs_op_by_op", grads, g_ob, net_fused, n_ob, mols, L)


def test_two_handles_alternate_without_waiting_for_each_other(gf):
    """A training loop prepares batch i+1 on a second handle while the device runs step i on the first (uploads on the
    handle's own stream; recycling a handle's buffers waits for ITS last launch only).  Same gradients as one handle that
    prepares and steps in turn."""
    from graphflow_amd.smp import SMPOmega
    F, D, C, L, cap = 5, 3, 64, 2, 29
    batches = []
    for b in range(4):
        mols, tg = [], []
        for seed in range(30 + 7 * b):   # different sizes: the pools grow and get reused
            adj, feat, t = synthetic_molecule(2000 + 100 * b + seed)
            mols.append((adj, feat))
            tg.append(t)
        batches.append((mols, dev(np.array(tg))))
    p = dev(smp_params(C, F, D, L, 2))

    def loop(nh):
        nets = [SMPOmega(L, C, F, D, cap, True) for _ in range(nh)]
        out = []
        nets[0].prepare(batches[0][0])
        for it in range(8):
            cur = nets[it % nh]
            if nh == 1:
                cur.prepare(batches[it % 4][0])
            g = torch.empty(cur.n_params, device="cuda")
            cur.forward(p, batches[it % 4][1])
            cur.backward(p, g)
            if nh == 2:   # no synchronisation between the launches above and this host work
                nets[(it + 1) % 2].prepare(batches[(it + 1) % 4][0])
            out.append(g)
        torch.cuda.synchronize()
        return [x.cpu().numpy() for x in out]

    one, two = loop(1), loop(2)
    for a, b in zip(one, two):
        assert np.array_equal(a, b)


def test_loader_threads_prepare_batches_side_by_side(gf):
    """The end-to-end loop of bench.py: two host threads call gf_smp_prepare on different handles at the same time (a worker pool
    per calling thread) while the main thread launches steps on a third.  Same gradients, bit for bit, as preparing and stepping
    in turn on one thread."""
    import threading
    from graphflow_amd.smp import SMPOmega
    F, D, C, L, cap = 5, 3, 64, 2, 29
    batches = []
    for b in range(4):
        mols, tg = [], []
        for seed in range(70 + 9 * b):   # (enough molecules for the parallel sections of the preparation to use their pools)
            adj, feat, t = synthetic_molecule(4000 + 100 * b + seed)
            mols.append((adj, feat))
            tg.append(t)
        batches.append((SMPOmega.pack(mols), dev(np.array(tg))))
    p = dev(smp_params(C, F, D, L, 2))
    steps, NH, NLOAD = 12, 4, 2

    def serial():
        net = SMPOmega(L, C, F, D, cap, True)
        out = []
        for it in range(steps):
            net.prepare(batches[it % 4][0])
            g = torch.empty(net.n_params, device="cuda")
            net.forward(p, batches[it % 4][1])
            net.backward(p, g)
            out.append(g)
        torch.cuda.synchronize()
        return [x.cpu().numpy() for x in out]

    def threaded():
        nets = [SMPOmega(L, C, F, D, cap, True) for _ in range(NH)]
        ready = [threading.Semaphore(0) for _ in range(NH)]
        free = [threading.Semaphore(1) for _ in range(NH)]
        errors = []

        def loader(t):
            try:
                for it in range(t, steps, NLOAD):
                    h = it % NH
                    free[h].acquire()
                    nets[h].prepare(batches[it % 4][0])
                    ready[h].release()
            except Exception as e:   # noqa: BLE001
                errors.append(e)
                for r in ready:
                    r.release()

        ths = [threading.Thread(target=loader, args=(t,)) for t in range(NLOAD)]
        for th in ths:
            th.start()
        out = []
        for it in range(steps):
            h = it % NH
            ready[h].acquire()
            assert not errors, errors
            g = torch.empty(nets[h].n_params, device="cuda")
            nets[h].forward(p, batches[it % 4][1])
            nets[h].backward(p, g)
            out.append(g)
            free[h].release()
        for th in ths:
            th.join()
        torch.cuda.synchronize()
        return [x.cpu().numpy() for x in out]

    one, two = serial(), threaded()
    for a, b in zip(one, two):
        assert np.array_equal(a, b)


def test_fused_levels_never_take_the_promotion_buffer(gf):
    """gf_smp_device_bytes: the fused path with the folded backward gather materialises neither P nor dP, so its batch holds
    less device memory than the op-by-op pipeline of the same batch (which takes the shared [sum s^3][C] buffer on first use)."""
    F, D, C, L, cap = 5, 3, 64, 2, 29
    mols, tg = [], []
    for seed in range(16):
        adj, feat, t = synthetic_molecule(3000 + seed)
        mols.append((adj, feat))
        tg.append(t)
    params = smp_params(C, F, D, L, 4)
    fused_net = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap, fused=True)[4]
    plain_net = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap, fused=False)[4]
    used_f, res_f = fused_net.device_bytes()
    used_p, res_p = plain_net.device_bytes()
    assert 0 < used_f <= res_f and 0 < used_p <= res_p
    ppos = plain_net.level_sizes(L)[2]
    assert used_p - used_f >= 4 * ppos * C * 0.9   # at least (most of) the promotion buffer


def test_cfg3_full_size_properties(gf):
    """BASELINE configs[2] at full size (3 levels, C = 64, cap 29, 1024 synthetic QM9-size molecules), where the oracle is
    out of reach: properties that do not depend on the size.  (1) Molecules are independent: a molecule's prediction does not
    depend on its batch mates or on its position in the batch.  (2) The gradient buffer is the SUM over the batch: two
    half batches add up to the full one.  (3) Run-to-run bit reproducibility."""
    F, D, C, L, cap = 5, 5, 64, 3, 29
    mols, tg = [], []
    for seed in range(1024):
        adj, feat, t = synthetic_molecule(seed)
        mols.append((adj, feat))
        tg.append(t)
    tg = np.array(tg)
    params = smp_params(C, F, D, L, 1)
    p_all, _, f_all, g_all, _ = run_batch(gf, mols, tg, params, L, C, F, D, cap)
    p_again, _, _, g_again, _ = run_batch(gf, mols, tg, params, L, C, F, D, cap)
    assert np.array_equal(p_all, p_again) and np.array_equal(g_all, g_again)
    h = 512
    p_a, _, f_a, g_a, _ = run_batch(gf, mols[:h], tg[:h], params, L, C, F, D, cap)
    p_b, _, f_b, g_b, _ = run_batch(gf, mols[h:], tg[h:], params, L, C, F, D, cap)
    assert rel_err(np.concatenate([p_a, p_b]), p_all) <= 1e-6
    assert rel_err(np.concatenate([f_a, f_b]), f_all) <= 1e-6
    assert rel_err(g_a + g_b, g_all) <= 1e-5            # different split-K ranges / summation order, same sum
    perm = np.random.default_rng(0).permutation(1024)
    p_perm = run_batch(gf, [mols[i] for i in perm], tg[perm], params, L, C, F, D, cap)[0]
    assert rel_err(p_perm, p_all[perm]) <= 1e-6


def test_cfg3_full_size_spot_check_against_the_port(gf):
    """Round-3 review, weak #2: batch-scale coverage against the oracle stopped at 64 molecules.  Here the FULL cfg3 batch (1024
    molecules) runs once and eight of its molecules, picked at random, are held to the fp64 C port of the reference
    (oracle/smp_port.c, bit-identical to the real reference on the goldens): prediction, loss, Feature() and the level-3 activation
    of one vertex to 1e-5 INSIDE the full batch; and the eight as a batch of their own: the summed gradient to 1e-5 against the
    port evaluated with the device's slopes inside KINK_TOL of the kink (a sign that differs outside it fails)."""
    from oracle import pyoracle
    F, D, C, L, cap = 5, 5, 64, 3, 29
    mols, tg = [], []
    for seed in range(1024):
        adj, feat, t = synthetic_molecule(seed)
        mols.append((adj, feat))
        tg.append(t)
    tg = np.array(tg)
    params = smp_params(C, F, D, L, 1)
    pred, loss, feat, _, net = run_batch(gf, mols, tg, params, L, C, F, D, cap)
    picks = sorted(int(i) for i in np.random.default_rng(4).choice(1024, 8, replace=False))
    refs, worst = [], dict(pred=0.0, loss=0.0, feat=0.0, act=0.0)
    for i in picks:
        adj, ft = mols[i]
        V = len(adj)
        signs = [[net.activation(i, l, v) for v in range(V)] for l in range(L + 1)]
        o = pyoracle.port_smp_molecule(adj, ft, float(tg[i]), params, L, C, D, cap, True, ext_sign=signs, kink_tol=KINK_TOL, want_acts=True)
        assert o["n_conflict"] == 0, (i, o["n_conflict"])
        refs.append(o)
        # (the prediction is an inner product of Feature() with W: its error is held relative to the larger of |predict| and the largest
        #  feature, as in test_split_operand_products_over_input_scales -- eight single numbers have no batch maximum to lean on)
        pscale = max(abs(o["predict"]), float(np.abs(o["graph_feature"]).max()), 1.0)
        worst["pred"] = max(worst["pred"], abs(float(pred[i]) - o["predict"]) / pscale)
        lscale = max(abs(o["loss"]), pscale * max(abs(o["predict"] - float(tg[i])), 1.0), 1.0)   # d loss = (y - t) d y
        worst["loss"] = max(worst["loss"], abs(float(loss[i]) - o["loss"]) / lscale)
        worst["feat"] = max(worst["feat"], rel_err(feat[i], o["graph_feature"]))
        v = V // 2
        worst["act"] = max(worst["act"], rel_err(signs[L][v].astype(np.float64), o["acts"][L][v]))
    # the same eight as their own batch: their summed gradient against the port's (the slopes are the full batch's: a molecule's
    # forward does not depend on its batch mates -- asserted bit for bit)
    sub = [mols[i] for i in picks]
    p8, _, f8, g8, _ = run_batch(gf, sub, tg[picks], params, L, C, F, D, cap)
    assert np.array_equal(p8, pred[picks]) and np.array_equal(f8, feat[picks])
    worst["grads"] = rel_err(g8, sum(o["grads"] for o in refs))
    note("cfg3_full_size_spot_check", **worst)
    print("cfg3 full-size spot check (molecules %s): %s; %d slopes taken from the device" %
          (picks, {k: "%.2e" % x for k, x in worst.items()}, sum(o["n_override"] for o in refs)))
    assert worst["pred"] <= TOL_FWD and worst["feat"] <= TOL_FWD and worst["act"] <= TOL_FWD and worst["loss"] <= 2 * TOL_FWD
    assert worst["grads"] <= TOL_GRAD


def _cfg3_batch(n=1024, first=0):
    mols, tg = [], []
    for seed in range(first, first + n):
        adj, feat, t = synthetic_molecule(seed)
        mols.append((adj, feat))
        tg.append(t)
    return mols, np.array(tg)


@pytest.mark.parametrize("pick,C", [("random", 64), ("largest", 64), ("smallest", 64), ("largest", 32), ("random", 10)])
def test_cfg3_full_batch_gradient_of_picked_molecules_against_the_port(gf, pick, C):
    """Round-5 review, weak #1: the 1024-molecule backward (its own panel packing, split-K ranges and size-class mixes) had no
    oracle-side check -- and a store-from-register-0 compiler bug once passed the 8-32-molecule batch gradient tests by cancellation.
    In-batch gradient ISOLATION: the loss is (y - t)^2 / 2, so with the targets of all but eight molecules set to the device's own
    predictions (bit-reproducible: dL/dy = 0 exactly there) the summed gradient gf_smp_backward returns IS those eight molecules'
    gradient, computed inside the full batch's panels and size classes.  Held to the fp64 port of the reference (oracle/smp_port.c) at
    1e-5, kink-aware as everywhere (slopes from the full batch's activations inside KINK_TOL, a sign that differs outside it fails).
    Three disjoint picks: eight at random, the eight largest molecules, the eight smallest.  SMP_omega.h:808-820 (sum of per-molecule
    gradients)."""
    from graphflow_amd.smp import SMPOmega
    from oracle import pyoracle
    F, D, L, cap = 5, 5, 3, 29   # (C = 32: the 32-channel kernel family; C = 10: the reference's own channel count, computed at 16 padded channels)
    mols, tg = _cfg3_batch()
    params = smp_params(C, F, D, L, 1)
    nv = np.array([len(a) for a, _ in mols])
    order = np.argsort(nv, kind="stable")
    largest, smallest = [int(i) for i in order[-8:]], [int(i) for i in order[:8]]
    rest = [int(i) for i in np.random.default_rng(6).permutation(1024) if int(i) not in largest and int(i) not in smallest]
    picks = sorted({"random": rest[:8], "largest": largest, "smallest": smallest}[pick])
    net = SMPOmega(L, C, F, D, cap, True)
    net.prepare(mols)
    p = dev(params)
    pred0, _, _ = net.forward(p, dev(tg))
    t2 = pred0.clone()                                   # dL/dy = y - t = 0 for everybody ...
    t2[picks] = dev(tg)[picks]                            # ... but the picked eight
    pred, loss, feat = net.forward(p, t2)
    assert torch.equal(pred, pred0)                       # (bit-reproducible forward: the zeros are exact)
    g = torch.full((net.n_params,), float("nan"), device="cuda")
    net.backward(p, g)
    g = g.cpu().numpy().astype(np.float64)
    ref, n_over = 0.0, 0
    for i in picks:
        adj, ft = mols[i]
        signs = [[net.activation(i, l, v) for v in range(len(adj))] for l in range(L + 1)]
        o = pyoracle.port_smp_molecule(adj, ft, float(tg[i]), params, L, C, D, cap, True, ext_sign=signs, kink_tol=KINK_TOL)
        assert o["n_conflict"] == 0, (i, o["n_conflict"])
        assert abs(float(pred[i]) - o["predict"]) <= TOL_FWD * max(abs(o["predict"]), float(np.abs(o["graph_feature"]).max()), 1.0)
        ref = ref + o["grads"]
        n_over += o["n_override"]
    e = rel_err(g, ref)
    # per-block as well: a wrong block of a small parameter group must not hide behind the largest group's magnitude
    C2, FD = C * C, F * (D + 1)
    bounds = [0, C * FD]
    for l in range(L):
        bounds += [bounds[-1] + 18 * C2, bounds[-1] + 18 * C2 + C]
    bounds.append(bounds[-1] + C)
    eb = max(rel_err(g[a:b], ref[a:b]) for a, b in zip(bounds[:-1], bounds[1:]))
    note("cfg3_in_batch_gradient_%s_C%d" % (pick, C), grads=e, grads_per_block=eb)
    print("cfg3 in-batch gradient (C = %d, %s: molecules %s, %d..%d atoms): rel err %.2e, worst parameter block %.2e; %d slopes taken from the device"
          % (C, pick, picks, min(nv[picks]), max(nv[picks]), e, eb, n_over))
    assert np.isfinite(g).all() and e <= TOL_GRAD and eb <= 10 * TOL_GRAD
    net.close()


def test_cfg4_workload_as_eight_one_rank_shards(gf):
    """BASELINE configs[3] (cfg4: 8192 molecules sharded 1024 per GPU with an RCCL all-reduce of the parameter gradients) on ONE GPU: the
    eight shards graphflow_amd.dist.shard hands to ranks 0..7 of a strong-scaling run go, one after the other, through a context
    with a (one-rank) RCCL communicator and gf_smp_set_grad_allreduce on -- the very calls every rank of the 8-GPU job makes -- and
    their gradients are summed on the host, which is what the all-reduce over eight ranks returns.  Held against the same 8192
    molecules as two 4096-molecule batches on a plain context (other panels, other split-K ranges, no communicator): predictions
    bit-identical (a molecule does not see its batch mates), summed gradient to 1e-5.  Reference: Threaded_BatchLearn,
    SMP_omega.h:750-792 (broadcast :771-773, serial add_gradient :784-786)."""
    from graphflow_amd.dist import shard
    from graphflow_amd.smp import SMPOmega
    F, D, C, L, cap = 5, 5, 64, 3, 29
    world, total = 8, 8192
    mols, tg = _cfg3_batch(total)
    params = smp_params(C, F, D, L, 1)
    p = dev(params)
    ctx = gf.Context(0)
    ctx.dist_init(ctx.dist_unique_id(), 0, 1)
    net = SMPOmega(L, C, F, D, cap, True, ctx=ctx)
    net.set_grad_allreduce(True)
    g_sum, preds, covered = np.zeros(net.n_params), [], 0
    for rank in range(world):
        lo, hi = shard(total, rank, world)
        assert hi - lo == 1024 and lo == covered
        covered = hi
        net.prepare(mols[lo:hi])
        pr, _, _ = net.forward(p, dev(tg[lo:hi]))
        g = torch.full((net.n_params,), float("nan"), device="cuda")
        net.backward(p, g)
        ctx.dist_quiesce()
        g_sum += g.cpu().numpy().astype(np.float64)
        preds.append(pr.cpu().numpy())
    assert covered == total
    net.close()
    ctx.close()
    preds = np.concatenate(preds)
    g_two, p_two = np.zeros_like(g_sum), []
    big = SMPOmega(L, C, F, D, cap, True)
    for lo in (0, 4096):
        big.prepare(mols[lo:lo + 4096])
        pr, _, _ = big.forward(p, dev(tg[lo:lo + 4096]))
        g = torch.full((big.n_params,), float("nan"), device="cuda")
        big.backward(p, g)
        g_two += g.cpu().numpy().astype(np.float64)
        p_two.append(pr.cpu().numpy())
    big.close()
    p_two = np.concatenate(p_two)
    e = rel_err(g_sum, g_two)
    note("cfg4_eight_shards_vs_two_halves", grads=e, pred=rel_err(preds.astype(np.float64), p_two.astype(np.float64)))
    print("cfg4 workload: 8 x 1024 one-rank shards vs 2 x 4096: gradient rel err %.2e, predictions %s" %
          (e, "bit-identical" if np.array_equal(preds, p_two) else "rel err %.2e" % rel_err(preds.astype(np.float64), p_two.astype(np.float64))))
    assert np.isfinite(g_sum).all() and np.abs(g_sum).max() > 0
    assert rel_err(preds.astype(np.float64), p_two.astype(np.float64)) <= 1e-6
    assert e <= TOL_GRAD


@pytest.mark.parametrize("nV,L,C", [(40, 3, 64), (48, 3, 64), (64, 3, 64), (48, 3, 32), (48, 3, 10), (40, 3, 16), (48, 4, 64), (48, 4, 10)])
def test_fields_above_32_run_the_fused_level(gf, nV, L, C, monkeypatch):
    """SMP_beta (no receptive-field cap, SMP_beta.h) on molecules larger than QM9's: a 40- / 48- / 64-atom molecule's level-3 fields
    reach 34 / 35 / 37 positions, beyond the 32 the fused level's register classes and 32-row panels take.  Since round 6 such a level
    stays on the fused kernels at C = 64: its few nodes above 32 positions run tables-forward on smp_tables_fwd_big and the two combine
    steps on the workgroup kernels (smp_fused.hip: big_part), everything else is row-based.  Held against (a) the op-by-op level
    pipeline of the same batch (fused = False), kink-aware, (b) the fp64 port of the reference for one of the big molecules, and
    (c) GF_SMP_BIG_FIELDS=0, which must reproduce round 5's behaviour (the level op by op: the promotion buffer is taken).  L = 4: the
    fields of level 4 reach 46 positions and its SOURCES (level 3) 41 -- the consumer gather's 64-record class (smp_bwd_gather_all)."""
    from oracle import pyoracle
    F, D = 5, 2
    mols, tg = [], []
    for seed in {40: (8003, 8017, 8026, 8001), 48: (8017, 8003, 8026, 8028, 8000), 64: (8005, 8003, 8001, 8007, 8000)}[nV]:
        adj, feat, t = synthetic_molecule(seed, nV=nV)   # (seeds whose level-3 fields exceed 32: 35 / 34 / 37 ..., 41 / 36 / ..., 40 / 38 / ...)
        mols.append((adj, feat))
        tg.append(t)
    for i in range(7):   # ... in one batch with ordinary molecules (the level's small size classes and panels beside the big nodes)
        adj, feat, t = synthetic_molecule(8100 + i)
        mols.append((adj, feat))
        tg.append(t)
    tg = np.array(tg)
    params = smp_params(C, F, D, L, 11)
    a = run_batch(gf, mols, tg, params, L, C, F, D, nV, fused=True)
    sizes = [len(a[4].receptive_field(0, L, v)) for v in range(nV)]
    assert max(sizes) > 32, sizes                       # the case this test is about
    b = run_batch(gf, mols, tg, params, L, C, F, D, nV, fused=False)
    note("big_fields_nV%d_C%d" % (nV, C), pred=rel_err(a[0], b[0]), feat=rel_err(a[2], b[2]))
    assert np.isfinite(a[3]).all() and np.abs(a[3]).max() > 0
    assert rel_err(a[0], b[0]) <= TOL_FWD and rel_err(a[2], b[2]) <= TOL_FWD
    assert_grads_agree_kink_aware("big_fields_nV%d_C%d" % (nV, C), a[3], b[3], a[4], b[4], mols, L)
    # ... and block by block (H, the 18 slices of every K_l, b_l, W), each against its OWN largest element: the first 32-channel build lost
    # the far rows of ONE slice's gradient (K_3 slice 11, 4e-4 of the block) and stayed within 2.4e-5 of the whole vector's maximum
    o, worst = C * F * (D + 1), 0.0
    spans = [(0, o)]
    for l in range(L):
        spans += [(o + k * C * C, o + (k + 1) * C * C) for k in range(18)] + [(o + 18 * C * C, o + 18 * C * C + C)]
        o += 18 * C * C + C
    spans.append((o, o + C))
    flips = slope_flips(a[4], b[4], mols, L)[0]
    for lo, hi in spans:
        worst = max(worst, np.abs(a[3][lo:hi] - b[3][lo:hi]).max() / max(np.abs(b[3][lo:hi]).max(), 1e-30))
    note("big_fields_nV%d_C%d" % (nV, C), grads_worst_block=worst)
    assert worst <= (2e-5 if flips == 0 else KINK_GRAD), worst
    used_f, used_p = a[4].device_bytes()[0], b[4].device_bytes()[0]
    ppos = b[4].level_sizes(L)[2]
    assert used_p - used_f >= 4 * ppos * C * 0.9        # the fused level took no promotion buffer: it really ran
    if nV == 40:   # (C = 64 and C = 16)   # the port (10 s per molecule of this size): one big molecule, prediction + Feature + gradient
        adj, ft = mols[0]
        one = run_batch(gf, [mols[0]], tg[:1], params, L, C, F, D, nV, fused=True)
        signs = [[one[4].activation(0, l, v) for v in range(nV)] for l in range(L + 1)]
        o = pyoracle.port_smp_molecule(adj, ft, float(tg[0]), params, L, C, D, nV, True, ext_sign=signs, kink_tol=KINK_TOL)
        assert o["n_conflict"] == 0
        e = dict(pred=abs(float(one[0][0]) - o["predict"]) / max(abs(o["predict"]), float(np.abs(o["graph_feature"]).max()), 1.0),
                 feat=rel_err(one[2][0], o["graph_feature"]), grads=rel_err(one[3], o["grads"]))
        note("big_fields_vs_port_C%d" % C, **e)
        print("fields above 32 (a %d-atom molecule, largest field %d) against the fp64 port: %s" % (nV, max(sizes), {k: "%.2e" % v for k, v in e.items()}))
        assert e["pred"] <= TOL_FWD and e["feat"] <= TOL_FWD and e["grads"] <= TOL_GRAD
        one[4].close()
    monkeypatch.setenv("GF_SMP_BIG_FIELDS", "0")
    c = run_batch(gf, mols, tg, params, L, C, F, D, nV, fused=True)
    assert c[4].device_bytes()[0] > used_f                # (round 5's path: level L op by op)
    assert rel_err(c[0], b[0]) <= TOL_FWD
    for r in (a, b, c):
        r[4].close()


@pytest.mark.parametrize("nK,C", [(10, 10), (50, 10), (10, 16)])
def test_smp_2d_ver6_ver7_with_fields_above_32(gf, monkeypatch, nK, C):
    """SMP_2D_ver6 / ver7 have no receptive-field cap: on 48-atom molecules (level-3 fields up to 41 positions) the `_10` / `_50` levels stay
    embedded in the fused 18-slice level since round 6 (a cap of up to 64; 32 before).  Against the op-by-op `_10` / `_50` levels
    (GF_SMP_VER6_FUSED=0 / GF_SMP_VER7_FUSED=0), which the goldens pin to the real SMP_2D_ver6 / ver7."""
    from graphflow_amd.smp import SMPOmega
    L, F, D, nV = 3, 5, 2, 48
    mols, tg = [], []
    for seed in (8017, 8003, 8026, 8028):
        adj, feat, t = synthetic_molecule(seed, nV=nV)
        mols.append((adj, feat))
        tg.append(t)
    for i in range(6):
        adj, feat, t = synthetic_molecule(8100 + i)
        mols.append((adj, feat))
        tg.append(t)
    tg = np.array(tg)

    def step():
        net = SMPOmega(L, C, F, D, nV, True, nContractions=nK, custom_matmul=True)
        params = f32exact(np.random.default_rng(19).uniform(-1, 1, net.n_params) / np.sqrt(nK * C))
        net.prepare(mols)
        p = dev(params)
        pred, loss, feat = net.forward(p, dev(tg))
        g = torch.full((net.n_params,), float("nan"), device="cuda")
        net.backward(p, g)
        return [x.cpu().numpy().astype(np.float64) for x in (pred, feat, g)] + [net]

    a = step()
    assert max(len(a[3].receptive_field(0, L, v)) for v in range(nV)) > 32
    monkeypatch.setenv("GF_SMP_VER6_FUSED", "0")
    monkeypatch.setenv("GF_SMP_VER7_FUSED", "0")
    b = step()
    name = "ver%d_big_fields_C%d" % (6 if nK == 10 else 7, C)
    note(name, pred=rel_err(a[0], b[0]), feat=rel_err(a[1], b[1]))
    assert np.isfinite(a[2]).all() and not np.array_equal(a[2], b[2])
    assert rel_err(a[0], b[0]) <= TOL_FWD and rel_err(a[1], b[1]) <= TOL_FWD
    assert_grads_agree_kink_aware(name, a[2], b[2], a[3], b[3], mols, L)
    a[3].close()
    b[3].close()


@pytest.mark.parametrize("C,mixed", [(64, False), (64, True), (10, False)])
def test_dense_graphs_whose_fields_are_all_above_32(gf, C, mixed):
    """An Erdos-Renyi graph of 50 vertices (the size tests/test_graph_permutation_invariant.cu runs): from level 2 on EVERY receptive
    field is (nearly) the whole graph -- a level without a single row panel, whose sources are above 32 positions as well (the consumer
    gather's 64-record class).  Alone in the batch, and beside QM9-size molecules.  Fused against the op-by-op levels."""
    from inputs import er_graph
    L, F, D, V = 3, 5, 2, 50
    mols, tg = [], []
    for seed in (3, 4):   # (p = 0.2: levels 2 AND 3 without a node within 32 positions -- 33 .. 50 and 50)
        adj, feat = er_graph(V, 0.2, F, seed)
        mols.append((adj, feat))
        tg.append(1.5)
    if mixed:
        for i in range(5):
            adj, feat, t = synthetic_molecule(8300 + i)
            mols.append((adj, feat))
            tg.append(t)
    tg = np.array(tg)
    params = smp_params(C, F, D, L, 13) * 0.1   # (dense fields: keep the activations of level 3 in range)
    a = run_batch(gf, mols, tg, params, L, C, F, D, V, fused=True)
    s2, s3 = ([len(a[4].receptive_field(0, l, v)) for v in range(V)] for l in (2, 3))
    assert min(s2) > 32 and min(s3) > 32, (s2, s3)   # not one node within 32 positions at levels 2 and 3; level 3's sources above 32 as well
    b = run_batch(gf, mols, tg, params, L, C, F, D, V, fused=False)
    name = "dense_er50_C%d%s" % (C, "_mixed" if mixed else "")
    note(name, pred=rel_err(a[0], b[0]), feat=rel_err(a[2], b[2]))
    assert np.isfinite(a[3]).all() and np.abs(a[3]).max() > 0
    assert rel_err(a[0], b[0]) <= TOL_FWD and rel_err(a[2], b[2]) <= TOL_FWD
    assert_grads_agree_kink_aware(name, a[3], b[3], a[4], b[4], mols, L)
    assert a[4].device_bytes()[0] < b[4].device_bytes()[0]   # (no promoted stack: the fused levels ran)
    a[4].close()
    b[4].close()


def test_fields_above_32_in_a_level_beyond_the_panel_kernels_offsets(gf, monkeypatch):
    """The same with a level of more than 2^21 rows (200 molecules of 64 atoms: the panel combine kernels address O with 32-bit byte
    offsets and hand such a level to the workgroup kernels -- with nodes above 32 positions, their 64-position build for every node):
    predictions, features and the gradient against the op-by-op level (GF_SMP_BIG_FIELDS=0)."""
    L, C, F, D, nV = 3, 64, 5, 2, 64
    mols, tg = [], []
    for seed in range(8000, 8200):
        adj, feat, t = synthetic_molecule(seed, nV=nV)
        mols.append((adj, feat))
        tg.append(t)
    tg = np.array(tg)
    params = smp_params(C, F, D, L, 12)
    a = run_batch(gf, mols, tg, params, L, C, F, D, nV, fused=True)
    assert a[4].level_sizes(L)[1] * 512 >= 0x3fffffff
    monkeypatch.setenv("GF_SMP_BIG_FIELDS", "0")
    b = run_batch(gf, mols, tg, params, L, C, F, D, nV, fused=True)
    assert b[4].device_bytes()[0] > a[4].device_bytes()[0]
    note("big_fields_large_level", pred=rel_err(a[0], b[0]), feat=rel_err(a[2], b[2]))
    assert rel_err(a[0], b[0]) <= TOL_FWD and rel_err(a[2], b[2]) <= TOL_FWD
    assert_grads_agree_kink_aware("big_fields_large_level", a[3], b[3], a[4], b[4], mols, L)
    a[4].close()
    b[4].close()


def test_fused_small_launch_equals_the_two_launches(gf, monkeypatch):
    """Round 6: smp_reduce_pairs and diag_gather_bwd run as ONE launch (smp_reduce_pairs_and_diag_gather); GF_SMP_FUSE_SMALL=0 keeps
    the two.  Same bodies, same summation order: predictions and gradients bit for bit."""
    F, D, C, L, cap = 5, 3, 64, 3, 29
    mols, tg = [], []
    for seed in range(40):
        adj, feat, t = synthetic_molecule(9100 + seed)
        mols.append((adj, feat))
        tg.append(t)
    params = smp_params(C, F, D, L, 5)
    a = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap)
    monkeypatch.setenv("GF_SMP_FUSE_SMALL", "0")
    b = run_batch(gf, mols, np.array(tg), params, L, C, F, D, cap)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[3], b[3])
    a[4].close()
    b[4].close()


def test_text_checkpoint_is_interchangeable_with_the_reference(gf, golden, tmp_path):
    """tests/golden/smp_syn12_checkpoint.txt was written by the REAL SMP_omega::save_model (SMP_omega.h:1033-1042).
    load_model must read it, predict like the golden, and save_model must write the very same bytes back."""
    import os
    from graphflow_amd.smp import SMPOmega
    path = os.path.join(os.path.dirname(__file__), "golden", "smp_syn12_checkpoint.txt")
    c = golden_cases(golden, "smp_syn12")["smp_syn12"]
    L, C, D, cap, wl = (int(x) for x in c["cfg"])
    net = SMPOmega(L, C, c["feature"].shape[1], D, cap, bool(wl))
    p = torch.zeros(net.n_params, device="cuda")
    net.load_model(p, path)
    text = np.array(open(path).read().split(), dtype=np.float64)
    assert text.size == net.n_params
    assert np.array_equal(p.cpu().numpy(), text.astype(np.float32))
    assert rel_err(p.cpu().numpy().astype(np.float64), c["params"].astype(np.float64)) <= 1e-6  # 6 printed digits
    net.prepare([(c["adj"], c["feature"])])
    pred, loss, feat = net.forward(p, dev(c["target"]))
    assert rel_err(pred.cpu().numpy().astype(np.float64), c["predict"]) <= TOL_FWD
    out = tmp_path / "resaved.txt"
    net.save_model(p, out)
    assert open(out, "rb").read() == open(path, "rb").read()
    # a short file is an error, not silent garbage
    short = tmp_path / "short.txt"
    short.write_text("1 2 3 ")
    with pytest.raises(RuntimeError):
        net.load_model(p, short)


def test_batchlearn_steps_match_the_reference(gf):
    """Three optimiser steps = SMP_omega::BatchLearn x 3 of the real reference (tests/golden/smp_train.npz): initial weights
    from gf_smp_uniform_init_host after srand(7), forward/backward on the four toy molecules as one batch, then
    gf_smp_adam_step (Adam::Learn(alpha, nBatch) with its per-element bias-correction powers).
    Tolerance: the fp32 path is held to 1e-4 on the losses and to 0.5 % of one Adam step (|step| = alpha = 1e-3) on every
    parameter -- Adam normalises the gradient, so a parameter whose true gradient is ~0 turns fp32 rounding into a
    visible fraction of a step; everything else agrees to ~1e-7."""
    import ctypes as C
    import os
    from graphflow_amd.smp import SMPOmega
    from inputs import toy_molecules
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "smp_train.npz"))
    L, Cn, D, cap, maxV, seed, nIter = (int(x) for x in z["train__cfg"])
    mols = [(adj, feat) for _, adj, feat, _ in toy_molecules()]
    tg = dev(z["train__targets"])
    lr = float(z["train__lr"][0])
    net = SMPOmega(L, Cn, mols[0][1].shape[1], D, cap, True)
    C.CDLL(None).srand(seed)
    p = dev(net.uniform_init())
    assert np.array_equal(p.cpu().numpy(), z["train__params0"].astype(np.float32))
    net.prepare(mols)
    grads = torch.empty(net.n_params, device="cuda")
    worst = 0.0
    for it in range(nIter):
        _, loss, _ = net.forward(p, tg)
        before = float(loss.sum())
        net.backward(p, grads)
        net.adam_step(p, grads, lr, len(mols))
        _, loss, _ = net.forward(p, tg)
        after = float(loss.sum())
        assert abs(before - z["train__losses"][it, 0]) <= TOL_FWD * max(1.0, before), it
        assert abs(after - z["train__losses"][it, 1]) <= 5 * TOL_FWD * max(1.0, after), it
    err = np.abs(p.cpu().numpy().astype(np.float64) - z["train__params"])
    print("max |param - reference| after %d steps: %.3e (median %.3e)" % (nIter, err.max(), np.median(err)))
    assert err.max() <= 0.005 * lr
    assert np.median(err) <= 1e-6
    # a second model instance restarts the bias-correction powers; reset does the same on this one
    net.adam_reset()


@pytest.mark.parametrize("fused", [True, False])
def test_edge_molecules_vs_oracle(gf, fused):
    """Single atom, bonded pair, disconnected fragments, isolated vertex, 9-clique, 12-path -- one at a time (a batch of one
    single-atom molecule is the smallest legal input) and as one ragged batch; the oracle is pinned to the real reference
    on the same molecules in the CPU suite."""
    from inputs import edge_molecules
    from oracle import smp_oracle
    L, C, F, D, cap = 3, 8, 5, 2, 6
    params = smp_params(C, F, D, L, 21)
    cases = edge_molecules()
    refs = [smp_oracle.run(a, f, t, params, L, C, D, cap) for _, a, f, t in cases]
    for (name, a, f, t), r in zip(cases, refs):
        pred, loss, feat, grads, net = run_batch(gf, [(a, f)], np.array([t]), params, L, C, F, D, cap, fused=fused)
        assert [net.receptive_field(0, L, v) for v in range(len(a))] == [list(map(int, x)) for x in r["phi"][L]], name
        assert rel_err(pred, np.array([r["predict"]])) <= TOL_FWD, name
        assert rel_err(feat[0], r["graph_feature"]) <= TOL_FWD, name
        assert rel_err(grads, r["grads"]) <= TOL_GRAD, name
    pred, loss, feat, grads, _ = run_batch(gf, [(a, f) for _, a, f, _ in cases], np.array([t for *_, t in cases]), params,
                                           L, C, F, D, cap, fused=fused)
    assert rel_err(pred, np.array([r["predict"] for r in refs])) <= TOL_FWD
    assert rel_err(loss, np.array([r["loss"] for r in refs])) <= 2 * TOL_FWD
    assert rel_err(grads, sum(r["grads"] for r in refs)) <= TOL_GRAD


def test_prepare_rejects_bad_input(gf):
    from graphflow_amd.smp import SMPOmega
    net = SMPOmega(2, 8, 5, 2, 6, True)
    with pytest.raises(Exception):
        net.prepare([])
    p = torch.zeros(net.n_params, device="cuda")
    with pytest.raises(Exception):
        net.forward(p)   # forward before prepare


@pytest.mark.parametrize("fused", [True, False])
def test_handle_reuse_across_batches_is_exact(gf, fused):
    """A training loop prepares a new batch on the same handle every step; the device buffers come from a pool that is
    reused (not re-allocated, not zeroed).  Results on batch B after batch A must be bit-identical to a fresh handle."""
    from graphflow_amd.smp import SMPOmega
    L, C, F, D, cap = 3, 16, 5, 2, 8
    params = dev(smp_params(C, F, D, L, 3))

    def batch(seeds, sizes):
        mols, tg = [], []
        for s, n in zip(seeds, sizes):
            a, f, t = synthetic_molecule(s, nV=n)
            mols.append((a, f))
            tg.append(t)
        return mols, dev(np.array(tg))

    A = batch(range(200, 212), [5, 9, 14, 7, 11, 3, 16, 8, 12, 6, 10, 15])
    B = batch(range(300, 309), [13, 4, 9, 17, 6, 11, 8, 15, 5])

    def run(net, mols, tg):
        net.set_fused(fused)
        net.prepare(mols)
        pred, loss, feat = net.forward(params, tg)
        g = torch.empty(net.n_params, device="cuda")
        net.backward(params, g)
        return pred.clone(), feat.clone(), g

    reused = SMPOmega(L, C, F, D, cap, True)
    run(reused, *A)
    got = run(reused, *B)
    got_a = run(reused, *A)      # and back to the larger batch
    fresh_b = run(SMPOmega(L, C, F, D, cap, True), *B)
    fresh_a = run(SMPOmega(L, C, F, D, cap, True), *A)
    for x, y in zip(got, fresh_b):
        assert torch.equal(x, y)
    for x, y in zip(got_a, fresh_a):
        assert torch.equal(x, y)


def test_smp_2d_ver6_batchlearn_matches_the_reference(gf):
    """Three BatchLearn steps of the real SMP_2D_ver6 (RisiContraction_10 + CustomMatMulTensor weights + Momentum 0.9;
    tests/golden/smp_train.npz): same srand -> same initial weights, then forward/backward/gf_smp_momentum_step."""
    import ctypes as C
    import os
    from graphflow_amd.smp import SMPOmega
    from inputs import toy_molecules
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "smp_train.npz"))
    L, Cn, D, maxV, seed, nIter = (int(x) for x in z["train2d6__cfg"])
    mols = [(adj, feat) for _, adj, feat, _ in toy_molecules()]
    tg = dev(np.array([t for *_, t in toy_molecules()]))
    lr, gamma = float(z["train2d6__lr"][0]), float(z["train2d6__momentum"][0])
    net = SMPOmega(L, Cn, mols[0][1].shape[1], D, maxV, True, nContractions=10, custom_matmul=True)
    C.CDLL(None).srand(seed)
    p = dev(net.uniform_init())
    assert np.array_equal(p.cpu().numpy(), z["train2d6__params0"].astype(np.float32))
    net.prepare(mols)
    grads = torch.empty(net.n_params, device="cuda")
    for it in range(nIter):
        _, loss, _ = net.forward(p, tg)
        before = float(loss.sum())
        net.backward(p, grads)
        net.momentum_step(p, grads, lr, len(mols), gamma)
        _, loss, _ = net.forward(p, tg)
        after = float(loss.sum())
        assert abs(before - z["train2d6__losses"][it, 0]) <= 5 * TOL_FWD * max(1.0, before), it
        assert abs(after - z["train2d6__losses"][it, 1]) <= 5 * TOL_FWD * max(1.0, after), it
    err = np.abs(p.cpu().numpy().astype(np.float64) - z["train2d6__params"])
    scale = np.abs(z["train2d6__params"] - z["train2d6__params0"]).max()
    print("max |param - reference| %.3e, largest parameter change %.3e" % (err.max(), scale))
    assert err.max() <= 1e-3 * scale


def test_smp_2d_ver7_wiring_at_32_channels_runs_the_matrix_pipe_contractions(gf, monkeypatch):
    """The SMP_2D_ver7 wiring (RisiContraction_50 per node, op by op) at 32 channels: the driver hands gf_contract_*_f32 one uniform
    batch per receptive-field size, which at C % 32 == 0 runs fam50_forward_mfma / fam50_bwd_tables_mfma.  Predictions, features
    and every parameter gradient against the same step on the thread-per-element kernels (GF_FAM_*_MFMA=0), which the goldens of
    tests/golden/smp.npz pin to the real SMP_2D_ver7 at small channel counts."""
    from graphflow_amd.smp import SMPOmega
    L, C, F, D, cap = 2, 32, 5, 2, 10
    mols, tg = [], []
    for seed in range(8):
        adj, feat, t = synthetic_molecule(300 + seed, nV=3 + 2 * seed)
        mols.append((adj, feat))
        tg.append(t)

    def step():
        net = SMPOmega(L, C, F, D, cap, True, nContractions=50, custom_matmul=False)
        rng = np.random.default_rng(7)
        params = f32exact(rng.uniform(-1, 1, net.n_params) / np.sqrt(50 * C))
        net.prepare(mols)
        p = dev(params)
        pred, loss, feat = net.forward(p, dev(np.array(tg)))
        grads = torch.empty(net.n_params, device="cuda")
        net.backward(p, grads)
        out = [x.cpu().numpy().astype(np.float64) for x in (pred, feat, grads)]
        net.close()
        return out

    monkeypatch.setenv("GF_SMP_VER7_FUSED", "0")   # (the op-by-op `_50` level; the default since round 5 is the 18-slice fused level, below)
    a = step()
    monkeypatch.setenv("GF_FAM_FWD_MFMA", "0")
    monkeypatch.setenv("GF_FAM_BWD_MFMA", "0")
    b = step()
    note("ver7_c32_mfma_vs_threads", pred=rel_err(a[0], b[0]), feat=rel_err(a[1], b[1]), grads=rel_err(a[2], b[2]))
    assert np.isfinite(a[2]).all() and np.abs(a[2]).max() > 0
    assert not np.array_equal(a[2], b[2])   # (the switches switch something)
    assert rel_err(a[0], b[0]) <= TOL_FWD and rel_err(a[1], b[1]) <= TOL_FWD
    assert rel_err(a[2], b[2]) <= TOL_GRAD


@pytest.mark.parametrize("nK", [10, 50])
def test_ver6_ver7_embedding_at_the_headline_sizes(gf, monkeypatch, nK):
    """BASELINE configs[2]'s molecules (QM9-size, cap 29, three levels) at the reference's 10 channels, 192 of them: the `_10` / `_50` models on
    the fused 18-slice level against their op-by-op levels -- fields up to 29, every size class of tables-forward and of the gather, the
    in-kernel extra products of ver7 (32 padded channels)."""
    from graphflow_amd.smp import SMPOmega
    L, C, F, D, cap = 3, 10, 5, 2, 29
    mols, tg = [], []
    for i in range(192):
        adj, feat, t = synthetic_molecule(i)
        mols.append((adj, feat))
        tg.append(t)

    def step():
        net = SMPOmega(L, C, F, D, cap, True, nContractions=nK, custom_matmul=True)
        params = f32exact(np.random.default_rng(nK).uniform(-1, 1, net.n_params) / np.sqrt(nK * C))
        net.prepare(mols)
        p = dev(params)
        pred, loss, feat = net.forward(p, dev(np.array(tg)))
        grads = torch.zeros(net.n_params, device="cuda")
        net.backward(p, grads)
        return [x.cpu().numpy().astype(np.float64) for x in (pred, feat, grads)] + [net]

    a = step()
    monkeypatch.setenv("GF_SMP_VER6_FUSED", "0")
    monkeypatch.setenv("GF_SMP_VER7_FUSED", "0")
    b = step()
    name = "ver%d_embedding_headline_sizes" % (6 if nK == 10 else 7)
    note(name, pred=rel_err(a[0], b[0]), feat=rel_err(a[1], b[1]))
    assert np.isfinite(a[2]).all() and not np.array_equal(a[2], b[2])
    assert rel_err(a[0], b[0]) <= TOL_FWD and rel_err(a[1], b[1]) <= TOL_FWD
    assert_grads_agree_kink_aware(name, a[2], b[2], a[3], b[3], mols, L)
    a[3].close(), b[3].close()


@pytest.mark.parametrize("C,custom,fused", [(10, False, True), (10, True, False), (6, False, True), (32, True, True), (3, False, False)])
def test_smp_2d_ver7_on_the_fused_level_equals_the_op_by_op_level(gf, monkeypatch, C, custom, fused):
    """RisiContraction_50 wiring on the 18-slice level (gf_smp::dup_channels + n_extra: 46 slices as RisiContraction_18 slices on f / f^T, the
    other four as three extra products on the level's tables), with the fused kernels and with the op-by-op `_18` pipeline under them
    (fused = False: the extra products on slices of Q), either weight layout, three levels, edge molecules: predictions, features, every
    gradient against the op-by-op `_50` level, which the goldens pin to the real SMP_2D_ver7 (the embedded path runs those goldens too)."""
    from graphflow_amd.smp import SMPOmega
    L, F, D, cap = 3, 5, 2, 9
    mols, tg = [], []
    for seed in range(24):
        adj, feat, t = synthetic_molecule(7300 + seed, nV=1 + seed % 11)
        mols.append((adj, feat))
        tg.append(t)

    def step(fz):
        net = SMPOmega(L, C, F, D, cap, True, nContractions=50, custom_matmul=custom)
        if not fz:
            net.set_fused(False)
        rng = np.random.default_rng(19)
        params = f32exact(rng.uniform(-1, 1, net.n_params) / np.sqrt(50 * C))
        net.prepare(mols)
        p = dev(params)
        pred, loss, feat = net.forward(p, dev(np.array(tg)))
        grads = torch.zeros(net.n_params, device="cuda")
        net.backward(p, grads)
        return [x.cpu().numpy().astype(np.float64) for x in (pred, feat, grads)] + [net]

    a = step(fused)
    monkeypatch.setenv("GF_SMP_VER7_FUSED", "0")
    b = step(True)
    name = "ver7_on_18_%s_vs_op_by_op_C%d" % ("fused" if fused else "unfused", C)
    note(name, pred=rel_err(a[0], b[0]), feat=rel_err(a[1], b[1]))
    assert np.isfinite(a[2]).all() and np.abs(a[2]).max() > 0 and not np.array_equal(a[2], b[2])
    assert rel_err(a[0], b[0]) <= TOL_FWD and rel_err(a[1], b[1]) <= TOL_FWD
    assert_grads_agree_kink_aware(name, a[2], b[2], a[3], b[3], mols, L)
    a[3].close(), b[3].close()


def test_smp_2d_ver6_wiring_runs_the_graph_stream_contractions(gf, monkeypatch):
    """The SMP_2D_ver6 wiring (RisiContraction_10 per node, op by op, [C][10 C] weights) on a batch whose size buckets hold hundreds of
    nodes: the driver hands gf_contract_*_f32 one uniform batch per receptive-field size, which from 96 nodes on runs r10_fwd_graph /
    r10_bwd_graph (round 4).  Predictions, features and every parameter gradient against the same step on the table kernels
    (GF_FAM10_GRAPH=0), which the goldens of tests/golden/smp.npz pin to the real SMP_2D_ver6."""
    from graphflow_amd.smp import SMPOmega
    L, C, F, D, cap = 2, 32, 5, 2, 10
    mols, tg = [], []
    for seed in range(160):
        adj, feat, t = synthetic_molecule(5000 + seed, nV=6 + seed % 5)
        mols.append((adj, feat))
        tg.append(t)

    def step(keep=False):
        net = SMPOmega(L, C, F, D, cap, True, nContractions=10, custom_matmul=True)
        rng = np.random.default_rng(8)
        params = f32exact(rng.uniform(-1, 1, net.n_params) / np.sqrt(10 * C))
        net.prepare(mols)
        p = dev(params)
        pred, loss, feat = net.forward(p, dev(np.array(tg)))
        grads = torch.empty(net.n_params, device="cuda")
        net.backward(p, grads)
        out = [x.cpu().numpy().astype(np.float64) for x in (pred, feat, grads)]
        if keep:
            return out + [net]
        net.close()
        return out

    monkeypatch.setenv("GF_SMP_VER6_FUSED", "0")   # (the op-by-op `_10` level; the default since round 5 is the 18-slice fused level, below)
    a = step()
    monkeypatch.setenv("GF_FAM10_GRAPH", "0")
    b = step()
    note("ver6_c32_graph_streams_vs_tables", pred=rel_err(a[0], b[0]), feat=rel_err(a[1], b[1]), grads=rel_err(a[2], b[2]))
    assert np.isfinite(a[2]).all() and np.abs(a[2]).max() > 0
    assert not np.array_equal(a[2], b[2])   # (the switch switches something)
    assert rel_err(a[0], b[0]) <= TOL_FWD and rel_err(a[1], b[1]) <= TOL_FWD
    assert rel_err(a[2], b[2]) <= TOL_GRAD
    # Round 5: the same model on the FUSED level -- RisiContraction_10's slices as slices of RisiContraction_18 on [f | f^T] channels
    # (gf_smp::dup_channels; 2 x 32 = 64 channels here) -- against the op-by-op `_10` level
    b = step(keep=True)
    monkeypatch.delenv("GF_SMP_VER6_FUSED")
    monkeypatch.delenv("GF_FAM10_GRAPH")
    c = step(keep=True)
    note("ver6_c32_fused18_vs_op_by_op", pred=rel_err(c[0], b[0]), feat=rel_err(c[1], b[1]))
    assert not np.array_equal(c[2], b[2])
    assert rel_err(c[0], b[0]) <= TOL_FWD and rel_err(c[1], b[1]) <= TOL_FWD
    assert_grads_agree_kink_aware("ver6_c32_fused18_vs_op_by_op", c[2], b[2], c[3], b[3], mols, L)
    c[3].close(), b[3].close()


@pytest.mark.parametrize("C,custom", [(10, True), (7, False), (16, True), (3, True)])
def test_smp_2d_ver6_on_the_fused_level_equals_the_op_by_op_level(gf, monkeypatch, C, custom):
    """RisiContraction_10 wiring at channel counts that pad to 32 (10, 16), 16 (7, 3 -> 14, 6 channels of [f | f^T]) with either weight layout,
    three levels, edge molecules included: predictions, features, every gradient against the op-by-op `_10` level (pinned to the real
    SMP_2D_ver6 by the goldens and the BatchLearn trajectory).  Then accumulate = 1 twice the gradient, and an asymmetric adjacency is refused."""
    from graphflow_amd.smp import SMPOmega
    from graphflow_amd import _lib
    L, F, D, cap = 3, 5, 2, 9
    mols, tg = [], []
    for seed in range(24):
        adj, feat, t = synthetic_molecule(7000 + seed, nV=1 + seed % 11)
        mols.append((adj, feat))
        tg.append(t)

    def step(acc2=False):
        net = SMPOmega(L, C, F, D, cap, True, nContractions=10, custom_matmul=custom)
        rng = np.random.default_rng(18)
        params = f32exact(rng.uniform(-1, 1, net.n_params) / np.sqrt(10 * C))
        net.prepare(mols)
        p = dev(params)
        pred, loss, feat = net.forward(p, dev(np.array(tg)))
        grads = torch.zeros(net.n_params, device="cuda")
        net.backward(p, grads)
        if acc2:
            net.forward(p, dev(np.array(tg)))
            net.backward(p, grads, accumulate=True)
        return [x.cpu().numpy().astype(np.float64) for x in (pred, feat, grads)] + [net]

    a = step()
    a2 = step(acc2=True)
    monkeypatch.setenv("GF_SMP_VER6_FUSED", "0")
    b = step()
    note("ver6_fused18_vs_op_by_op_C%d" % C, pred=rel_err(a[0], b[0]), feat=rel_err(a[1], b[1]))
    assert np.isfinite(a[2]).all() and np.abs(a[2]).max() > 0 and not np.array_equal(a[2], b[2])
    assert rel_err(a[0], b[0]) <= TOL_FWD and rel_err(a[1], b[1]) <= TOL_FWD
    assert_grads_agree_kink_aware("ver6_fused18_vs_op_by_op_C%d" % C, a[2], b[2], a[3], b[3], mols, L)
    assert rel_err(a2[2], 2 * a[2]) <= 1e-6
    for r in (a, a2, b):
        r[3].close()
    monkeypatch.delenv("GF_SMP_VER6_FUSED")
    # Round 6 (round-5 advice): a batch the embedding cannot take -- an asymmetric adjacency, a Coulomb matrix with entries <= 0 (`_18` gates
    # A > 0, `_10` does not) -- is no longer refused at gf_smp_prepare: the handle computes it on the op-by-op `_10` levels and goes back to
    # the fused level for the next batch that qualifies.  Held against a handle created op-by-op (GF_SMP_VER6_FUSED=0), bit for bit.
    net = SMPOmega(L, C, F, D, cap, True, nContractions=10, custom_matmul=custom)
    adj, feat, _ = synthetic_molecule(7100, nV=6)
    bad = np.array(adj).copy()
    i, j = np.argwhere(bad > 0)[0]
    bad[j, i] = 0 if i != j else bad[j, i]
    assert not np.array_equal(bad, bad.T)
    rng = np.random.default_rng(5)
    cpos = rng.uniform(0.1, 2.0, (6, 6))
    cpos = 0.5 * (cpos + cpos.T)
    p = dev(f32exact(rng.uniform(-1, 1, net.n_params) / np.sqrt(10 * C)))
    one = dev(np.array([1.0]))

    def run(h, mol, coulomb=None):
        h.prepare([mol], coulomb=coulomb)
        pr, _, ft = h.forward(p, one)
        g = torch.full((h.n_params,), float("nan"), device="cuda")
        h.backward(p, g)
        return pr.cpu().numpy(), ft.cpu().numpy(), g.cpu().numpy()

    first = run(net, (adj, feat))                       # fused (embedded) plan
    asym = run(net, (bad, feat))                        # -> op-by-op plan, by itself
    signed = run(net, (adj, feat), [cpos - 1.0])        # Coulomb entries <= 0: op-by-op as well
    pos = run(net, (adj, feat), [cpos])                 # positive symmetric Coulomb matrix: back on the fused level
    again = run(net, (adj, feat))                       # and the first batch once more: the same bits as before the detour
    assert all(np.array_equal(x, y) for x, y in zip(first, again))
    net.close()
    monkeypatch.setenv("GF_SMP_VER6_FUSED", "0")
    ref = SMPOmega(L, C, F, D, cap, True, nContractions=10, custom_matmul=custom)
    r_asym, r_signed, r_pos = run(ref, (bad, feat)), run(ref, (adj, feat), [cpos - 1.0]), run(ref, (adj, feat), [cpos])
    ref.close()
    for got, want in ((asym, r_asym), (signed, r_signed)):
        assert np.isfinite(got[2]).all() and all(np.array_equal(x, y) for x, y in zip(got, want))   # the same plan: the same bits
    assert rel_err(pos[0], r_pos[0]) <= TOL_FWD and rel_err(pos[1], r_pos[1]) <= TOL_FWD
    assert rel_err(pos[2], r_pos[2]) <= TOL_SELF
    assert not np.array_equal(pos[2], r_pos[2])          # (really the other plan)


@pytest.mark.parametrize("C,fused,cap,coul", [(64, True, 29, False), (8, False, 6, False), (16, True, 12, True)])
def test_device_level_tables_equal_the_host_built_ones(gf, monkeypatch, C, fused, cap, coul):
    """The rows-sized level tables (reduced adjacency, gated row sums, (tot, tr), selection maps, inverse maps) are built on the
    device from the receptive fields (smp.hip: build_level_rows / build_level_inv; GF_PREP_DEVICE_TABLES=0: by the host, as
    gfsmp::build_batch's phases B and D write them, SMP_omega.h:461-474, 556-581).  Same batch both ways: reduced adjacencies of
    every node, the count of rows with data, and -- every kernel being deterministic -- predictions, losses and gradients must
    be IDENTICAL bit for bit."""
    from graphflow_amd.smp import SMPOmega
    L, F, D = 3, 5, 2
    rng = np.random.default_rng(C)
    mols = [synthetic_molecule(200 + i)[:2] for i in range(12)]
    cm = None
    if coul:   # the use_coulomb variant: dense signed "adjacency" (SMP_omega.h:568-579)
        cm = [rng.uniform(-1, 2, (len(a), len(a))) for a, _ in mols]
        cm = [0.5 * (c + c.T) for c in cm]
    tg = np.array([synthetic_molecule(200 + i)[2] for i in range(12)], dtype=np.float32)
    params = smp_params(C, F, D, L, 3).astype(np.float32)
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GF_PREP_DEVICE_TABLES", mode)
        net = SMPOmega(L, C, F, D, cap, True)
        net.set_fused(fused)
        net.prepare(mols, coulomb=cm)
        p = dev(params)
        pred, loss, feat = net.forward(p, dev(tg))
        g = torch.empty(net.n_params, device="cuda")
        net.backward(p, g)
        adjs = [net.reduced_adjacency(m, l, v) for m in (0, 5, 11) for l in range(1, L + 1) for v in range(len(mols[m][0]))]
        got[mode] = (pred.cpu().numpy().copy(), loss.cpu().numpy().copy(), g.cpu().numpy().copy(), adjs,
                     [net.level_present_rows(l) for l in range(L + 1)], [net.level_sizes(l) for l in range(L + 1)])
        net.close()
    a, b = got["0"], got["1"]
    assert a[5] == b[5] and a[4] == b[4], (a[4], b[4])
    assert all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    if fused and C <= 32:
        # computed at 32 channels (padded, gf_smp_create): smp_wgrad_direct<32> takes its column exponents from the device-built
        # statistics words (largest |tot|, |tr|) when the tables were built there and from the operands' exact column maxima
        # otherwise -- two valid power-of-two scalings of the same split, equal to the last bits only
        assert rel_err(a[2].astype(np.float64), b[2].astype(np.float64)) <= 2e-7
    else:
        assert np.array_equal(a[2], b[2]), float(np.abs(a[2] - b[2]).max())
    assert np.isfinite(a[2]).all() and np.abs(a[2]).max() > 0


def test_row_flags_count_the_rows_with_data(gf):
    """The structural zeros the C = 64 level skips, counted from the receptive fields: row (a, b) of the S_ab / T6 blocks has data when
    b lies in the field of a's source; row (b, c) of the S_bc / T10 blocks when some source holds both (SMP_omega.h:461-474: the
    selection matrices of MatTensorMul / TensorMatMul have a 1 in those columns).  gf_smp_level_present_rows / _covered_rows report the
    device's flags (build_trow), which the kernels mask with."""
    from graphflow_amd.smp import SMPOmega
    L, C, F, D, cap = 3, 64, 5, 2, 29
    mols = [synthetic_molecule(300 + i)[:2] for i in range(5)]
    net = SMPOmega(L, C, F, D, cap, True)
    net.prepare(mols)
    present, covered, rows = [0] * (L + 1), [0] * (L + 1), [0] * (L + 1)
    for m, (adj, _) in enumerate(mols):
        V = len(adj)
        f = [[net.receptive_field(m, l, v) for v in range(V)] for l in range(L + 1)]
        for l in range(1, L + 1):
            for v in range(V):
                fld = f[l][v]
                srcs = [set(f[l - 1][a]) for a in fld]
                rows[l] += len(fld) ** 2
                present[l] += sum(1 for sa in srcs for b in fld if b in sa)
                for b in fld:
                    u = set().union(*[sa for sa in srcs if b in sa])
                    covered[l] += sum(1 for c in fld if c in u)
    for l in range(1, L + 1):
        assert net.level_sizes(l)[1] == rows[l]
        assert net.level_present_rows(l) == present[l], (l, net.level_present_rows(l), present[l])
        assert net.level_covered_rows(l) == covered[l], (l, net.level_covered_rows(l), covered[l])
        assert present[l] <= covered[l] <= rows[l]
    net.close()


def test_no_kernel_reads_what_nobody_wrote():
    """GF_POISON=1 fills every buffer the library hands out without contents (workspace, pooled level buffers, model buffers) with
    NaN patterns before anything is launched.  The golden, headline, gather and physics parity tests must pass unchanged in such a
    process: no kernel may read memory nobody wrote -- in particular the dS_ab / dT6 rows of structurally-zero slab rows, which
    the backward block products no longer write (smp_level_c64_split.hip), must have no reader."""
    import subprocess
    import sys
    env = dict(os.environ, GF_POISON="1")
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "reference_goldens or headline_shape or gather_kernels_agree or folded_backward or batch_equals_sum or physics_and_pairgraphs"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_smp_gpu.py"), os.path.join(here, "test_physics_gpu.py"),
                        "-q", "-x", "-m", "gpu", "-k", sel, "-p", "no:cacheprovider"], env=env, capture_output=True, text=True,
                       timeout=900)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
    # the contraction families' workspace (pair tables, per-row partials of the scalars, the column launch's tables the row launch
    # starts from): every word a kernel reads there must have been written by an earlier kernel of the same call
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_families_gpu.py"), "-q", "-x", "-m", "gpu", "-k",
                        "golden_vectors or matrix_pipe or cfg5_shape", "-p", "no:cacheprovider"], env=env, capture_output=True,
                       text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail


def test_zz_print_margins(gf):
    """Not a check: prints the measured end-to-end maxima collected above (run with -s; copied to profiles/)."""
    import os
    lines = ["margin %-56s %.3e" % (k, MARGINS[k]) for k in sorted(MARGINS)]
    for ln in lines:
        print(ln)
    if os.environ.get("GF_MARGINS_OUT"):   # (the GPU runs copy this file to profiles/rNN_parity_margins.txt)
        with open(os.environ["GF_MARGINS_OUT"], "w") as fh:
            fh.write("\n".join(lines) + "\n")
