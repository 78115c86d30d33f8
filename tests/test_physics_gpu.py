"""GPU parity for SURVEY 8 f3's drivers: SMP_omega_physics / SMP_beta_physics (one tower) and SMP_omega_pairgraphs /
SMP_beta_pairgraphs / SMP_sigma_pairgraphs (two towers, the last with RisiContraction_18_dropout), through gf_smp_model_*,
against goldens captured from the REAL reference classes (tests/golden/smp_physics.npz, make_golden.py: physics_fixtures)."""
import ctypes as C
import os

import numpy as np
import pytest

from util import golden_cases, rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
TOL = 1e-5


def cases():
    g = {}
    for name in ("smp_physics.npz", "smp_physics_big.npz"):   # (the second: round 6, 40-atom graphs with fields above 32 positions)
        with np.load(os.path.join(os.path.dirname(__file__), "golden", name)) as z:
            g.update({k: z[k] for k in z.files})
    return golden_cases(g, "physics_")


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def run(c):
    from graphflow_amd.smp import SMPModel
    towers, L, Cn, cap, nKept, train = (int(x) for x in c["cfg"])
    feats = [c["feature"].shape[1]] + ([c["feature2"].shape[1]] if towers == 2 else [])
    net = SMPModel(L, Cn, cap, feats, nKept=nKept)
    assert net.n_params == c["params"].size
    net.prepare([(c["adj"], c["feature"])], [(c["adj2"], c["feature2"])] if towers == 2 else None)
    net.set_mode(bool(train))
    if nKept and train:
        C.CDLL(None).srand(int(c["seed"][0]))   # the reference drew its slice masks with rand() after this srand
    p = dev(c["params"])
    pred, loss = net.forward(p, dev(c["target"]))
    g = torch.full((net.n_params,), float("nan"), device="cuda")
    if train:
        net.backward(p, g)
    return pred.cpu().numpy().astype(np.float64), loss.cpu().numpy().astype(np.float64), g.cpu().numpy().astype(np.float64), net


def test_physics_and_pairgraphs_goldens(gf):
    cs = cases()
    assert len(cs) >= 9
    for tag, c in cs.items():
        pred, loss, grads, net = run(c)
        e = (rel_err(pred, c["predict"]), rel_err(loss, c["loss"]))
        assert e[0] <= TOL and e[1] <= 2 * TOL, (tag, e)
        if int(c["cfg"][5]):
            eg = rel_err(grads, c["grads"])
            print("%-28s predict %.2e loss %.2e grads %.2e" % (tag, e[0], e[1], eg))
            assert eg <= TOL, (tag, eg)
        net.close()


def test_batch_is_the_sum_of_its_samples(gf):
    """Three copies of the pair sample and one different pair as one batch: per-sample predictions unchanged, gradient = sum."""
    from graphflow_amd.smp import SMPModel
    cs = cases()
    a, b = cs["physics_pair_omega"], cs["physics_pair_beta"]
    towers, L, Cn, cap, _, _ = (int(x) for x in a["cfg"])
    net = SMPModel(L, Cn, 12, [5, 5])
    p = dev(a["params"])
    g1 = [(a["adj"], a["feature"]), (b["adj"], b["feature"]), (a["adj"], a["feature"])]
    g2 = [(a["adj2"], a["feature2"]), (b["adj2"], b["feature2"]), (a["adj2"], a["feature2"])]
    t = dev(np.array([12.0, 9.0, 3.0]))
    net.prepare(g1, g2)
    pred = net.forward(p, t)[0].clone()
    g = torch.empty(net.n_params, device="cuda")
    net.backward(p, g)
    total = torch.zeros_like(g)
    for i in range(3):
        net.prepare([g1[i]], [g2[i]])
        pi = net.forward(p, t[i:i + 1])[0]
        assert abs(float(pi[0]) - float(pred[i])) <= 1e-6 * max(1.0, abs(float(pred[i])))
        net.backward(p, total, accumulate=True)
    assert rel_err(g.cpu().numpy(), total.cpu().numpy()) <= 2e-6
    with pytest.raises(Exception):
        net.forward(p)
        net.backward(p, g)   # no targets in the last forward


@pytest.mark.parametrize("towers,Cn,nKept", [(1, 16, 0), (2, 8, 0), (1, 48, 0), (2, 16, 9)])
def test_padded_towers_equal_the_towers_at_their_own_widths(gf, monkeypatch, towers, Cn, nKept):
    """Round 4: a tower (channels C, C/2, C/4, ... per level, SMP_omega_physics.h:141-151) is COMPUTED at one padded width (32 / 64) so
    that its levels run the fused level kernels; K_l [18 C_{l-1}][C_l] sits in the corner of a square block, the level features are
    cropped level by level, the feature gradient is padded with zeros (gf_smp_create / gf_smp_backward_features).  Same batch with
    GF_SMP_PAD_CHANNELS=0 (op-by-op levels at the halving widths).  nKept > 0 (SMP_sigma_pairgraphs: RisiContraction_18_dropout masks single
    slices of Q) keeps the levels op by op, so gf_smp_model_create leaves such towers at their own widths (padded: 55 against 34 ms per
    1024-sample step); GF_SMP_PAD_CHANNELS=2 pads them all the same -- the path of a caller who sets dropout masks on a padded tower."""
    from graphflow_amd.smp import SMPModel
    from inputs import synthetic_molecule
    L, cap, F = 3, 10, 5
    mols = [synthetic_molecule(3100 + i, nV=4 + i % 8)[:2] for i in range(20)]
    mols2 = [synthetic_molecule(3200 + i, nV=3 + i % 7)[:2] for i in range(20)]
    tg = dev(np.array([synthetic_molecule(3100 + i)[2] for i in range(20)]))
    got = []
    for mode in ("2" if nKept else "1", "0"):
        monkeypatch.setenv("GF_SMP_PAD_CHANNELS", mode)
        net = SMPModel(L, Cn, cap, [F] * towers, nKept=nKept)
        p = dev(np.random.default_rng(11).uniform(-0.3, 0.3, net.n_params))
        net.prepare(mols, mols2 if towers == 2 else None)
        net.set_mode(True)
        C.CDLL(None).srand(5)
        pred, loss = net.forward(p, tg)
        g = torch.full((net.n_params,), float("nan"), device="cuda")
        net.backward(p, g)
        got.append((pred.cpu().numpy().astype(np.float64), loss.cpu().numpy().astype(np.float64), g.cpu().numpy().astype(np.float64)))
        net.close()
    (p1, l1, g1), (p0, l0, g0) = got
    assert np.isfinite(g1).all() and np.abs(g1).max() > 0
    print("towers %d C %d nKept %d: predict %.2e loss %.2e grads %.2e" % (towers, Cn, nKept, rel_err(p1, p0), rel_err(l1, l0), rel_err(g1, g0)))
    assert rel_err(p1, p0) <= 2e-6 and rel_err(l1, l0) <= 4e-6
    assert rel_err(g1, g0) <= TOL


@pytest.mark.parametrize("train", [True, False])
def test_fused_levels_under_slice_dropout_equal_the_op_by_op_levels(gf, monkeypatch, train):
    """Round 5: RisiContraction_18_dropout on the FUSED level.  A dropped slice k of a node's contraction is a zero factor on the block
    product K^(k) (RisiContraction_18_dropout.h:106-132; test mode scales every slice by nKept / 18, :465-471), so the towers of
    SMP_sigma_pairgraphs keep the fused level kernels with per-product row factors.  Same batch, same rand() sequence, against
    GF_SMP_FUSED_DROPOUT=0 (the levels op by op with the slices of Q zeroed, as rounds 2-4 ran them)."""
    from graphflow_amd.smp import SMPModel
    from inputs import synthetic_molecule
    L, cap, F, Cn, nKept = 3, 10, 5, 16, 7
    mols = [synthetic_molecule(5100 + i, nV=4 + i % 9)[:2] for i in range(24)]
    mols2 = [synthetic_molecule(5200 + i, nV=3 + i % 8)[:2] for i in range(24)]
    tg = dev(np.array([synthetic_molecule(5100 + i)[2] for i in range(24)]))
    got = []
    for mode in ("1", "0"):
        monkeypatch.setenv("GF_SMP_FUSED_DROPOUT", mode)
        net = SMPModel(L, Cn, cap, [F, F], nKept=nKept)
        p = dev(np.random.default_rng(13).uniform(-0.3, 0.3, net.n_params))
        net.prepare(mols, mols2)
        net.set_mode(train)
        C.CDLL(None).srand(9)
        pred, loss = net.forward(p, tg)
        g = torch.full((net.n_params,), float("nan"), device="cuda")
        if train:
            net.backward(p, g)
        elif mode == "1":   # test mode scales the forward only (RisiContraction_18_dropout.h:465-471): the fused level refuses the sweep
            with pytest.raises(Exception, match="test mode"):
                net.backward(p, g)
        got.append((pred.cpu().numpy().astype(np.float64), loss.cpu().numpy().astype(np.float64), g.cpu().numpy().astype(np.float64)))
        net.close()
    (p1, l1, g1), (p0, l0, g0) = got
    if not train:
        print("dropout fused vs op-by-op (test mode): predict %.2e loss %.2e" % (rel_err(p1, p0), rel_err(l1, l0)))
        assert rel_err(p1, p0) <= 2e-6 and rel_err(l1, l0) <= 4e-6
        return
    assert np.isfinite(g1).all() and np.abs(g1).max() > 0
    print("dropout fused vs op-by-op (train %s): predict %.2e loss %.2e grads %.2e" % (train, rel_err(p1, p0), rel_err(l1, l0), rel_err(g1, g0)))
    assert rel_err(p1, p0) <= 2e-6 and rel_err(l1, l0) <= 4e-6
    assert rel_err(g1, g0) <= TOL


@pytest.mark.parametrize("towers,Cn,nKept", [(1, 32, 0), (2, 10, 0), (2, 10, 9), (2, 32, 12)])
def test_towers_with_fields_above_32_stay_on_the_fused_levels(gf, monkeypatch, towers, Cn, nKept):
    """SMP_beta_physics / SMP_beta_pairgraphs have no receptive-field cap (SMP_beta_physics.h): 48-atom graphs reach 36 - 41 positions at level 3.
    Round 6: such a level stays on the fused kernels (smp_fused.hip: big_part; every level of a tower is read out -- the nodes above
    32 positions from their rows).  Against GF_SMP_BIG_FIELDS=0, the op-by-op level of round 5: predictions and every gradient."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from inputs import synthetic_molecule
    from graphflow_amd.smp import SMPModel
    L, cap = 3, 48
    g1 = [synthetic_molecule(sd, nV=48)[:2] for sd in (8017, 8003, 8026)] + [synthetic_molecule(8100 + i)[:2] for i in range(3)]
    g2 = [synthetic_molecule(sd, nV=48)[:2] for sd in (8028, 8000, 8005)] + [synthetic_molecule(8200 + i)[:2] for i in range(3)]
    t = dev(np.arange(1.0, 7.0))

    def step():
        net = SMPModel(L, Cn, cap, [5] * towers, nKept=nKept)
        if nKept:   # (SMP_sigma_pairgraphs: slice dropout -- the same rand() stream for both runs)
            net.set_mode(True)
            C.CDLL(None).srand(4242)
        rng = np.random.default_rng(21)
        p = dev(rng.uniform(-1, 1, net.n_params) / np.sqrt(18 * Cn))
        net.prepare(g1, g2 if towers == 2 else None)
        pred = net.forward(p, t)[0].cpu().numpy().astype(np.float64)
        g = torch.full((net.n_params,), float("nan"), device="cuda")
        net.backward(p, g)
        used = net.device_bytes() if hasattr(net, "device_bytes") else None
        net.close()
        return pred, g.cpu().numpy().astype(np.float64), used

    a = step()
    monkeypatch.setenv("GF_SMP_BIG_FIELDS", "0")
    b = step()
    e = (rel_err(a[0], b[0]), rel_err(a[1], b[1]))
    print("towers %d C %d nKept %d, fields above 32: fused vs op-by-op level: predict %.2e grads %.2e" % (towers, Cn, nKept, e[0], e[1]))
    assert np.isfinite(a[1]).all() and np.abs(a[1]).max() > 0
    assert e[0] <= TOL and e[1] <= 2 * TOL, e
    assert not np.array_equal(a[1], b[1])   # (really two different paths)
