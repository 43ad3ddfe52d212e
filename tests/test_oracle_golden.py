"""Pin the oracle (oracle/*) against fixtures the REFERENCE produced (tests/golden/make_golden.py).

CPU only.  Tolerances: the oracle and the reference both run PyTorch-CPU fp32 but batch
differently (the reference runs the TDNN at batch 1 per utterance), so values
agree to fp32 round-off, not bit-for-bit; integer outputs (decisions, success flags) must be equal.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, weights_checksum
from oracle import attacks as oatk
from oracle import xv_plda as oxv
from toy_model import ToyModel


@pytest.fixture(scope="module")
def models(xv_weights):
    return {
        "fast": oxv.XvPlda(xv_weights, faithful=False),
        "faithful": oxv.XvPlda(xv_weights, faithful=True),
    }


def test_weights_checksum(xv_weights):
    g = load_golden("xv_f300.npz")
    assert weights_checksum(xv_weights) == g["meta"]["weights_sha256"], "synthetic weight generator drifted"


@pytest.mark.parametrize("tag", ["f300", "f331"])
@pytest.mark.parametrize("style", ["fast", "faithful"])
def test_xv_forward_backward(models, tag, style):
    g = load_golden("xv_%s.npz" % tag)
    m = models[style]
    feats = torch.from_numpy(g["feats"]).requires_grad_(True)
    y = torch.from_numpy(g["y"])
    cm = m.cmvn(feats)
    np.testing.assert_allclose(cm.detach().numpy(), g["cmvn"], rtol=0, atol=2e-5)
    layers = m.tdnn_layers(cm[-1:].transpose(1, 2))
    for i, (a, _) in enumerate(layers, 1):
        np.testing.assert_allclose(a[0, ::37, ::11].detach().numpy(), g["relu%d_sub" % i], rtol=1e-4, atol=1e-5)
        assert abs(a.double().sum().item() - g["relu%d_sum" % i][0]) <= 1e-5 * g["relu%d_sum" % i][1]
    temb = m.tdnn_embedding(cm.transpose(1, 2))
    np.testing.assert_allclose(temb.detach().numpy(), g["tdnn_emb"], rtol=1e-4, atol=1e-5)
    dec, scores = m.make_decision(feats, flag=1)
    np.testing.assert_allclose(m.embedding(feats, flag=1).detach().numpy(), g["emb"], rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(scores.detach().numpy(), g["scores"], rtol=1e-3, atol=5e-2)
    assert dec.tolist() == g["decisions"].tolist()
    ce = oatk.cross_entropy_loss(scores, y)
    np.testing.assert_allclose(ce.detach().numpy(), g["ce"], rtol=1e-3, atol=5e-2)
    ce.backward(torch.ones_like(ce))
    gscale = np.abs(g["grad_ce"]).max()
    np.testing.assert_allclose(feats.grad.numpy(), g["grad_ce"], rtol=0, atol=2e-3 * gscale)
    feats.grad = None
    _, scores = m.make_decision(feats, flag=1)
    mg = oatk.margin_loss(scores, y, targeted=False, task="CSI", clip_max=False)
    np.testing.assert_allclose(mg.detach().numpy(), g["margin"], rtol=1e-3, atol=5e-2)
    mg.backward(torch.ones_like(mg))
    gscale = np.abs(g["grad_margin"]).max()
    np.testing.assert_allclose(feats.grad.numpy(), g["grad_margin"], rtol=0, atol=2e-3 * gscale)


def test_xv_threshold_and_margin_variants(xv_weights):
    g = load_golden("xv_thresh.npz")
    m = oxv.XvPlda(xv_weights, threshold=g["meta"]["threshold"])
    dec, scores = m.make_decision(torch.from_numpy(g["feats"]), flag=1)
    assert dec.tolist() == g["decisions"].tolist()
    assert -1 in dec.tolist() and max(dec.tolist()) >= 0
    ref_scores = torch.from_numpy(g["scores"])
    y = torch.from_numpy(g["y"])
    thr = g["meta"]["threshold"]
    for task in ("CSI", "OSI"):
        for targeted in (False, True):
            for clip in (False, True):
                l = oatk.margin_loss(ref_scores, y, targeted, 0.5, task, thr, clip)
                np.testing.assert_allclose(l.numpy(), g["margin_%s_%d_%d" % (task, targeted, clip)], rtol=1e-6, atol=1e-5)
    ysv = torch.from_numpy(g["ysv"])
    for targeted in (False, True):
        l = oatk.margin_loss(ref_scores[:, :1], ysv, targeted, 0.5, "SV", thr, False)
        np.testing.assert_allclose(l.numpy(), g["margin_SV_%d" % targeted], rtol=1e-6, atol=1e-5)


class _FeatAdapter:
    def __init__(self, model, F, scale):
        self.model, self.F, self.scale, self.threshold = model, F, scale, model.threshold

    def make_decision(self, x):
        return self.model.make_decision(x.view(x.shape[0], self.F, 30) * self.scale, flag=1)


def test_xv_pgd_feature_level(models):
    """PGD / CWinf driven through the reference xv_plda from flag=1 (A3-A14 end to end)."""
    g = load_golden("xv_pgd_featlevel.npz")
    x0 = torch.from_numpy(g["x0"])
    adapter = _FeatAdapter(models["fast"], 300, g["meta"]["scale"])
    d0, s0 = adapter.make_decision(x0)
    assert d0.tolist() == g["clean_decisions"].tolist()
    for name, cls, kw in (("pgd_ce", oatk.PGD, dict(loss="Entropy")),
                          ("pgd_ce_t", oatk.PGD, dict(loss="Entropy", targeted=True)),
                          ("cwinf", oatk.CWinf, dict())):
        atk = cls(adapter, task="CSI", epsilon=g["meta"]["eps"], step_size=g["meta"]["step"],
                  max_iter=g["meta"]["max_iter"], batch_size=3, **kw)
        adv, success = atk.attack(x0.clone(), torch.from_numpy(g[name + "_y"]))
        ref = g[name + "_adv"]
        # sign() turns fp32 noise on near-zero gradient entries into +-step flips:
        # require almost all samples identical and the rest within the epsilon ball.
        diff = np.abs(adv.numpy() - ref)
        assert (diff > 1e-6).mean() < 0.05, name
        assert diff.max() <= 2 * g["meta"]["eps"] + 1e-6
        assert list(success) == g[name + "_success"].tolist()
        d1, s1 = adapter.make_decision(adv)
        assert d1.tolist() == g[name + "_decisions"].tolist()
        np.testing.assert_allclose(s1.detach().numpy(), g[name + "_scores"], rtol=1e-2, atol=0.5)


@pytest.mark.parametrize("tag,thr,task", [("csi", None, "CSI"), ("osi", 1.5, "OSI")])
def test_attack_logic_on_toy_model(tag, thr, task):
    g = load_golden("attack_toy.npz")
    x = torch.from_numpy(g["x"])
    model = ToyModel(threshold=thr).eval()
    for p in model.parameters():
        p.requires_grad_(False)
    d0, s0 = model.make_decision(x)
    assert d0.tolist() == g["%s_clean_dec" % tag].tolist()

    def check(name, atk, exact=True):
        torch.manual_seed(123)
        np.random.seed(123)
        y = torch.from_numpy(g["%s_%s_y" % (tag, name)])
        adv, success = atk.attack(x.clone(), y)
        assert list(success) == g["%s_%s_success" % (tag, name)].tolist(), name
        np.testing.assert_allclose(adv.detach().numpy(), g["%s_%s_adv" % (tag, name)], rtol=0,
                                   atol=1e-6 if exact else 2e-4, err_msg=name)

    check("fgsm", oatk.FGSM(model, task=task, epsilon=0.01, batch_size=4))
    check("pgd", oatk.PGD(model, task=task, epsilon=0.01, step_size=0.002, max_iter=8, batch_size=3))
    check("pgd_t", oatk.PGD(model, task=task, epsilon=0.01, step_size=0.002, max_iter=8, batch_size=2, targeted=True))
    check("pgd_eot", oatk.PGD(model, task=task, epsilon=0.01, step_size=0.002, max_iter=4, batch_size=4,
                              EOT_size=4, EOT_batch_size=2))
    check("pgd_rand", oatk.PGD(model, task=task, epsilon=0.01, step_size=0.002, max_iter=4, batch_size=4,
                               num_random_init=3))
    check("cwinf", oatk.CWinf(model, task=task, epsilon=0.01, step_size=0.002, max_iter=8, batch_size=4))
    check("cw2", oatk.CW2(model, task=task, initial_const=0.5, binary_search_steps=4, max_iter=30, stop_early=True,
                          stop_early_iter=10, lr=5e-3, batch_size=4), exact=False)
    check("cw2_t", oatk.CW2(model, task=task, initial_const=0.5, binary_search_steps=3, max_iter=25, stop_early=False,
                            lr=5e-3, batch_size=2, targeted=True, confidence=0.1), exact=False)
    fb = dict(task=task, epsilon=0.02, max_iter=30, max_lr=0.004, min_lr=1e-4, samples_per_draw=16,
              samples_per_draw_batch_size=8, sigma=0.01, stop_early=True, stop_early_iter=10)
    if thr is not None:
        fb["threshold"] = thr
    check("fakebob", oatk.FAKEBOB(model, batch_size=1, **fb))
    check("fakebob_t", oatk.FAKEBOB(model, batch_size=1, targeted=True, confidence=0.05, **fb))


def test_estimate_threshold_matches_reference():
    """FAKEBOB.estimate_threshold (FAKEBOB.py:210-295), incl. the whole-batch / example-0 quirk."""
    import json
    g = load_golden("estimate_threshold.npz")
    x = torch.from_numpy(g["x"])
    kw = json.loads(g["meta"]["kw"])
    kw.pop("verbose", None)
    for name in ("single", "batch_quirk", "accepted", "negative"):
        model = ToyModel(threshold=float(g[name + "_model_threshold"])).eval()
        atk = oatk.FAKEBOB(model, **kw)
        torch.manual_seed(321)
        np.random.seed(321)
        est = atk.estimate_threshold(x[g[name + "_idx"].tolist()].clone(), step=g["meta"]["step"])
        want = float(g[name + "_estimate"])
        if np.isnan(want):
            assert est is None and atk.threshold is None
        else:
            assert abs(est - want) < 1e-5, (name, est, want)
            # the estimate is the first accepted top score: just above the model's true threshold
            assert est > float(g[name + "_model_threshold"])
