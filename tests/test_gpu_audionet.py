"""GPU parity tests of the AudioNet CSI-NE path: against the oracle (oracle/audionet.py) and, since round 2,
directly against tests/golden/an_ref.npz -- outputs of the reference's own audionet_csine / Preprocessor code run in
the build container with two disclosed harness accommodations (tests/golden/make_golden_frontends.py: third-party
mel basis in place of the uninstalled librosa, pre-1.8 torch.stft return convention).  The oracle itself reproduces
that fixture to round-off (tests/test_oracle_frontends.py).
"""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
LOG = os.path.join(ROOT, "gpurun_out", "parity_log.txt")


def log(msg):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, "a") as f:
        f.write(msg + "\n")
    print(msg)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def sd():
    from speakerguard_amd import synth
    return synth.make_audionet_state_dict(seed=0, num_class=251)


@pytest.fixture(scope="module")
def hip(sd, dev):
    from speakerguard_amd.model.audionet_csine import audionet_csine
    m = audionet_csine.from_weights(sd, device=dev)
    cfg = os.environ.get("SG_AN_TEST_CFG")  # A/B runs of this file: "fft_bits,spectrum_cache,fused_overlap_add" (default: the library's)
    if cfg:
        bits, cache, ola = (int(v) for v in cfg.split(","))
        m.configure_frontend(bits, bool(cache), bool(ola))
    return m


@pytest.fixture(scope="module")
def ora(sd):
    from oracle.audionet import AudioNet
    return AudioNet(sd)


@pytest.mark.parametrize("T", [48000, 16000, 20011])
def test_logmel_matches_oracle(hip, ora, dev, T):
    from speakerguard_amd import synth
    x = torch.from_numpy(synth.make_waveforms(3, T, seed=41))
    got = hip.compute_feat(x.to(dev)).cpu()
    want = ora.compute_feat(x)
    assert got.shape == want.shape
    log("audionet log-mel T=%d: max abs err %.3e dB (values %.1f..%.1f)" % (T, (got - want).abs().max().item(), want.min().item(), want.max().item()))
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=2e-3)
    # int16-scaled input is divided by 32768 (check_input_range, range_type='scale')
    got16 = hip.compute_feat((x * 32768.0).to(dev)).cpu()
    np.testing.assert_allclose(got16.numpy(), want.numpy(), rtol=0, atol=2e-3)


def test_layers_scores_and_decisions(hip, ora, dev):
    from speakerguard_amd import synth
    x = torch.from_numpy(synth.make_waveforms(4, 48000, seed=42))
    with torch.no_grad():
        feats = ora.compute_feat(x)
        outs = ora.layers(feats)
        odec, oscores = ora.make_decision(x)
    dec, scores = hip.make_decision(feats.to(dev), flag=1)  # same features -> isolates the CNN
    for i, o in enumerate(outs, 1):
        act = hip.read_activation(i, 4).cpu().numpy()  # (B, rows, C)
        ref = o.numpy().transpose(0, 2, 1)
        assert act.shape == ref.shape, (i, act.shape, ref.shape)
        log("audionet layer %d: max abs err %.3e (max %.2f)" % (i, np.abs(act - ref).max(), np.abs(ref).max()))
        np.testing.assert_allclose(act, ref, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(scores.cpu().numpy(), oscores.numpy(), rtol=1e-3, atol=2e-3)
    assert dec.cpu().tolist() == odec.tolist()
    dec2, scores2 = hip.make_decision(x.to(dev))
    np.testing.assert_allclose(scores2.cpu().numpy(), oscores.numpy(), rtol=1e-3, atol=5e-3)
    assert dec2.cpu().tolist() == odec.tolist()
    emb = hip.embedding(x.to(dev)).cpu()
    np.testing.assert_allclose(emb.numpy(), outs[-1].max(2)[0].numpy(), rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("loss_name", ["Entropy", "Margin"])
def test_gradients_match_oracle_autograd(hip, ora, dev, loss_name):
    from oracle import attacks as oatk
    from speakerguard_amd import synth
    from speakerguard_amd.attack.utils import resolve_loss
    x = torch.from_numpy(synth.make_waveforms(3, 48000, seed=43))
    with torch.no_grad():
        y = ora.make_decision(x)[0]
        feats = ora.compute_feat(x)
    spec, _ = resolve_loss(loss_name, False, 0., "CSI", None, False)
    ofn, _ = oatk.resolve_loss(loss_name, False, 0., "CSI", None, False)
    # feature level
    fin = feats.clone().requires_grad_(True)
    _, sc = ora.make_decision(fin, flag=1)
    lo = ofn(sc, y)
    lo.backward(torch.ones_like(lo))
    dec, scores, loss, grad = hip.loss_grad(feats.to(dev), y.to(dev), spec, flag=1)
    np.testing.assert_allclose(loss.cpu().numpy(), lo.detach().numpy(), rtol=1e-3, atol=2e-3)
    gs = fin.grad.abs().max().item()
    e = (grad.cpu() - fin.grad).abs().max().item() / gs
    log("audionet d loss/d logmel (%s): max err / max|grad| = %.3e" % (loss_name, e))
    assert e < 2e-3
    # waveform level
    xin = x.clone().requires_grad_(True)
    _, sc = ora.make_decision(xin)
    lo = ofn(sc, y)
    lo.backward(torch.ones_like(lo))
    dec, scores, loss, grad = hip.loss_grad(x.to(dev), y.to(dev), spec)
    want, got = xin.grad.numpy(), grad.cpu().numpy()
    gs = np.abs(want).max()
    e = np.abs(got - want).max() / gs
    sm = float((np.sign(got) != np.sign(want)).mean())
    log("audionet d loss/d wav (%s): max err / max|grad| = %.3e, sign mismatch %.3e" % (loss_name, e, sm))
    assert e < 5e-3 and sm < 5e-3


def test_fused_loop_equals_stepwise_and_fgsm_config(hip, ora, dev):
    """BASELINE configs[0]: FGSM 1-step L-inf on AudioNet CSI-NE, one 3 s utterance."""
    from oracle import attacks as oatk
    from speakerguard_amd import synth
    from speakerguard_amd.attack.FGSM import FGSM
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    x = torch.from_numpy(synth.make_waveforms(1, 48000, seed=44))
    with torch.no_grad():
        y = ora.make_decision(x)[0]
    oadv, osucc = oatk.FGSM(ora, task="CSI", epsilon=0.002, batch_size=1).attack(x.clone(), y)
    adv, succ = FGSM(hip, task="CSI", epsilon=0.002, batch_size=1, verbose=0).attack(x.to(dev), y.to(dev))
    diff = (adv.cpu() - oadv).abs()
    frac = float((diff > 1e-7).float().mean())
    log("audionet FGSM: samples differing %.4f%%, success hip=%s oracle=%s" % (100 * frac, succ, osucc))
    assert frac < 0.01 and succ == osucc
    # fused == stepwise
    xb = torch.from_numpy(synth.make_waveforms(3, 32000, seed=45)).to(dev)
    yb = hip.make_decision(xb)[0]
    lower, upper = torch.clamp(xb - 0.002, min=-1), torch.clamp(xb + 0.002, max=1)
    spec = SEC4SR_CrossEntropy()
    xa, success, dec, scores, loss, _, _ = hip.pgd_run(xb, yb, lower, upper, spec, 0.0004, 4, 1)
    xs = xb.clone()
    for _ in range(4):
        _, _, _, g = hip.loss_grad(xs, yb, spec)
        hip.pgd_update(xs, g, lower, upper, 0.0004, 1)
    d2, s2, l2, _ = hip.loss_grad(xs, yb, spec, want_grad=False)
    assert torch.equal(xa, xs) and torch.equal(dec, d2) and torch.equal(scores, s2)


def test_pgd_matches_oracle(hip, ora, dev):
    from oracle import attacks as oatk
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    B, iters = 4, 5
    x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=46))
    with torch.no_grad():
        y = ora.make_decision(x)[0]
    kw = dict(task="CSI", epsilon=0.002, step_size=0.0004, max_iter=iters, batch_size=B)
    oadv, osucc = oatk.PGD(ora, **kw).attack(x.clone(), y)
    adv, succ = PGD(hip, verbose=0, **kw).attack(x.to(dev), y.to(dev))
    diff = (adv.cpu() - oadv).abs()
    frac = float((diff > 1e-7).float().mean())
    log("audionet PGD-%d: samples differing %.3f%%, success hip=%s oracle=%s" % (iters, 100 * frac, succ, osucc))
    assert frac < 0.02 * iters and diff.max().item() <= 0.004 + 1e-6
    assert succ == osucc
    with torch.no_grad():
        assert hip.make_decision(adv)[0].cpu().tolist() == ora.make_decision(oadv)[0].tolist()


def test_full_batch_properties(hip, dev):
    """B=64 (one GPU's shard of configs[3]): determinism and shard invariance."""
    from speakerguard_amd import synth
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    x = torch.from_numpy(synth.make_waveforms(64, 48000, seed=47)).to(dev)
    y = hip.make_decision(x)[0]
    lower, upper = torch.clamp(x - 0.002, min=-1), torch.clamp(x + 0.002, max=1)
    spec = SEC4SR_CrossEntropy()
    run = lambda sl: hip.pgd_run(x[sl], y[sl], lower[sl], upper[sl], spec, 0.0004, 3, 1)
    a, b = run(slice(0, 64)), run(slice(0, 64))
    assert torch.equal(a[0], b[0])
    lo, hi = run(slice(0, 32)), run(slice(32, 64))
    assert torch.equal(a[0], torch.cat((lo[0], hi[0])))
    assert (a[0] - x).abs().max().item() <= 0.002 + 1e-7
    log("audionet full-size PGD-3: successes %d/64" % int(a[1].sum()))


@pytest.mark.parametrize("tag", ["t48000", "t20011", "tones"])
def test_against_reference_run_fixture(hip, dev, tag):
    """HIP path vs the reference's own code (an_ref.npz): log-mel, logits, decisions, CE loss, d loss/d log-mel and
    d loss/d waveform as the reference's autograd produced them."""
    import hashlib

    from conftest import load_golden
    from speakerguard_amd import synth
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    g = load_golden("an_ref.npz")
    B, T, seed = (int(v) for v in g[tag + "_gen"])
    x = synth.make_tone_waveforms(B, T, seed) if tag == "tones" else synth.make_waveforms(B, T, seed=seed)
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g[tag + "_x_sha256"])
    x = torch.from_numpy(x).to(dev)
    feats = hip.compute_feat(x).cpu().numpy()
    e_f = np.abs(feats - g[tag + "_feats"]).max()
    assert e_f < 2e-3, e_f                                                         # dB
    dec, scores = hip.make_decision(x)
    assert dec.cpu().tolist() == g[tag + "_decisions"].tolist()                    # bit-exact IDs
    e_s = np.abs(scores.cpu().numpy() - g[tag + "_scores"]).max()
    assert e_s < 5e-3, e_s
    y = torch.from_numpy(g[tag + "_y"]).to(dev)
    spec = SEC4SR_CrossEntropy()
    _, _, loss, grad = hip.loss_grad(x, y, spec)
    np.testing.assert_allclose(loss.cpu().numpy(), g[tag + "_ce"], rtol=1e-3, atol=5e-3)
    got, ref = grad.cpu().numpy()[..., ::3], g[tag + "_grad_wav_sub3"]
    e_g = np.abs(got - ref).max() / np.abs(ref).max()
    sm = float((np.sign(got) != np.sign(ref)).mean())
    assert e_g < 5e-3 and sm < 5e-3, (e_g, sm)
    f = torch.from_numpy(g[tag + "_feats"]).to(dev)
    d1, s1, l1, gf = hip.loss_grad(f, y, spec, flag=1)
    e_sf = np.abs(s1.cpu().numpy() - g[tag + "_scores_from_feats"]).max()
    e_gf = np.abs(gf.cpu().numpy() - g[tag + "_grad_feats"]).max() / np.abs(g[tag + "_grad_feats"]).max()
    assert e_sf < 2e-3 and e_gf < 2e-3, (e_sf, e_gf)
    log("audionet vs REFERENCE run (%s): log-mel %.2e dB, logits %.2e, decisions equal, d/d wav %.2e of max (sign mismatch "
        "%.2e), d/d log-mel %.2e" % (tag, e_f, e_s, e_g, sm, e_gf))


def test_int16_scaled_input_against_reference_run(hip, dev):
    from conftest import load_golden
    from speakerguard_amd import synth
    g = load_golden("an_ref.npz")
    x16 = torch.from_numpy(synth.make_waveforms(2, 48000, seed=63) * 32768.0).to(dev)
    dec, scores = hip.make_decision(x16)
    assert dec.cpu().tolist() == g["int16_decisions"].tolist()
    np.testing.assert_allclose(scores.cpu().numpy(), g["int16_scores"], rtol=1e-3, atol=5e-3)


def test_logmel_backward_reusing_the_forward_pass(hip, dev):
    """sg_an_logmel_backward(reuse_forward=1) starts from the mel energies the last sg_an_logmel left in the workspace;
    it must give what the recomputing form gives, and must fall back to recomputing when pointer or shape differ."""
    from speakerguard_amd import _native as N
    from speakerguard_amd import synth
    x = torch.from_numpy(synth.make_waveforms(3, 32000, seed=48)).to(dev)
    other = torch.from_numpy(synth.make_waveforms(3, 32000, seed=49)).to(dev)
    feats = hip.compute_feat(x)
    dfe = torch.randn_like(feats)

    def bwd(t, reuse):
        g = torch.empty_like(t)
        hip.ctx.call("sg_an_logmel_backward", N._ptr(t), 3, 32000, N._ptr(dfe), N._ptr(g), reuse, N.current_stream_ptr(dev))
        return g

    hip.compute_feat(x)
    g_reuse = bwd(x, 1)
    g_plain = bwd(x, 0)
    # the same an_frame_forward instantiation under fp contract(off) in both kernels: the same bits (round 6: tools/an_reuse_probe.py
    # finds 0 differing samples in all four front-end configurations; the 1e-5 bound of round 5 was left over from an
    # intermediate build)
    assert torch.equal(g_reuse, g_plain)
    # the cache belongs to x: a backward for ANOTHER tensor must not use it even when asked to
    g_other = bwd(other, 1)
    assert torch.equal(g_other, bwd(other, 0))
    assert not torch.allclose(g_other, g_plain)


@pytest.mark.parametrize("B,T", [(3, 48000), (2, 20011), (64, 48000), (1, 16000), (5, 65000)])
def test_fused_cnn_equals_the_per_layer_sequence_bit_for_bit(hip, dev, monkeypatch, B, T):
    """Round 4: the AudioNet CNN of a pass as ONE launch per direction (k_audionet_fused.hip: LDS-resident
    activations, time slices with recomputed halos) against the ~22 per-layer launches of rounds 1-3 (SG_AN_FUSED=0), which
    stay in the library as its counterpart.  Same fmaf chain per output element: every activation, the logits, the
    decisions, the loss and d loss/d waveform and d loss/d log-mel must be EQUAL, for the planner's cut and for forced cuts
    into 1 / 2 / 3 / 5 / 7 time slices (the result does not depend on the cut, i.e. on the batch size either)."""
    from speakerguard_amd import synth
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    x = torch.from_numpy(synth.make_waveforms(B, T, seed=60 + B)).to(dev)
    y = (torch.arange(B, device=dev) * 7) % 251
    ce = SEC4SR_CrossEntropy()

    def run():
        dec, sc, ls, g = hip.loss_grad(x, y, ce)
        acts = [hip.read_activation(i, B).clone() for i in range(1, 9)]
        feats = hip.compute_feat(x)
        d1, s1, l1, g1 = hip.loss_grad(feats, y, ce, flag=1)
        return [dec, sc, ls, g, d1, s1, l1, g1] + acts

    monkeypatch.setenv("SG_AN_FUSED", "0")
    ref = run()
    monkeypatch.setenv("SG_AN_FUSED", "1")
    names = ["decisions", "scores", "loss", "d/d wav", "decisions(feat)", "scores(feat)", "loss(feat)", "d/d log-mel"] + ["layer %d" % i for i in range(1, 9)]
    for slices in (0, 1, 2, 3, 5, 7):
        if slices:
            monkeypatch.setenv("SG_AN_SLICES", str(slices))
        got = run()
        for n, a, b in zip(names, got, ref):
            assert a.shape == b.shape, (n, slices)
            assert torch.equal(a, b), "%s differs with %s slices: max |diff| %.3e" % (n, slices or "planned", (a.float() - b.float()).abs().max().item())
    monkeypatch.delenv("SG_AN_SLICES")
    assert float(ref[3].abs().max()) > 0
    log("audionet fused CNN (B=%d, T=%d): forward activations, logits, loss, d/d wav, d/d log-mel equal the per-layer sequence bit for bit "
        "(planned cut and 1/2/3/5/7 slices)" % (B, T))


@pytest.mark.parametrize("B,T", [(3, 48000), (2, 20011), (5, 40000)])  # (whole utterances fit a block's LDS up to ~3.2 s)
def test_head_in_the_backward_and_the_one_launch_form_equal_the_separate_launches(hip, dev, monkeypatch, B, T):
    """Round 6: when a gradient follows, the network's head (max over time, fc, loss, d loss / d conv8) runs inside the fused
    backward launch, and with whole utterances per block (S = 1, forced here through SG_AN_SLICES=1 -- the planner takes it from
    ~256 utterances) forward + head + backward are ONE launch.  Both against the three separate launches (SG_AN_HEAD=0): scores,
    decisions, loss, d loss / d wav, d loss / d log-mel bit for bit, and the stage trace names what ran."""
    from speakerguard_amd import synth
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy, SEC4SR_MarginLoss
    x = torch.from_numpy(synth.make_waveforms(B, T, seed=160 + B)).to(dev)
    y = (torch.arange(B, device=dev) * 11) % 251
    feats = hip.compute_feat(x)
    for spec in (SEC4SR_CrossEntropy(), SEC4SR_MarginLoss(targeted=False, task="CSI")):
        def run():
            return list(hip.loss_grad(x, y, spec)) + list(hip.loss_grad(feats, y, spec, flag=1))

        monkeypatch.setenv("SG_AN_HEAD", "0")
        monkeypatch.delenv("SG_AN_SLICES", raising=False)
        ref = run()
        tags = [t for t, _ in hip.trace_stages(lambda: hip.loss_grad(x, y, spec), max_records=256)]
        assert tags.count("an_tail") == 1 and tags.count("an_cnn_bwd") == 1
        monkeypatch.setenv("SG_AN_HEAD", "1")
        for slices, one in ((0, "1"), (2, "1"), (1, "0"), (1, "1")):
            monkeypatch.setenv("SG_AN_ONE", one)
            if slices:
                monkeypatch.setenv("SG_AN_SLICES", str(slices))
            got = run()
            for a, b in zip(got, ref):
                assert torch.equal(a, b), (B, T, slices, one, (a.float() - b.float()).abs().max().item())
            tags = [t for t, _ in hip.trace_stages(lambda: hip.loss_grad(x, y, spec), max_records=256)]
            assert "an_tail" not in tags
            if slices == 1 and one == "1":
                assert tags.count("an_cnn_fwdbwd") == 1 and "an_cnn_fwd" not in tags and "an_cnn_bwd" not in tags, tags
            else:
                assert tags.count("an_cnn_fwd") == 1 and tags.count("an_cnn_bwd") == 1, tags
        monkeypatch.delenv("SG_AN_SLICES", raising=False)
        monkeypatch.delenv("SG_AN_ONE", raising=False)
    log("audionet B=%d T=%d: head inside the fused backward launch, and forward + head + backward as one launch (whole utterances per block), "
        "equal the three separate launches bit for bit (cross-entropy and margin loss)" % (B, T))


def test_pgd_loops_with_the_head_inside_equal_the_separate_launches(hip, dev, monkeypatch):
    """The device loops (sg_an_pgd_run) with the head inside the backward launch / the one-launch form against the separate
    launches: adversarial audio, flags, decisions, per-step loss and decision records equal."""
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    x = torch.from_numpy(synth.make_waveforms(6, 32000, seed=171)).to(dev)
    y = hip.make_decision(x)[0]

    def run():
        atk = PGD(hip, task="CSI", epsilon=0.002, step_size=0.0004, max_iter=5, batch_size=6, verbose=0)
        adv, succ = atk.attack(x, y)
        return adv.clone(), list(succ)

    monkeypatch.setenv("SG_AN_HEAD", "0")
    ref = run()
    for head, slices, one in (("1", None, "1"), ("1", "1", "1"), ("1", "1", "0")):
        monkeypatch.setenv("SG_AN_HEAD", head)
        monkeypatch.setenv("SG_AN_ONE", one)
        if slices:
            monkeypatch.setenv("SG_AN_SLICES", slices)
        got = run()
        assert torch.equal(got[0], ref[0]) and got[1] == ref[1], (head, slices, one)
        tags = [t for t, _ in hip.trace_stages(lambda: run(), max_records=1024)]
        assert ("an_cnn_fwdbwd" in tags) == (slices == "1" and one == "1"), (tags, slices, one)
        monkeypatch.delenv("SG_AN_SLICES", raising=False)
    monkeypatch.delenv("SG_AN_HEAD", raising=False)
    monkeypatch.delenv("SG_AN_ONE", raising=False)


def test_fused_cnn_is_the_path_that_runs(hip, dev, monkeypatch):
    """The stage trace names what ran: one fused launch per direction and no per-layer contraction with the default
    setting; the per-layer tags with SG_AN_FUSED=0."""
    from speakerguard_amd import _native as N
    from speakerguard_amd import synth
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    x = torch.from_numpy(synth.make_waveforms(4, 48000, seed=77)).to(dev)
    y = torch.zeros(4, dtype=torch.int64, device=dev)
    for fused in (1, 0):
        monkeypatch.setenv("SG_AN_FUSED", str(fused))
        recs = hip.trace_stages(lambda: hip.loss_grad(x, y, SEC4SR_CrossEntropy()), max_records=256)
        tags = [t for t, _ in recs]
        if fused:
            assert tags.count("an_cnn_fwd") == 1 and tags.count("an_cnn_bwd") == 1 and not any(t.startswith("an_conv") for t in tags), tags
        else:
            assert "an_cnn_fwd" not in tags and sum(t.startswith("an_conv") for t in tags) == 14, tags


def test_fused_cnn_random_shapes(hip, dev, monkeypatch):
    """Shapes the planner was not tuned on: random utterance lengths from just above the shortest the stack accepts (conv8
    needs 3 frames: 24 log-mel frames) to 9 s, batches 1..7, random forced cuts -- fused kernels == per-layer sequence, bit
    for bit, forward and gradient; and an utterance too long for the LDS-resident form falls back to the per-layer path."""
    from speakerguard_amd import synth
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    rs = np.random.RandomState(404)
    ce = SEC4SR_CrossEntropy()
    shapes = [(1, 160 * 23 + 1), (2, 160 * 24 + 7)] + [(int(rs.randint(1, 8)), int(rs.randint(4000, 144000))) for _ in range(10)]
    for B, T in shapes:
        x = torch.from_numpy(synth.make_waveforms(B, T, seed=int(rs.randint(1 << 20)))).to(dev)
        y = torch.from_numpy(rs.randint(0, 251, size=B)).to(dev)
        monkeypatch.setenv("SG_AN_FUSED", "0")
        monkeypatch.delenv("SG_AN_SLICES", raising=False)
        ref = hip.loss_grad(x, y, ce)
        ref_act = [hip.read_activation(i, B).clone() for i in (1, 2, 5, 8)]
        monkeypatch.setenv("SG_AN_FUSED", "1")
        for slices in (0, int(rs.randint(1, 12))):
            if slices:
                monkeypatch.setenv("SG_AN_SLICES", str(slices))
            got = hip.loss_grad(x, y, ce)
            for a, b in zip(got, ref):
                assert torch.equal(a, b), (B, T, slices)
            for i, b in zip((1, 2, 5, 8), ref_act):
                assert torch.equal(hip.read_activation(i, B), b), (B, T, slices, i)
    monkeypatch.delenv("SG_AN_SLICES", raising=False)
    # 3 minutes of audio: 18 000 frames do not fit the LDS-resident form even in 16 slices -> the per-layer sequence runs
    x = torch.from_numpy(synth.make_waveforms(1, 16000 * 180, seed=5)).to(dev)
    tags = [t for t, _ in hip.trace_stages(lambda: hip.make_decision(x), max_records=64)]
    assert "an_cnn_fwd" not in tags and any(t.startswith("an_conv") for t in tags), tags
    log("audionet fused CNN: %d random shapes (B 1..7, 0.25..9 s, random cuts) equal the per-layer sequence bit for bit; 180 s falls back" % len(shapes))


def test_frontend_picks_the_fused_overlap_add_for_large_batches(sd, dev):
    """sg_an_configure's default (fused_overlap_add = -1): the library takes the overlap-add inside the adjoint where cutting
    utterances into runs (5 halo frames each) costs little -- 256 utterances of 3 s --, the separate pair for a handful; the
    iterate it steps to is the same either way (the runs here are real cuts: 12 per utterance)."""
    from speakerguard_amd import synth
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    from speakerguard_amd.model.audionet_csine import audionet_csine
    m = audionet_csine.from_weights(sd, device=dev)
    spec = SEC4SR_CrossEntropy()
    got = {}
    for B in (4, 256):
        x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=70 + B)).to(dev)
        y = m.make_decision(x)[0]
        lower, upper = torch.clamp(x - 0.002, min=-1), torch.clamp(x + 0.002, max=1)
        for ola in (None, False, True):
            m.configure_frontend(32, True, ola)
            tags = [t for t, _ in m.trace_stages(lambda: got.__setitem__((B, ola), m.pgd_run(x, y, lower, upper, spec, 0.0004, 2, 1)[0]),
                                                 max_records=128)]
            fused = "an_logmel_bwd" in tags and "an_overlap_add" not in tags
            assert "an_logmel_bwd" in tags, tags
            assert fused == (ola is True or (ola is None and B == 256)), (B, ola, tags)
        assert torch.equal(got[(B, None)], got[(B, False)]) and torch.equal(got[(B, None)], got[(B, True)]), B
    m.configure_frontend()


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("B,T", [(3, 48000), (2, 20011), (1, 16000), (5, 4000), (2, 6000), (9, 65000)])
def test_overlap_add_inside_the_adjoint_equals_the_separate_pair(sd, dev, bits, B, T):
    """Round 5: the log-mel adjoint adds the frames' gradients up on chip (an_logmel_bwd_ola_kernel + an_edge_to_wave_kernel)
    instead of writing them out for an_frames_to_wave_kernel -- same sums in the same order: d loss / d wav and the stepped
    iterate of the device loop are EQUAL, whatever the utterance length (short ones are all 'edge'), the slice cut and the
    transform precision; in float32 the spectrum cache holds exactly what the backward would recompute."""
    from speakerguard_amd import synth
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    from speakerguard_amd.model.audionet_csine import audionet_csine
    m = audionet_csine.from_weights(sd, device=dev)
    x = torch.from_numpy(synth.make_waveforms(B, T, seed=90 + B)).to(dev)
    y = m.make_decision(x)[0]
    lower, upper = torch.clamp(x - 0.002, min=-1), torch.clamp(x + 0.002, max=1)
    spec = SEC4SR_CrossEntropy()
    out = {}
    for cache in (False, True):
        for ola in (False, True):
            m.configure_frontend(bits, cache, ola)
            g = m.loss_grad(x, y, spec)[3]
            xa = m.pgd_run(x, y, lower, upper, spec, 0.0004, 3, 1)[0]  # odd number of steps: the iterate ends on the twin buffer
            xb = m.pgd_run(x, y, lower, upper, spec, 0.0004, 2, 1)[0]
            out[(cache, ola)] = (g, xa, xb)
    ref = out[(False, False)]
    assert ref[0].abs().max().item() > 0
    for key, val in out.items():
        if bits == 64 and key[0]:
            continue  # the cached float32 spectrum rounds the float64 one: compared among themselves below
        for a, b in zip(ref, val):
            assert torch.equal(a, b), "fft %d, spectrum cache %s, fused overlap-add %s differs" % ((bits,) + key)
    if bits == 64:
        for a, b in zip(out[(True, False)], out[(True, True)]):
            assert torch.equal(a, b)
        d = (out[(True, True)][0] - ref[0]).abs().max().item() / ref[0].abs().max().item()
        log("audionet adjoint, float64 transforms, B=%d T=%d: cached float32 spectrum vs re-transform, d loss/d wav differs by %.2e of max" % (B, T, d))
        assert d < 1e-5


def test_float32_transforms_against_float64(sd, dev):
    """The two transform precisions of sg_an_configure side by side on the full-size input: log-mel in dB, logits,
    decisions, d loss / d wav (recorded in the parity log; the reference's own STFT is float32)."""
    from speakerguard_amd import synth
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    from speakerguard_amd.model.audionet_csine import audionet_csine
    m = audionet_csine.from_weights(sd, device=dev)
    x = torch.from_numpy(synth.make_waveforms(16, 48000, seed=77)).to(dev)
    spec = SEC4SR_CrossEntropy()
    res = {}
    for bits in (64, 32):
        m.configure_frontend(bits, False, True)
        feats = m.compute_feat(x)
        y = m.make_decision(x)[0] if bits == 64 else res[64][1]
        dec, scores, loss, g = m.loss_grad(x, y, spec)
        res[bits] = (feats, y, dec, scores, g)
    f64, _, d64, s64, g64 = res[64]
    f32, _, d32, s32, g32 = res[32]
    gerr = (g32 - g64).abs().max().item() / g64.abs().max().item()
    sign = float((torch.sign(g32) != torch.sign(g64)).float().mean())
    log("audionet float32 vs float64 transforms (16 x 3 s): log-mel max %.2e dB, logits max %.2e, decisions equal %s, "
        "d loss/d wav %.2e of max, sign mismatch %.2e" % ((f32 - f64).abs().max().item(), (s32 - s64).abs().max().item(),
                                                          bool(torch.equal(d32, d64)), gerr, sign))
    assert (f32 - f64).abs().max().item() < 2e-3 and torch.equal(d32, d64) and gerr < 1e-3


def test_environment_knobs_count_only_behind_sg_tune(tmp_path):
    """Round 6 (hygiene): the library reads no SG_* variable unless SG_TUNE=1 is set -- a stray knob in a user's
    environment changes nothing.  A child process with SG_AN_FUSED=0 in its environment still runs the fused CNN launches
    without the gate, and the per-layer sequence with it ("Environment")."""
    import subprocess
    import sys
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from speakerguard_amd import synth\n"
        "from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy\n"
        "from speakerguard_amd.model.audionet_csine import audionet_csine\n"
        "dev = torch.device('cuda:0')\n"
        "m = audionet_csine.from_weights(synth.make_audionet_state_dict(seed=0, num_class=251), device=dev)\n"
        "x = torch.from_numpy(synth.make_waveforms(2, 32000, seed=3)).to(dev)\n"
        "y = torch.zeros(2, dtype=torch.int64, device=dev)\n"
        "tags = [t for t, _ in m.trace_stages(lambda: m.loss_grad(x, y, SEC4SR_CrossEntropy()), max_records=256)]\n"
        "print('FUSED' if 'an_cnn_fwd' in tags else 'PERLAYER')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    base = {k: v for k, v in os.environ.items() if not k.startswith("SG_")}
    for env, want in ((dict(base, SG_AN_FUSED="0"), "FUSED"), (dict(base, SG_AN_FUSED="0", SG_TUNE="1"), "PERLAYER")):
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.strip().splitlines()[-1] == want, (want, r.stdout, r.stderr[-500:])
