"""The BASELINE.json configurations at (or near) their full sizes, HIP path vs the oracle, with parameters chosen so that BOTH
outcomes of the success predicate occur (round-2 review: several flag comparisons were all-False against all-False), plus
the two drop-in gaps of the x-vector system: the file-based constructor on the GPU and PGD's random restarts on the engine.

The oracle (reference-pinned CPU restatement) is the checker; these tests need tens of seconds of it
each on the GPU box's host cores.  Which parameters leave some utterances un-fooled was found with the HIP path alone
(tests/tools/probe_outcomes.py).
"""
import os

import numpy as np
import pytest
import torch

from conftest import log

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def hip_model(xv_weights):
    from speakerguard_amd.model.xv_plda import xv_plda
    return xv_plda.from_weights(xv_weights, device=DEV, dither=0.0)


@pytest.fixture(scope="module")
def oracle_model(xv_weights):
    from oracle.xv_plda import XvPlda
    return XvPlda(xv_weights, faithful=False, freeze=True)


def _compare(tag, x, adv, succ, oadv, osucc, hip_dec, ora_dec, eps, steps, want_mixed=True):
    """north_star's parity statement: success flags and predicted ids bit-exact, perturbation within the stated tolerance
    (<= 2 % of the samples differ per sign step, by at most 2 eps: DESIGN.md section 2)."""
    B = x.shape[0]
    succ, osucc = [bool(s) for s in succ], [bool(s) for s in osucc]
    diff = (adv.cpu() - oadv).abs()
    frac = float((diff > 1e-7).float().mean())
    with torch.no_grad():
        hd, od = hip_dec(adv), ora_dec(oadv)
        od_on_h, hd_on_o = ora_dec(adv.cpu()), hip_dec(oadv.to(DEV))
    log("%s: success HIP %d/%d oracle %d/%d (equal per utterance: %s); ids on own audio equal %d/%d; samples differing %.2f %%, max |diff| %.6f"
        % (tag, sum(succ), B, sum(osucc), B, succ == osucc, int((hd.cpu() == od).sum()), B, 100 * frac, diff.max().item()))
    assert succ == osucc, (tag, succ, osucc)
    if want_mixed:
        assert 0 < sum(succ) < B, "%s: the parameters were chosen so that both outcomes occur (%d/%d)" % (tag, sum(succ), B)
    assert hd.cpu().tolist() == od.tolist()                      # predicted ids on the adversarial audio
    assert hd.cpu().tolist() == od_on_h.tolist() and hd_on_o.cpu().tolist() == od.tolist()  # both models agree on either audio
    assert (adv.cpu() - x).abs().max().item() <= eps + 1e-7
    assert diff.max().item() <= 2 * eps + 1e-6 and frac <= 0.02 * steps


def test_config1_full_size_both_outcomes(hip_model, oracle_model):
    """configs[1] at full size -- 64 utterances x 3 s, PGD-5, cross-entropy, untargeted, CSI-E -- with the model's own clean
    decisions as labels and a ball small enough (eps 0.0005) that ~47 of 64 utterances are fooled and the rest are not."""
    from oracle import attacks as oatk
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    eps, step, K, B = 0.0005, 0.0001, 5, 64
    x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=1234))
    with torch.no_grad():
        y = oracle_model.make_decision(x)[0]
    assert hip_model.make_decision(x.to(DEV))[0].cpu().tolist() == y.tolist()
    adv, succ = PGD(hip_model, task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B, verbose=0).attack(x.to(DEV), y.to(DEV))
    oadv, osucc = oatk.PGD(oracle_model, task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B).attack(x, y)
    _compare("configs[1] PGD-5 x 64 x 3 s, eps 0.0005", x, adv, succ, oadv, osucc,
             lambda a: hip_model.make_decision(a)[0], lambda a: oracle_model.make_decision(a)[0], eps, K)


def test_config1_bench_workload_targeted(hip_model, oracle_model):
    """The bench's inputs (labels arange % 10, eps 0.002, step 0.0004) as a TARGETED PGD-5 towards label + 3: ~59 of 64
    reach the target, the others do not."""
    from oracle import attacks as oatk
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    eps, step, K, B = 0.002, 0.0004, 5, 64
    x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=1234))
    y = (torch.arange(B) % 10 + 3) % 10
    adv, succ = PGD(hip_model, task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B, targeted=True, verbose=0).attack(
        x.to(DEV), y.to(DEV))
    oadv, osucc = oatk.PGD(oracle_model, task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B, targeted=True).attack(x, y)
    _compare("configs[1] targeted PGD-5 x 64 x 3 s", x, adv, succ, oadv, osucc,
             lambda a: hip_model.make_decision(a)[0], lambda a: oracle_model.make_decision(a)[0], eps, K)


def test_config3_one_gpus_shard(capsys):
    """configs[3] on one GPU's shard: PGD-10 against the FeCo-defended AudioNet (FeCo at the log-mel level, cl_r 0.5,
    deterministic clustering), 64 utterances x 3 s, through the ONE device loop (sg_an_pgd_run_feco) vs the oracle --
    reference-pinned AudioNet restatement + oracle.feco on ITS OWN features + torch autograd.  25 of 64 utterances are
    fooled."""
    from oracle import attacks as oatk
    from oracle import feco
    from oracle.audionet import AudioNet
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    from speakerguard_amd.defense.feature_level import FeCoDefense
    from speakerguard_amd.model.audionet_csine import audionet_csine
    from speakerguard_amd.model.defended_model import defended_model
    eps, step, K, B, ratio = 0.002, 0.0004, 10, 64, 0.5
    sd = synth.make_audionet_state_dict(seed=0, num_class=251)
    hip, ora = audionet_csine.from_weights(sd, device=DEV), AudioNet(sd)

    class OracleDefended:
        threshold = -np.inf

        def make_decision(self, xx):
            feats = ora.compute_feat(xx, flag=1)
            k = int(feats.shape[1] * ratio)
            comp = [feco.compress_from_ids(feats[b], feco.kmeans_ids(feats[b].detach().numpy(), k), k, force=True)
                    for b in range(feats.shape[0])]
            return ora.make_decision(torch.stack(comp), flag=1)

    x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=3))
    dm, om = defended_model(hip, defense=[(1, FeCoDefense(ratio))]), OracleDefended()
    y = dm.make_decision(x.to(DEV))[0].cpu()
    with torch.no_grad():
        assert om.make_decision(x)[0].tolist() == y.tolist()
    atk = PGD(dm, task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B, verbose=0)
    assert atk._fused_feco(B) is not None  # the device loop, not the host-chained one
    adv, succ = atk.attack(x.to(DEV), y.to(DEV))
    oadv, osucc = oatk.PGD(om, task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B).attack(x, y)

    def odec(a):
        with torch.no_grad():
            return om.make_decision(a)[0]
    _compare("configs[3] PGD-10 vs FeCo-defended AudioNet x 64 x 3 s", x, adv, succ, oadv, osucc, lambda a: dm.make_decision(a)[0], odec, eps, K)


def test_audionet_fgsm_and_pgd_both_outcomes():
    """configs[0] (FGSM on AudioNet CSI-NE) on 16 utterances x 3 s and PGD-5 on the same: eps 0.002 fools 5 / 7 of 16."""
    from oracle import attacks as oatk
    from oracle.audionet import AudioNet
    from speakerguard_amd import synth
    from speakerguard_amd.attack.FGSM import FGSM
    from speakerguard_amd.attack.PGD import PGD
    from speakerguard_amd.model.audionet_csine import audionet_csine
    sd = synth.make_audionet_state_dict(seed=0, num_class=251)
    hip, ora = audionet_csine.from_weights(sd, device=DEV), AudioNet(sd)
    x = torch.from_numpy(synth.make_waveforms(16, 48000, seed=3))
    with torch.no_grad():
        y = ora.make_decision(x)[0]
    assert hip.make_decision(x.to(DEV))[0].cpu().tolist() == y.tolist()
    eps = 0.002

    def odec(a):
        with torch.no_grad():
            return ora.make_decision(a)[0]
    adv, succ = FGSM(hip, task="CSI", epsilon=eps, batch_size=16, verbose=0).attack(x.to(DEV), y.to(DEV))
    oadv, osucc = oatk.FGSM(ora, task="CSI", epsilon=eps, batch_size=16).attack(x.clone(), y)
    _compare("configs[0] FGSM on AudioNet x 16 x 3 s", x, adv, succ, oadv, osucc, lambda a: hip.make_decision(a)[0], odec, eps, 1)
    adv, succ = PGD(hip, task="CSI", epsilon=eps, step_size=eps / 5, max_iter=5, batch_size=16, verbose=0).attack(x.to(DEV), y.to(DEV))
    oadv, osucc = oatk.PGD(ora, task="CSI", epsilon=eps, step_size=eps / 5, max_iter=5, batch_size=16).attack(x.clone(), y)
    _compare("PGD-5 on AudioNet x 16 x 3 s", x, adv, succ, oadv, osucc, lambda a: hip.make_decision(a)[0], odec, eps, 5)


def test_config4_fakebob_osi_targeted_s50(xv_weights):
    """configs[4] as written: FAKEBOB / NES on the x-vector system in the OSI task, 50 samples per draw (+ the clean query),
    targeted at each voice's nearest enrolled speaker, threshold -5: two voices are accepted as their target from the start
    (the early-stop / delete_found branch, attack/FAKEBOB.py:85-90,125-168), one rejected voice (top score -8.3) is pushed over
    the threshold, one (-54.7) is not within 4 iterations.  Both sides draw the NES noise from the same seeded host stream."""
    from oracle import attacks as oatk
    from oracle.xv_plda import XvPlda
    from speakerguard_amd import synth
    from speakerguard_amd.attack.FAKEBOB import FAKEBOB
    from speakerguard_amd.model.xv_plda import xv_plda
    th, eps = -5.0, 0.002
    hm = xv_plda.from_weights(xv_weights, threshold=th, device=DEV, dither=0.0)
    om = XvPlda(xv_weights, threshold=th, faithful=False, freeze=True)
    x = torch.from_numpy(synth.make_waveforms(64, 48000, seed=1234))[[0, 1, 4, 5]]
    with torch.no_grad():
        d0, s0 = om.make_decision(x)
    hd0, hs0 = hm.make_decision(x.to(DEV))
    assert hd0.cpu().tolist() == d0.tolist() and sorted(d0.tolist())[0] == -1  # some voices start rejected
    yt = s0.argmax(1)
    kw = dict(task="OSI", targeted=True, threshold=th, epsilon=eps, max_iter=4, max_lr=0.001, min_lr=1e-6, samples_per_draw=50,
              samples_per_draw_batch_size=50, sigma=0.001, stop_early=True, stop_early_iter=100, batch_size=4)
    g = torch.Generator().manual_seed(5)
    oadv, osucc = oatk.FAKEBOB(om, noise_fn=lambda shape: torch.randn(shape, generator=g), **kw).attack(x.clone(), yt)
    g2 = torch.Generator().manual_seed(5)
    adv, succ = FAKEBOB(hm, verbose=0, noise_fn=lambda shape: torch.randn(shape, generator=g2), **kw).attack(x.to(DEV), yt.to(DEV))

    def odec(a):
        with torch.no_grad():
            return om.make_decision(a)[0]
    _compare("configs[4] FAKEBOB OSI targeted S=50 x 4 x 3 s", x, adv, succ, oadv, osucc, lambda a: hm.make_decision(a)[0], odec, eps, 4)
    # a voice that started rejected ended accepted as its target
    hd1 = hm.make_decision(adv)[0].cpu()
    assert any(int(a) == -1 and int(b) == int(t) for a, b, t in zip(d0, hd1, yt))


def test_file_based_constructor_on_the_gpu(xv_weights, tmp_path):
    """The drop-in entry point: xv_plda(extractor_file, plda_file, mean_file, transform_mat_file, model_file, ...) exactly as
    reference model/xv_plda.py:17-47 is called, on files written in the reference's formats (state_dict checkpoint, Kaldi-text
    PLDA, mean.vec, transform.txt, speaker_model list + per-speaker embedding files) -- equals from_weights bit for bit on the device."""
    from speakerguard_amd import synth
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    from speakerguard_amd.model.xv_plda import xv_plda
    paths = synth.write_xv_model_dir(str(tmp_path), xv_weights)
    from_files = xv_plda(paths["extractor_file"], paths["plda_file"], paths["mean_file"], paths["transform_mat_file"],
                         model_file=paths.get("model_file"), device=DEV, dither=0.0)
    direct = xv_plda.from_weights(xv_weights, device=DEV, dither=0.0)
    assert from_files.num_spks == direct.num_spks and list(from_files.allowed_flags) == [0, 1, 2]
    x = torch.from_numpy(synth.make_waveforms(3, 32000, seed=91)).to(DEV)
    d1, s1 = from_files.make_decision(x)
    d2, s2 = direct.make_decision(x)
    assert torch.equal(d1, d2) and torch.equal(s1, s2)
    y = torch.tensor([0, 1, 2], device=DEV)
    g1 = from_files.loss_grad(x, y, SEC4SR_CrossEntropy())
    g2 = direct.loss_grad(x, y, SEC4SR_CrossEntropy())
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))
    assert torch.equal(from_files.embedding(x), direct.embedding(x))


def test_pgd_random_restarts_on_the_engine(hip_model, oracle_model):
    """attack/PGD.py:58-61,74-77 on the HIP engine: three random restarts drawn from numpy's global generator (seeded alike on
    both sides), the restart with the best WHOLE-BATCH success rate is returned.  eps 0.0002 leaves some utterances
    un-fooled, so the restarts differ in their success counts."""
    from oracle import attacks as oatk
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    eps, step, K, B = 0.0002, 0.00004, 4, 8
    x = torch.from_numpy(synth.make_waveforms(B, 32000, seed=1234))
    with torch.no_grad():
        y = oracle_model.make_decision(x)[0]
    np.random.seed(77)
    adv, succ = PGD(hip_model, task="CSI", epsilon=eps, step_size=step, max_iter=K, num_random_init=3, batch_size=B, verbose=0).attack(
        x.to(DEV), y.to(DEV))
    np.random.seed(77)
    oadv, osucc = oatk.PGD(oracle_model, task="CSI", epsilon=eps, step_size=step, max_iter=K, num_random_init=3, batch_size=B).attack(x.clone(), y)
    # every restart, replayed one by one on the engine: the returned one is the first with the best batch success rate
    np.random.seed(77)
    rates = []
    for _ in range(3):
        noise = torch.tensor(np.random.uniform(-eps, eps, tuple(x.shape)), dtype=x.dtype)  # PGD.py:60
        xi = (x + noise).to(DEV)
        one = PGD(hip_model, task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B, verbose=0)
        lower, upper = torch.clamp(x - eps, min=-1).to(DEV), torch.clamp(x + eps, max=1).to(DEV)
        one._begin_attack()
        a_i, s_i = one._run_batches(xi, y.to(DEV), lower, upper, tag=0)
        rates.append((sum(s_i), a_i))
    best = max(range(3), key=lambda i: (rates[i][0], -i))
    assert sum(succ) == rates[best][0] and torch.equal(adv, rates[best][1])
    _compare("PGD-4 with 3 random restarts x 8 x 2 s", x, adv, succ, oadv, osucc,
             lambda a: hip_model.make_decision(a)[0], lambda a: oracle_model.make_decision(a)[0], eps, K, want_mixed=False)
    log("  success counts of the three restarts on the engine: %s (returned: restart %d)" % ([r[0] for r in rates], best))


# ---- round 4: the metric's EXACT shapes inside the driver-run suite ("Next round" item 6) -----------------------
def test_config1_exact_metric_pgd20(hip_model, oracle_model):
    """BASELINE.json's metric as written and as bench.py times it: PGD-**20**, eps 0.002, step 0.0004, cross-entropy,
    untargeted, labels arange % 10, 64 utterances x 3 s, dither off -- HIP device loop vs the oracle (rounds 2-3 ran this
    shape as a one-off outside the suite; the suite had PGD-5).  Every utterance is fooled at these parameters on both
    sides, so the mixed-outcome assertion is off here (the two PGD-5 tests above carry it)."""
    from oracle import attacks as oatk
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    eps, step, K, B = 0.002, 0.0004, 20, 64
    x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=1234))
    y = torch.arange(B) % 10
    adv, succ = PGD(hip_model, task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B, verbose=0).attack(x.to(DEV), y.to(DEV))
    oadv, osucc = oatk.PGD(oracle_model, task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B).attack(x, y)
    _compare("configs[1] EXACT: PGD-20 x 64 x 3 s, eps 0.002", x, adv, succ, oadv, osucc,
             lambda a: hip_model.make_decision(a)[0], lambda a: oracle_model.make_decision(a)[0], eps, K, want_mixed=False)
    # the same audio scored by both models: the scores themselves agree (the trajectories end at different points of the
    # eps-ball -- sign() feeds round-off back -- so scores on OWN audio are not comparable)
    with torch.no_grad():
        osc = oracle_model.make_decision(adv.cpu())[1]
    hsc = hip_model.make_decision(adv)[1].cpu()
    assert (hsc - osc).abs().max().item() < 2e-2 * max(1.0, osc.abs().max().item() / 100.0)


def test_config1_full_size_with_the_reference_default_dither(xv_weights):
    """The reference's front-end as it ships: dither = 1.0 (xv_plda.py:119).  PGD-5 x 64 x 3 s through the device loop
    with the engine's own in-kernel noise (Philox keyed by global utterance, frame, sample and pass) against the oracle
    fed THE SAME noise as explicit tensors (oracle/philox.py restates the streams; the generator is pinned by the
    Random123 vectors): flags and ids equal, perturbation within the stated tolerance -- the full-size case with
    dither != 0 the round-3 review found missing."""
    from oracle import attacks as oatk
    from oracle import kaldi_mfcc, philox
    from oracle.xv_plda import XvPlda
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    from speakerguard_amd.model.xv_plda import xv_plda
    eps, step, K, B, T = 0.0005, 0.0001, 5, 64, 48000
    hip = xv_plda.from_weights(xv_weights, device=DEV, dither=1.0, dither_seed=21)
    ora = XvPlda(xv_weights, faithful=False, freeze=True)
    F = kaldi_mfcc.num_frames(T)
    x = torch.from_numpy(synth.make_waveforms(B, T, seed=1234))
    y = hip_clean = None
    quiet = xv_plda.from_weights(xv_weights, device=DEV, dither=0.0)
    y = quiet.make_decision(x.to(DEV))[0].cpu()  # labels: the clean decisions of the noise-free model

    atk = PGD(hip, task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B, verbose=0)
    adv, succ = atk.attack(x.to(DEV), y.to(DEV))
    base = hip.last_fused_seed  # generator key of the fused call; pass `it` uses fused_pass_seed(base, it)

    class Dithered:
        """the oracle with the engine's noise of pass number `calls` handed in as a tensor"""
        threshold = ora.threshold
        calls = 0

        def make_decision(self, xx):
            key = xv_plda.fused_pass_seed(base, self.calls)
            type(self).calls += 1
            noise = torch.from_numpy(np.stack([philox.dither_noise(key, b, F) for b in range(xx.shape[0])]))
            return ora.make_decision(xx, dither_noise=noise)

    oadv, osucc = oatk.PGD(Dithered(), task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B).attack(x, y)
    assert Dithered.calls == K + 1
    # decisions on the adversarial audio are taken with the noise-free models on both sides (a fresh draw would differ)
    _compare("configs[1] with dither 1.0 (reference default): PGD-5 x 64 x 3 s", x, adv, succ, oadv, osucc,
             lambda a: quiet.make_decision(a)[0], lambda a: ora.make_decision(a)[0], eps, K)
    # ... and the dither matters at this size: the noise-free trajectory is a different one
    adv0, _ = PGD(quiet, task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B, verbose=0).attack(x.to(DEV), y.to(DEV))
    moved = float(((adv0 - adv).abs() > 1e-7).float().mean())
    log("  dither 1.0 vs dither 0 on the device: %.2f %% of the samples end elsewhere" % (100 * moved))
    assert moved > 0.001


def test_config2_cw2_sv_batch32_at_three_seconds(xv_weights):
    """configs[2] at its own size: CW2 (L2, Adam) targeted on the SV task, 32 utterances x 3 s, one search step x 10
    iterations, against the oracle loop (round 3 met the oracle at 1 s and checked 3 s by properties only)."""
    from oracle import attacks as oatk
    from oracle.xv_plda import XvPlda
    from speakerguard_amd import synth
    from speakerguard_amd.attack.CW2 import CW2
    from speakerguard_amd.model.xv_plda import xv_plda
    w = dict(xv_weights)
    w["enroll"] = xv_weights["enroll"][:1].copy()  # SV: one enrolled speaker
    x = torch.from_numpy(synth.make_waveforms(32, 48000, seed=35))
    probe = xv_plda.from_weights(w, device=DEV, dither=0.0)
    clean = probe.make_decision(x.to(DEV))[1][:, 0].cpu()
    assert float(clean.max()) < 0.0
    thr = 55.0  # every voice starts rejected (clean scores -170 .. -2); ten Adam steps of 2e-4 carry 26 of the 32 over it
    om, hm = XvPlda(w, threshold=thr, faithful=False, freeze=True), xv_plda.from_weights(w, threshold=thr, device=DEV, dither=0.0)
    y = torch.zeros(32, dtype=torch.long)
    lr = 2e-4
    kw = dict(task="SV", targeted=True, confidence=0.0, initial_const=1e-2, binary_search_steps=1, max_iter=10,
              stop_early=True, stop_early_iter=5, lr=lr, batch_size=32)
    oadv, osucc = oatk.CW2(om, **kw).attack(x.clone(), y)
    adv, succ = CW2(hm, verbose=0, **kw).attack(x.to(DEV), y.to(DEV))
    d = (adv.cpu() - oadv).abs().numpy()
    l2h, l2o = (adv.cpu() - x).flatten(1).norm(dim=1), (oadv - x).flatten(1).norm(dim=1)
    log("configs[2] CW2 targeted SV x 32 x 3 s, 1 search step x 10 iterations: success HIP %d/32 oracle %d/32 (equal per utterance: %s); "
        "max |x_adv - oracle| %.3e, differing by more than lr / 10: %.2f %%, by more than 2 lr: %.4f %%; L2 of the perturbation %.4f vs %.4f"
        % (sum(succ), sum(osucc), [bool(a) for a in succ] == [bool(a) for a in osucc], d.max(), 100 * (d > lr / 10).mean(), 100 * (d > 2 * lr).mean(),
           float(l2h.mean()), float(l2o.mean())))
    assert [bool(a) for a in succ] == [bool(a) for a in osucc]
    assert 0 < sum(succ) < 32, "both outcomes: %d/32" % sum(succ)
    assert hm.make_decision(adv)[0].cpu().tolist() == om.make_decision(oadv)[0].tolist()
    # Adam's update is ~ lr * sign(g) while its second moment is young: a round-off-level gradient entry moves by +-lr on either
    # side per iteration (2.5 % of the samples end more than lr / 10 apart); stated tolerance: <= 0.5 % of the samples further
    # apart than two full steps, none further than 2 lr per iteration
    assert (d > 2 * lr).mean() < 5e-3 and d.max() <= 2 * lr * 10 + 1e-6
    assert abs(float(l2h.mean()) - float(l2o.mean())) < 0.02 * float(l2o.mean())


def test_config4_fakebob_twenty_iterations_with_early_stop(xv_weights):
    """configs[4] for 20 iterations (round 3 ran 4 of the ~196 a 10k-query budget allows): FAKEBOB OSI targeted, 50 + 1
    queries per voice and iteration, two voices, shared seeded NES noise -- one voice crosses the threshold during the run and
    leaves the batch through the early-stop branch (attack/FAKEBOB.py:85-90,125-168), the other keeps being queried."""
    from oracle import attacks as oatk
    from oracle.xv_plda import XvPlda
    from speakerguard_amd import synth
    from speakerguard_amd.attack.FAKEBOB import FAKEBOB
    from speakerguard_amd.model.xv_plda import xv_plda
    eps, iters = 0.002, 20
    x = torch.from_numpy(synth.make_waveforms(64, 48000, seed=1234))[[6, 7]]
    probe = xv_plda.from_weights(xv_weights, device=DEV, dither=0.0)
    s0 = probe.make_decision(x.to(DEV))[1].cpu()
    yt = s0.argmax(1)
    top = s0.max(1).values
    th = float(top.max()) + 4.0  # both start rejected; the nearer one is 4 below the threshold (4 joint iterations), the other 158
    hm = xv_plda.from_weights(xv_weights, threshold=th, device=DEV, dither=0.0)
    om = XvPlda(xv_weights, threshold=th, faithful=False, freeze=True)
    assert hm.make_decision(x.to(DEV))[0].cpu().tolist() == [-1, -1]
    kw = dict(task="OSI", targeted=True, threshold=th, epsilon=eps, max_iter=iters, max_lr=0.001, min_lr=1e-6, samples_per_draw=50,
              samples_per_draw_batch_size=50, sigma=0.001, stop_early=True, stop_early_iter=100, batch_size=2)
    g = torch.Generator().manual_seed(9)
    draws_o = []
    oadv, osucc = oatk.FAKEBOB(om, noise_fn=lambda shape: (draws_o.append(shape[0]), torch.randn(shape, generator=g))[1], **kw).attack(x.clone(), yt)
    g2 = torch.Generator().manual_seed(9)
    draws_h = []
    adv, succ = FAKEBOB(hm, verbose=0, noise_fn=lambda shape: (draws_h.append(shape[0]), torch.randn(shape, generator=g2))[1], **kw).attack(
        x.to(DEV), yt.to(DEV))

    def odec(a):
        with torch.no_grad():
            return om.make_decision(a)[0]
    log("configs[4] FAKEBOB OSI targeted S=50 x 2 x 3 s, %d iterations: voices queried per iteration HIP %s oracle %s" % (iters, draws_h, draws_o))
    assert draws_h == draws_o, "the batch shrinks at the same iteration on both sides"
    assert len(draws_h) >= 8 and draws_h[0] == 2 and draws_h[-1] == 1, "an early-stop hit during the run, the other voice goes on: %s" % draws_h
    _compare("configs[4] FAKEBOB OSI targeted S=50 x 2 x 3 s, 20 iterations", x, adv, succ, oadv, osucc,
             lambda a: hm.make_decision(a)[0], odec, eps, iters)


def test_attack_survives_a_lost_streamk_handoff(xv_weights):
    """include/speakerguard_hip.h, sg_set_streamk: stream-K needs the GPU to itself; when a hand-off wait times out (here:
    provoked through the fault-injection hook in the middle of a PGD attack) the attack driver re-runs its batches once as
    one block per tile -- the same fused multiply-add chains -- and the caller gets the result of an undisturbed run."""
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    from speakerguard_amd.model.xv_plda import xv_plda
    m = xv_plda.from_weights(xv_weights, device=DEV, dither=0.0)
    x = torch.from_numpy(synth.make_waveforms(64, 48000, seed=321)).to(DEV)
    y = m.make_decision(x)[0]
    kw = dict(task="CSI", epsilon=0.0005, step_size=0.0001, max_iter=3, batch_size=64, verbose=0)
    clean_adv, clean_succ = PGD(m, **kw).attack(x, y)
    m.ctx.call("sg_debug_lose_handoffs", 2)  # two contraction launches of the coming attack lose their hand-off flags
    adv, succ = PGD(m, **kw).attack(x, y)
    assert getattr(m, "streamk", True) is False, "the driver should have switched stream-K off for the retry"
    assert torch.equal(adv, clean_adv) and succ == clean_succ
    m.ctx.call("sg_health")  # nothing pending
    m.set_streamk(True)
    adv2, succ2 = PGD(m, **kw).attack(x, y)
    assert torch.equal(adv2, clean_adv) and succ2 == clean_succ
    log("PGD-3 x 64 x 3 s with two lost stream-K hand-offs: retried as tile launches, result equal to the undisturbed run (%d/64 fooled)" % sum(succ))


def test_config3_timed_workload_random_start_eot2():
    """the configs[3] workload as bench.py times it -- the RANDOMISED defense (every clustering starts from fresh
    random frames) attacked with EOT 2, 64 utterances x 3 s through the one device loop (sg_an_pgd_run_feco) -- against the
    oracle loop (reference-pinned AudioNet restatement + oracle.feco + torch autograd, adaptive_attack/EOT.py:16-54) whose
    clusterings start from the Philox RESTATEMENT of the engine's draws (oracle/philox.py: pass (step it, repeat r) uses key
    seed + it * 0x9E37.. + r * 0xC2B2.., utterance = global row), the way test_config1_..._default_dither feeds the dither.
    PGD-5 (the suite's time budget; the bench runs 20 steps of the same loop)."""
    from oracle import attacks as oatk
    from oracle import feco, philox
    from oracle.audionet import AudioNet
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    from speakerguard_amd.defense.feature_level import FeCoDefense
    from speakerguard_amd.model.audionet_csine import audionet_csine
    from speakerguard_amd.model.defended_model import defended_model
    eps, step, K, B, ratio, R = 0.002, 0.0004, 5, 64, 0.5, 2
    sd = synth.make_audionet_state_dict(seed=0, num_class=251)
    hip, ora = audionet_csine.from_weights(sd, device=DEV), AudioNet(sd)

    class OracleDefended:
        threshold = -np.inf
        base_seed, it = 0, 0

        def make_decision(self, xx):
            feats = ora.compute_feat(xx, flag=1)
            F = feats.shape[1]
            k = int(F * ratio)
            comp = []
            for row in range(feats.shape[0]):
                r, u = divmod(row, B)
                init = philox.feco_random_init(hip.fused_pass_seed(self.base_seed, self.it, r), u, F, k)
                ids = feco.kmeans_ids(feats[row].detach().numpy(), k, init_frames=init)
                comp.append(feco.compress_from_ids(feats[row], ids, k, force=True))
            self.it += 1
            return ora.make_decision(torch.stack(comp), flag=1)

    x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=3))
    y = hip.make_decision(x.to(DEV))[0].cpu()  # the undefended model's clean decisions
    dm, om = defended_model(hip, defense=[(1, FeCoDefense(ratio, init='random', seed=11))]), OracleDefended()
    atk = PGD(dm, task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B, EOT_size=R, EOT_batch_size=R, verbose=0)
    assert atk._fused_feco(B) is not None  # the device loop, not the host-chained one
    adv, succ = atk.attack(x.to(DEV), y.to(DEV))
    om.base_seed, om.it = hip.last_fused_seed, 0
    oadv, osucc = oatk.PGD(om, task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B, EOT_size=R, EOT_batch_size=R).attack(x, y)
    succ, osucc = [bool(s) for s in succ], [bool(s) for s in osucc]
    diff = (adv.cpu() - oadv).abs()
    frac = float((diff > 1e-7).float().mean())
    log("configs[3] as timed (random start, EOT %d, PGD-%d x %d x 3 s): success HIP %d/%d oracle %d/%d (equal per utterance: %s); samples "
        "differing %.2f %%, max |diff| %.6f" % (R, K, B, sum(succ), B, sum(osucc), B, succ == osucc, 100 * frac, diff.max().item()))
    assert succ == osucc
    assert 0 < sum(succ) < B, "both outcomes: %d/%d" % (sum(succ), B)
    assert (adv.cpu() - x).abs().max().item() <= eps + 1e-7
    assert diff.max().item() <= 2 * eps + 1e-6 and frac <= 0.02 * K


def test_config3_pgd100_on_eight_utterances():
    """BASELINE configs[3] says PGD-100: one hundred sign steps against the FeCo-defended AudioNet (deterministic start) on 8
    utterances x 3 s, device loop vs oracle.  sign() turns round-off on near-zero gradient entries into +-step flips that
    feed back, so two float32 implementations drift apart sample by sample; what the hundred steps do to
    the outcome is bounded here: flags and ids equal, every sample within 2 eps, the fraction of differing samples logged."""
    from oracle import attacks as oatk
    from oracle import feco
    from oracle.audionet import AudioNet
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    from speakerguard_amd.defense.feature_level import FeCoDefense
    from speakerguard_amd.model.audionet_csine import audionet_csine
    from speakerguard_amd.model.defended_model import defended_model
    eps, step, K, B, ratio = 0.0004, 0.00001, 100, 8, 0.5
    sd = synth.make_audionet_state_dict(seed=0, num_class=251)
    hip, ora = audionet_csine.from_weights(sd, device=DEV), AudioNet(sd)

    class OracleDefended:
        threshold = -np.inf

        def make_decision(self, xx):
            feats = ora.compute_feat(xx, flag=1)
            k = int(feats.shape[1] * ratio)
            comp = [feco.compress_from_ids(feats[b], feco.kmeans_ids(feats[b].detach().numpy(), k), k, force=True)
                    for b in range(feats.shape[0])]
            return ora.make_decision(torch.stack(comp), flag=1)

    x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=3))
    dm, om = defended_model(hip, defense=[(1, FeCoDefense(ratio))]), OracleDefended()
    y = dm.make_decision(x.to(DEV))[0].cpu()
    adv, succ = PGD(dm, task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B, verbose=0).attack(x.to(DEV), y.to(DEV))
    oadv, osucc = oatk.PGD(om, task="CSI", epsilon=eps, step_size=step, max_iter=K, batch_size=B).attack(x, y)
    succ, osucc = [bool(s) for s in succ], [bool(s) for s in osucc]
    diff = (adv.cpu() - oadv).abs()
    frac = float((diff > 1e-7).float().mean())
    with torch.no_grad():
        hd, od = dm.make_decision(adv)[0].cpu(), om.make_decision(oadv)[0]
    log("configs[3] PGD-100 vs FeCo-defended AudioNet x %d x 3 s (eps %g, step %g): success HIP %d/%d oracle %d/%d (equal per utterance: %s); "
        "ids on own audio equal %d/%d; samples differing %.2f %%, max |diff| %.6f (2 eps = %.6f)"
        % (B, eps, step, sum(succ), B, sum(osucc), B, succ == osucc, int((hd == od).sum()), B, 100 * frac, diff.max().item(), 2 * eps))
    assert succ == osucc and hd.tolist() == od.tolist()
    assert (adv.cpu() - x).abs().max().item() <= eps + 1e-7 and diff.max().item() <= 2 * eps + 1e-6


def test_config2_cw2_three_search_steps_both_const_branches(xv_weights):
    """configs[2] with the outer loop of attack/CW2.py:113-123 at full size: 3 binary-search steps x 8 iterations on 32
    utterances x 3 s.  After the first step some voices have crossed the threshold (their `const` is bisected towards the lower
    bound) and some have not (theirs is multiplied by 10): both branches of the update run, on both sides, and the outcomes,
    ids and perturbation sizes agree with the oracle loop."""
    from oracle import attacks as oatk
    from oracle.xv_plda import XvPlda
    from speakerguard_amd import synth
    from speakerguard_amd.attack.CW2 import CW2
    from speakerguard_amd.model.xv_plda import xv_plda
    w = dict(xv_weights)
    w["enroll"] = xv_weights["enroll"][:1].copy()  # SV: one enrolled speaker
    x = torch.from_numpy(synth.make_waveforms(32, 48000, seed=35))
    thr = 55.0
    om, hm = XvPlda(w, threshold=thr, faithful=False, freeze=True), xv_plda.from_weights(w, threshold=thr, device=DEV, dither=0.0)
    y = torch.zeros(32, dtype=torch.long)
    lr, steps, iters = 2e-4, 3, 8
    kw = dict(task="SV", targeted=True, confidence=0.0, initial_const=1e-2, binary_search_steps=steps, max_iter=iters,
              stop_early=True, stop_early_iter=4, lr=lr, batch_size=32)
    oatt, hatt = oatk.CW2(om, **kw), CW2(hm, verbose=0, **kw)
    oadv, osucc = oatt.attack(x.clone(), y)
    adv, succ = hatt.attack(x.to(DEV), y.to(DEV))
    succ, osucc = [bool(a) for a in succ], [bool(a) for a in osucc]
    d = (adv.cpu() - oadv).abs().numpy()
    l2h, l2o = (adv.cpu() - x).flatten(1).norm(dim=1), (oadv - x).flatten(1).norm(dim=1)
    log("configs[2] CW2 targeted SV x 32 x 3 s, %d search steps x %d iterations: success HIP %d/32 oracle %d/32 (equal per utterance: %s); "
        "max |x_adv - oracle| %.3e; L2 of the perturbation %.4f vs %.4f (per utterance max rel. difference %.3f)"
        % (steps, iters, sum(succ), sum(osucc), succ == osucc, d.max(), float(l2h.mean()), float(l2o.mean()),
           float(((l2h - l2o).abs() / l2o.clamp_min(1e-6)).max())))
    assert succ == osucc and 0 < sum(succ) < 32, "both outcomes: %d/32" % sum(succ)
    assert hm.make_decision(adv)[0].cpu().tolist() == om.make_decision(oadv)[0].tolist()
    assert abs(float(l2h.mean()) - float(l2o.mean())) < 0.03 * float(l2o.mean())
