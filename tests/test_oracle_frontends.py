"""The two front-ends round 1 left "parity unpinned", now checked on the CPU:

* oracle/audionet.py against tests/golden/an_ref.npz -- outputs of the REFERENCE's own audionet_csine /
  Preprocessor / SEC4SR_CrossEntropy code (run by tests/golden/make_golden_frontends.py with two disclosed harness
  accommodations: a third-party mel basis in place of the uninstalled librosa, the pre-1.8 return convention of
  torch.stft).  The restatement must reproduce them to fp32 round-off on features, logits, decisions, the loss
  and both gradients.
* oracle/kaldi_mfcc.py and the AudioNet log-mel against tests/golden/frontend_xcheck.npz -- INDEPENDENT
  implementations (transformers.audio_utils + scipy.fft.dct).  Not reference outputs: torchaudio==0.6.0 is not
  installable here, the Kaldi MFCC stays "unpinned by the reference, corroborated by an independent implementation".
"""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import attacks as oatk
from oracle import audionet as A
from oracle import kaldi_mfcc as K
from speakerguard_amd import synth


def an_inputs(g, tag):
    B, T, seed = (int(v) for v in g[tag + "_gen"])
    x = synth.make_tone_waveforms(B, T, seed) if tag == "tones" else synth.make_waveforms(B, T, seed=seed)
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g[tag + "_x_sha256"]), "synthetic inputs changed: regenerate the fixture"
    return torch.from_numpy(x)


@pytest.fixture(scope="module")
def an_oracle():
    return A.AudioNet(synth.make_audionet_state_dict(seed=0, num_class=251))


def test_fixture_discloses_its_accommodations():
    meta = load_golden("an_ref.npz")["meta"]
    assert len(meta["accommodations"]) == 3 and "librosa placeholder" in meta["accommodations"][1]


def test_mel_basis_is_the_one_the_reference_held():
    g = load_golden("an_ref.npz")
    np.testing.assert_allclose(A.mel_basis().T, g["mel_basis"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("tag", ["t48000", "t20011", "tones"])
def test_audionet_oracle_reproduces_reference_run(an_oracle, tag):
    g = load_golden("an_ref.npz")
    x = an_inputs(g, tag).requires_grad_(True)
    feats = an_oracle.compute_feat(x, 1)
    np.testing.assert_allclose(feats.detach().numpy(), g[tag + "_feats"], rtol=0, atol=1e-4)          # dB, values to -160
    dec, scores = an_oracle.make_decision(x)
    assert dec.tolist() == g[tag + "_decisions"].tolist()
    np.testing.assert_allclose(scores.detach().numpy(), g[tag + "_scores"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(an_oracle.embedding(x.detach()).numpy(), g[tag + "_emb"], rtol=1e-5, atol=1e-5)
    y = torch.from_numpy(g[tag + "_y"])
    ce = oatk.cross_entropy_loss(scores, y)
    np.testing.assert_allclose(ce.detach().numpy(), g[tag + "_ce"], rtol=1e-5, atol=1e-5)
    ce.backward(torch.ones_like(ce))
    gw = x.grad.numpy()
    ref = g[tag + "_grad_wav_sub3"]
    assert np.abs(gw[..., ::3] - ref).max() <= 1e-5 * np.abs(ref).max()
    assert np.mean(np.sign(gw[..., ::3]) != np.sign(ref)) < 1e-4                                     # what a sign step sees
    l1, l2 = g[tag + "_grad_wav_norms"]
    assert abs(np.abs(gw.astype(np.float64)).sum() - l1) <= 1e-5 * l1
    assert abs(np.sqrt((gw.astype(np.float64) ** 2).sum()) - l2) <= 1e-5 * l2
    f = torch.from_numpy(g[tag + "_feats"]).requires_grad_(True)
    d1, s1 = an_oracle.make_decision(f, flag=1)
    np.testing.assert_allclose(s1.detach().numpy(), g[tag + "_scores_from_feats"], rtol=1e-5, atol=1e-5)
    c1 = oatk.cross_entropy_loss(s1, y)
    c1.backward(torch.ones_like(c1))
    gf = g[tag + "_grad_feats"]
    assert np.abs(f.grad.numpy() - gf).max() <= 1e-5 * np.abs(gf).max()


def test_audionet_int16_range_rule(an_oracle):
    g = load_golden("an_ref.npz")
    x16 = torch.from_numpy(synth.make_waveforms(2, 48000, seed=63)) * 32768.0
    with torch.no_grad():
        d, s = an_oracle.make_decision(x16)
    assert d.tolist() == g["int16_decisions"].tolist()
    np.testing.assert_allclose(s.numpy(), g["int16_scores"], rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------- independent implementations
def xcheck_wave(g, tag):
    T, seed = (int(v) for v in g[tag + "_gen"])
    x = (synth.make_waveforms(1, T, seed=seed)[0, 0] * 32768.0).astype(np.float32)
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g[tag + "_x_sha256"])
    return x


@pytest.mark.parametrize("tag", ["t48000", "t16123"])
def test_kaldi_mfcc_restatement_agrees_with_independent_implementation(tag):
    g = load_golden("frontend_xcheck.npz")
    x = torch.from_numpy(xcheck_wave(g, tag))
    want = g[tag + "_mfcc"]
    # fp64 arithmetic on the restatement's fp32-rounded tables (torchaudio builds them in fp32): the table rounding
    # alone moves a cepstrum by up to 3e-4 (trap 1)
    got64 = K.mfcc(x.double()).numpy()
    assert got64.shape == want.shape
    assert np.abs(got64 - want).max() < 5e-4, np.abs(got64 - want).max()
    got32 = K.mfcc(x).numpy()
    assert np.abs(got32 - want).max() < 1e-3
    # c0 (log energy) and the high cepstra separately: a wrong lifter / DCT row would show up here
    assert np.abs(got64[:, 0] - want[:, 0]).max() < 1e-6 * np.abs(want[:, 0]).max() + 1e-9
    assert np.abs(got64[:, 20:] - want[:, 20:]).max() < 5e-4


def test_audionet_logmel_restatement_agrees_with_independent_implementation():
    g = load_golden("frontend_xcheck.npz")
    x = torch.from_numpy(synth.make_waveforms(1, 48000, seed=73)[0, 0])[None]
    got = A.preprocess(x.double()).numpy()[0].T
    assert got.shape == g["an_logmel"].shape
    assert np.abs(got - g["an_logmel"]).max() < 1e-4          # dB; the float32 rounding of the mel basis is 5e-5 dB
    np.testing.assert_allclose(A.mel_basis(), g["an_mel_basis"], rtol=0, atol=1e-8)


def test_live_transformers_cross_check_when_available():
    """Same comparison against the installed package itself (not only the committed vectors)."""
    au = pytest.importorskip("transformers.audio_utils")
    bank = au.mel_filter_bank(num_frequency_bins=257, num_mel_filters=30, min_frequency=20, max_frequency=7600,
                              sampling_rate=16000, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    np.testing.assert_allclose(bank.T[:, :256], K.get_mel_banks().numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(au.window_function(400, "povey", periodic=False), K.povey_window().numpy(), rtol=0, atol=1e-6)
    import scipy.fft
    np.testing.assert_allclose(scipy.fft.dct(np.eye(30), type=2, norm="ortho", axis=-1), K.get_dct_matrix().numpy(), rtol=0, atol=5e-6)
