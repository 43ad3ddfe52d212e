"""Generate tests/golden/an_ref.npz and tests/golden/frontend_xcheck.npz (build container only).

    python tests/golden/make_golden_frontends.py      # needs /root/reference and the `transformers` package

Two different kinds of evidence for the two front-ends round 1 had to leave "parity unpinned":

(1) ``an_ref.npz`` -- the REFERENCE's own AudioNet code, executed unmodified:
    model/audionet_csine.py (class audionet_csine: conv stack, extract_emb, forward, make_decision) and
    model/_audionet/Preprocessor.py (pre-emphasis, torch.stft call, _square, mel matmul, 10 log10), plus
    attack/utils.py SEC4SR_CrossEntropy and torch autograd for d loss/d wav and d loss/d log-mel.
    The class cannot be built in this image as is; two harness-side accommodations make it run,
    both disclosed in the fixture's ``meta``:
      * ``librosa`` (0.8.0, README.md:56) is not installed.  A placeholder module is put in ``sys.modules`` whose
        ``filters.mel(sr, n_fft, n_mels, fmin, fmax)`` returns
        ``transformers.audio_utils.mel_filter_bank(..., norm="slaney", mel_scale="slaney").T`` -- a THIRD-PARTY
        implementation of the same published function (written to reproduce librosa's defaults), not code of this
        repository.  It is consumed once, by ``Preprocessor.__init__`` (Preprocessor.py:57-64).
      * ``torch.stft`` without ``return_complex`` (Preprocessor.py:100-105) raises on torch >= 2.0.  During the
        reference calls ``torch.stft`` is wrapped so that a call WITHOUT ``return_complex`` gets
        ``return_complex=True`` followed by ``torch.view_as_real`` -- exactly the (..., 2) real view torch < 1.8
        returned; every other argument (centre, reflect padding, one-sided, window placement) is the reference's.
    So the mel basis is corroborated-third-party, everything else in the fixture is reference arithmetic.

(2) ``frontend_xcheck.npz`` -- INDEPENDENT implementations of the two published front-end algorithms, evaluated on
    seeded waveforms: ``transformers.audio_utils`` (its numpy replacement for ``torchaudio.compliance.kaldi.fbank``:
    povey window, kaldi mel bank triangularised in mel space, DC removal, 0.97 pre-emphasis, log floor eps) +
    ``scipy.fft.dct(norm='ortho')`` for the cepstra.  torchaudio==0.6.0 itself (reference README.md:54) is not
    installable here, so the Kaldi MFCC stays formally unpinned BY THE REFERENCE; this fixture shows the restatement
    agrees with an implementation written by someone else against the same specification.  Not covered by it:
    the reflect padding of snip_edges=False (taken from the restatement) and the energy/lifter lines (standard
    formulas, restated).
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import scipy.fft  # noqa: E402
import torch  # noqa: E402
import transformers  # noqa: E402
from transformers import audio_utils as au  # noqa: E402

from speakerguard_amd import synth  # noqa: E402

META = {
    "generator": "tests/golden/make_golden_frontends.py",
    "reference": "SpeakerGuard @ 2024-12-20 (/root/reference)",
    "torch": torch.__version__, "numpy": np.__version__, "transformers": transformers.__version__,
}


def save(name, meta, **arrays):
    m = dict(META)
    m.update(meta)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, meta=json.dumps(m), **arrays)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


# ------------------------------------------------------------------------------------------------ (1)
def _install_librosa_placeholder():
    def mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, **kw):
        assert not kw, kw
        fmax = sr / 2.0 if fmax is None else fmax
        return au.mel_filter_bank(num_frequency_bins=1 + n_fft // 2, num_mel_filters=n_mels, min_frequency=fmin,
                                  max_frequency=fmax, sampling_rate=sr, norm="slaney", mel_scale="slaney").T.astype(np.float32)
    lib = types.ModuleType("librosa")
    lib.filters = types.ModuleType("librosa.filters")
    lib.filters.mel = mel
    sys.modules["librosa"] = lib
    sys.modules["librosa.filters"] = lib.filters


class _LegacyStft:
    """torch.stft of torch < 1.8: no return_complex argument, result is the (..., 2) real view."""

    def __enter__(self):
        self.orig = torch.stft

        def stft(*a, **k):
            if "return_complex" in k:
                return self.orig(*a, **k)
            return torch.view_as_real(self.orig(*a, return_complex=True, **k))
        torch.stft = stft
        return self

    def __exit__(self, *exc):
        torch.stft = self.orig


def gen_audionet_reference():
    np.infty = np.inf  # audionet_csine.py:121 (NumPy 2 removed the alias)
    _install_librosa_placeholder()
    sys.path.insert(0, REF)
    from attack.utils import SEC4SR_CrossEntropy
    from model.audionet_csine import audionet_csine

    num_class = 251
    sd = synth.make_audionet_state_dict(seed=0, num_class=num_class)
    torch.manual_seed(0)
    model = audionet_csine(num_class=num_class)
    # the reference class as published defines conv2..conv8 and fc in __init__ lines the survey cites (:66-118)
    missing = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    # the synthetic state_dict has no entries for the Preprocessor's constant matrices: they keep their constructed values
    assert sorted(missing.missing_keys) == ['prep._inverse_mel', 'prep.mel_basis'] and not missing.unexpected_keys, missing
    model.eval()
    for p in model.parameters():
        p.requires_grad_(False)
    out = {}
    with _LegacyStft():
        for tag, B, T, seed in (("t48000", 2, 48000, 61), ("t20011", 2, 20011, 62), ("tones", 8, 24000, 9)):
            x = torch.from_numpy(synth.make_tone_waveforms(B, T, seed) if tag == "tones" else synth.make_waveforms(B, T, seed=seed))
            xin = x.clone().requires_grad_(True)
            feats = model.compute_feat(xin, flag=1)                      # reference Preprocessor, (B, F, 32)
            decisions, scores = model.make_decision(xin, flag=0)
            emb = model.embedding(xin, flag=0)
            y = (decisions + 1) % num_class                               # untargeted CE away from a wrong label
            ce = SEC4SR_CrossEntropy(reduction="none", task="CSI")(scores, y)
            ce.backward(torch.ones_like(ce))
            grad_wav = xin.grad.clone()
            f_in = feats.detach().clone().requires_grad_(True)
            d1, s1 = model.make_decision(f_in, flag=1)
            ce1 = SEC4SR_CrossEntropy(reduction="none", task="CSI")(s1, y)
            ce1.backward(torch.ones_like(ce1))
            # inputs are re-created from the synth seeds by the tests (guarded by a checksum); the waveform gradient is
            # kept every third sample plus its exact L1 / L2 norms -- keeps the fixture small
            gw = grad_wav.numpy()
            out.update({tag + "_x_sha256": np.array(hashlib.sha256(x.numpy().tobytes()).hexdigest()),
                        tag + "_gen": np.array([B, T, seed]), tag + "_feats": feats.detach().numpy(),
                        tag + "_scores": scores.detach().numpy(),
                        tag + "_decisions": decisions.numpy(), tag + "_emb": emb.detach().numpy(), tag + "_y": y.numpy(),
                        tag + "_ce": ce.detach().numpy(), tag + "_grad_wav_sub3": gw[..., ::3].copy(),
                        tag + "_grad_wav_norms": np.array([np.abs(gw.astype(np.float64)).sum(), np.sqrt((gw.astype(np.float64) ** 2).sum())]),
                        tag + "_grad_feats": f_in.grad.numpy(), tag + "_scores_from_feats": s1.detach().numpy()})
            print(tag, "frames", feats.shape[1], "decisions", decisions.tolist(), "ce", ce.tolist())
        # int16-scaled input: check_input_range('scale') divides by 2^15 (model/utils.py:15-16)
        x16 = torch.from_numpy(synth.make_waveforms(2, 48000, seed=63)) * 32768.0
        with torch.no_grad():
            d16, s16 = model.make_decision(x16, flag=0)
        out.update({"int16_scores": s16.numpy(), "int16_decisions": d16.numpy()})  # input: make_waveforms(2, 48000, seed=63) * 2^15
    out["mel_basis"] = model.prep.mel_basis.detach().numpy()              # (513, 32) as the reference holds it
    save("an_ref.npz", {
        "what": "reference model/audionet_csine.py + model/_audionet/Preprocessor.py + attack/utils.py executed unmodified",
        "weights": "speakerguard_amd.synth.make_audionet_state_dict(seed=0, num_class=251)",
        "accommodations": [
            "numpy.infty alias",
            "librosa placeholder: filters.mel -> transformers.audio_utils.mel_filter_bank(norm='slaney', mel_scale='slaney').T "
            "(third-party implementation of the published librosa function; consumed only by Preprocessor.__init__)",
            "torch.stft legacy return: calls without return_complex get return_complex=True + torch.view_as_real",
        ]}, **out)


# ------------------------------------------------------------------------------------------------ (2)
def gen_frontend_xcheck():
    from oracle import kaldi_mfcc as K  # ONLY for num_frames and the reflect-padding rule, see the docstring
    out = {}
    eps = float(np.finfo(np.float32).eps)
    bank = au.mel_filter_bank(num_frequency_bins=257, num_mel_filters=30, min_frequency=20, max_frequency=7600,
                              sampling_rate=16000, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    win = au.window_function(400, "povey", periodic=False)
    lifter = 1.0 + 11.0 * np.sin(np.pi * np.arange(30) / 22.0)
    for tag, T, seed in (("t48000", 48000, 71), ("t16123", 16123, 72)):
        x = (synth.make_waveforms(1, T, seed=seed)[0, 0] * 32768.0).astype(np.float32)
        m = K.num_frames(T)
        rev = x[::-1]
        padded = np.concatenate((rev[-120:], x, rev)).astype(np.float64)   # kaldi.py _get_strided, snip_edges=False
        sig = padded[:(m - 1) * 160 + 400]
        logmel = au.spectrogram(sig, win, frame_length=400, hop_length=160, fft_length=512, power=2.0, center=False,
                                preemphasis=0.97, mel_filters=bank, log_mel="log", mel_floor=eps, remove_dc_offset=True,
                                dtype=np.float64).T                          # (m, 30)
        ceps = scipy.fft.dct(logmel, type=2, norm="ortho", axis=-1) * lifter
        frames = np.stack([sig[i * 160:i * 160 + 400] for i in range(m)])
        frames = frames - frames.mean(1, keepdims=True)
        ceps[:, 0] = np.log(np.maximum((frames ** 2).sum(1), eps))          # use_energy, raw_energy: c0 <- log energy
        out.update({tag + "_gen": np.array([T, seed]), tag + "_x_sha256": np.array(hashlib.sha256(x.tobytes()).hexdigest()),
                    tag + "_logmel": logmel, tag + "_mfcc": ceps})   # input: make_waveforms(1, T, seed)[0, 0] * 2^15 as float32
    # AudioNet front-end: librosa-style slaney mel on a centred, reflect-padded, periodic-hann(800)-in-1024 STFT
    bank_an = au.mel_filter_bank(num_frequency_bins=513, num_mel_filters=32, min_frequency=0, max_frequency=8000,
                                 sampling_rate=16000, norm="slaney", mel_scale="slaney")
    win_an = au.window_function(800, "hann", periodic=True, frame_length=1024, center=True)
    x = synth.make_waveforms(1, 48000, seed=73)[0, 0].astype(np.float64)   # tests re-create it from the seed
    pre = x[1:] - 0.97 * x[:-1]
    mel = au.spectrogram(pre, win_an, frame_length=1024, hop_length=160, fft_length=1024, power=2.0, center=True,
                         pad_mode="reflect", mel_filters=bank_an, mel_floor=1e-16, log_mel=None, dtype=np.float64)
    out.update({"an_logmel": 10.0 * np.log10(np.maximum(mel, 1e-16)).T, "an_mel_basis": bank_an.T})
    save("frontend_xcheck.npz", {
        "what": "independent implementations (transformers.audio_utils + scipy.fft.dct) of Kaldi MFCC (kwargs of "
                "reference model/xv_plda.py:116-148) and of the AudioNet log-mel front-end (Preprocessor.py:13-23,88-112)",
        "not_independent": "snip_edges=False reflect padding, c0 <- log raw energy and the lifter are restated here"}, **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    gen_frontend_xcheck()
    gen_audionet_reference()
