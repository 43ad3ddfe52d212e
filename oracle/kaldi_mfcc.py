"""Kaldi-compatible MFCC, restated.  TEST INFRASTRUCTURE (see oracle/__init__.py).

PARITY UNPINNED: the reference calls ``torchaudio.compliance.kaldi.mfcc`` (reference
model/xv_plda.py:114-148) from torchaudio==0.6.0 (reference README.md:54), a third-party
dependency that is neither vendored under /root/reference nor installed in this image.  This
file restates that published algorithm (Kaldi ``feature-window`` / ``feature-mfcc`` /
``mel-computations`` as torchaudio/compliance/kaldi.py v0.6.0 implements them) with exactly the
keyword values of the reference call site.  The reference repo holds no test or golden vector
for this boundary, so nothing OF THE REFERENCE pins it.

Corroboration (round 2, not a pin): an independent implementation of the same specification --
``transformers.audio_utils`` (its numpy stand-in for ``torchaudio.compliance.kaldi.fbank``) +
``scipy.fft.dct`` -- agrees with this file to 9e-6 on the log-mel energies and 3e-4 on cepstra up
to 40 (the latter is torchaudio's own fp32 table rounding); vectors in
tests/golden/frontend_xcheck.npz, checked by tests/test_oracle_frontends.py.  Still restated only:
the snip_edges=False reflect padding, c0 <- log raw energy, the lifter.

Everything is differentiable torch so that ``torch.autograd`` of this file is the checker for
the hand-written HIP backward.
"""
import math

import torch

EPSILON = torch.finfo(torch.float32).eps  # kaldi.py: EPSILON = torch.tensor(torch.finfo(torch.float).eps)

# keyword values at reference model/xv_plda.py:116-148
SAMPLE_FREQUENCY = 16000.0
FRAME_SHIFT_MS = 10.0
FRAME_LENGTH_MS = 25.0
PREEMPH = 0.97
NUM_MEL_BINS = 30
LOW_FREQ = 20.0
HIGH_FREQ = 7600.0
NUM_CEPS = 30
CEPSTRAL_LIFTER = 22.0
ENERGY_FLOOR = 0.0

WINDOW_SHIFT = int(SAMPLE_FREQUENCY * FRAME_SHIFT_MS * 0.001)  # 160
WINDOW_SIZE = int(SAMPLE_FREQUENCY * FRAME_LENGTH_MS * 0.001)  # 400
PADDED_WINDOW_SIZE = 512  # round_to_power_of_two=True


def num_frames(num_samples):
    """snip_edges=False: kaldi.py _get_strided, m = (num_samples + shift // 2) // shift."""
    return (num_samples + WINDOW_SHIFT // 2) // WINDOW_SHIFT


def get_strided(waveform):
    """kaldi.py _get_strided with snip_edges=False: reflect both edges, frames of WINDOW_SIZE.

    waveform: (num_samples,) -> (m, WINDOW_SIZE)
    """
    n = waveform.shape[0]
    m = num_frames(n)
    rev = torch.flip(waveform, [0])
    pad = WINDOW_SIZE // 2 - WINDOW_SHIFT // 2  # 120
    pad_left = rev[-pad:]
    padded = torch.cat((pad_left, waveform, rev), dim=0)
    idx = (torch.arange(m).unsqueeze(1) * WINDOW_SHIFT + torch.arange(WINDOW_SIZE).unsqueeze(0))
    return padded[idx]


def povey_window():
    """kaldi.py _feature_window_function('povey'): hann(N, periodic=False) ** 0.85."""
    return torch.hann_window(WINDOW_SIZE, periodic=False, dtype=torch.float32).pow(0.85)


def mel_scale(freq):
    return 1127.0 * (1.0 + freq / 700.0).log()


def mel_scale_scalar(freq):
    return 1127.0 * math.log(1.0 + freq / 700.0)


def get_mel_banks():
    """kaldi.py get_mel_banks (vtln_warp == 1.0): (NUM_MEL_BINS, PADDED/2) triangular weights."""
    num_fft_bins = PADDED_WINDOW_SIZE // 2
    fft_bin_width = SAMPLE_FREQUENCY / PADDED_WINDOW_SIZE
    mel_low = mel_scale_scalar(LOW_FREQ)
    mel_high = mel_scale_scalar(HIGH_FREQ)
    delta = (mel_high - mel_low) / (NUM_MEL_BINS + 1)
    b = torch.arange(NUM_MEL_BINS, dtype=torch.float32).unsqueeze(1)
    left = mel_low + b * delta
    center = mel_low + (b + 1.0) * delta
    right = mel_low + (b + 2.0) * delta
    mel = mel_scale(fft_bin_width * torch.arange(num_fft_bins, dtype=torch.float32)).unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    return torch.max(torch.zeros(1), torch.min(up, down))


def get_dct_matrix():
    """kaldi.py _get_dct_matrix: (NUM_MEL_BINS, NUM_CEPS), ortho DCT-II, first column sqrt(1/N)."""
    n = torch.arange(float(NUM_MEL_BINS))
    k = torch.arange(float(NUM_MEL_BINS)).unsqueeze(1)
    dct = torch.cos(math.pi / float(NUM_MEL_BINS) * (n + 0.5) * k)  # (n_mfcc, n_mels)
    dct[0] *= 1.0 / math.sqrt(2.0)
    dct *= math.sqrt(2.0 / float(NUM_MEL_BINS))
    dct = dct.t().contiguous()
    dct[:, 0] = math.sqrt(1 / float(NUM_MEL_BINS))
    return dct[:, :NUM_CEPS]


def get_lifter_coeffs():
    i = torch.arange(NUM_CEPS, dtype=torch.float32)
    return 1.0 + 0.5 * CEPSTRAL_LIFTER * torch.sin(math.pi * i / CEPSTRAL_LIFTER)


_CONST = {}


def constants():
    if not _CONST:
        _CONST["window"] = povey_window()
        _CONST["mel"] = get_mel_banks()
        _CONST["dct"] = get_dct_matrix()
        _CONST["lifter"] = get_lifter_coeffs()
    return _CONST


def dither_noise_from_uniform(u):
    """kaldi.py _get_window dither: the SAME uniform draw feeds both factors (v0.6.0 quirk)."""
    x = torch.clamp(u, min=EPSILON)
    return torch.sqrt(-2 * x.log()) * torch.cos(2 * math.pi * x)


def mfcc(waveform, dither_noise=None):
    """One utterance: waveform (1, num_samples) or (num_samples,), int16-scaled floats -> (m, 30).

    ``dither_noise``: None for dither=0, else an (m, 400) tensor that is ADDED to the strided
    frames (the reference hard-codes dither=1.0 and draws from the global RNG, xv_plda.py:119;
    here the draw is an explicit input so two implementations can be compared).
    """
    c = constants()
    if waveform.dtype != torch.float32:  # fp64 "truth" runs: same fp32-rounded constants, wider arithmetic
        c = {k: v.to(waveform.dtype) for k, v in c.items()}
    if waveform.dim() == 2:
        waveform = waveform[0]
    frames = get_strided(waveform)
    if dither_noise is not None:
        frames = frames + dither_noise
    # remove_dc_offset
    frames = frames - frames.mean(dim=1, keepdim=True)
    # raw_energy=True, energy_floor=0
    energy = torch.clamp(frames.pow(2).sum(1), min=EPSILON).log()
    # preemphasis with replicate pad on the left
    prev = torch.cat((frames[:, :1], frames[:, :-1]), dim=1)
    frames = frames - PREEMPH * prev
    frames = frames * c["window"].unsqueeze(0)
    frames = torch.nn.functional.pad(frames, (0, PADDED_WINDOW_SIZE - WINDOW_SIZE))
    spec = torch.fft.rfft(frames, dim=1)
    power = spec.real.pow(2) + spec.imag.pow(2)  # (m, 257)
    mel = torch.nn.functional.pad(c["mel"], (0, 1))  # (30, 257), Nyquist column zero
    mel_energies = (power.unsqueeze(1) * mel.unsqueeze(0)).sum(dim=2)  # (m, 30)
    mel_energies = torch.clamp(mel_energies, min=EPSILON).log()
    feature = mel_energies.matmul(c["dct"])
    feature = feature * c["lifter"].unsqueeze(0)
    # use_energy=True, htk_compat=False: c0 <- log energy
    feature = torch.cat((energy.unsqueeze(1), feature[:, 1:]), dim=1)
    return feature


def mfcc_batch(x, dither_noise=None):
    """x: (B, 1, T) -> (B, m, 30); mirrors the per-utterance loop at reference xv_plda.py:112."""
    out = []
    for b in range(x.shape[0]):
        out.append(mfcc(x[b], None if dither_noise is None else dither_noise[b]))
    return torch.stack(out, 0)
