"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- numpy restatement of the data formats after the attack.

Pinned: tests/golden/post_formats.npz holds the outputs of the reference's OWN functions (extracted by AST
from attackMain.py / metric/metric.py / set_threshold.py and executed unmodified by tests/golden/make_golden.py,
because those modules import packages that are not installed: pesq, pystoi, torch_lfilter, torchaudio).
"""
import numpy as np
from numpy import linalg as LA


def save_audio_pcm(adver, bits=16):
    """attackMain.py:154-166 save_audio without the file write: (T,) float32 -> int16."""
    adver = np.asarray(adver, dtype=np.float32)
    if 0.9 * adver.max() <= 1 and 0.9 * adver.min() >= -1:
        adver = adver * (2 ** (bits - 1))
    with np.errstate(invalid="ignore"):
        return adver.astype(np.int16)


def preprocess(x, bits=16):
    """metric/metric.py:8-12: only the MAX is range-tested."""
    x = np.asarray(x, dtype=np.float32)
    if not -1 <= x.max() <= 1:
        x = x / (2 ** (bits - 1))
    return x.flatten()


def all_metrics(benign, adver, bits=16):
    """metric/metric.py:14-42: [L2, L0, L1, Linf, SNR]."""
    b, a = preprocess(benign, bits), preprocess(adver, bits)
    d = a - b
    pn = np.sum(d ** 2)
    snr = np.inf if pn <= 0.0 else 10 * np.log10(np.sum(b ** 2) / pn)
    return [LA.norm(d, 2), LA.norm(d, 0), LA.norm(d, 1), LA.norm(d, np.inf), snr]


def set_threshold(score_target, score_untarget):
    """set_threshold.py:22-47: first target score minimising |FRR - FAR| (percent)."""
    st, su = np.asarray(score_target), np.asarray(score_untarget)
    best, thr, frr_b, far_b = np.inf, 0.0, 0.0, 0.0
    for c in st:
        frr = np.argwhere(st < c).flatten().size * 100 / st.size
        far = np.argwhere(su >= c).flatten().size * 100 / su.size
        if np.abs(frr - far) < best:
            best, thr, frr_b, far_b = np.abs(frr - far), c, frr, far
    return thr, frr_b, far_b
