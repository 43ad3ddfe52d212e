"""TEST INFRASTRUCTURE (checker only; never imported by the product).

Philox4x32-10 counter-based generator -- J. Salmon, M. Moraes, R. Dror, D. Shaw, "Parallel random numbers: as
easy as 1, 2, 3", SC'11 (the Random123 library's ``philox4x32``, 10 rounds) -- restated in numpy, plus the three
noise streams the HIP engine derives from it.  The reference draws every random number from torch's / numpy's
process-global generators (adaptive_attack/NES.py:19 ``torch.randn``, torchaudio's dither ``torch.rand``,
kmeans_pytorch's ``np.random.choice`` initialisation), streams that no other implementation can reproduce; the
engine's streams are its own contract -- keyed by position, not by history -- and this file is what pins them:

  * ``philox4x32_10`` is checked against the Random123 known-answer vectors (tests/test_oracle_philox.py);
  * ``dither_noise``   restates k_mfcc.hip ``dither_draw``  (Kaldi dither, torchaudio 0.6.0 _get_window quirk: the same
                       uniform draw feeds both Box-Muller factors);
  * ``nes_normal``     restates k_attack.hip ``nes_normal`` (antithetic NES noise, NES.py:19-23);
  * ``feco_random_init`` restates k_feco.hip's random k-means initialisation (distinct frames, one key per frame).
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_LO = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Counter words c0..c3 and key words k0, k1 (broadcastable integer arrays, taken mod 2^32) -> four uint32 arrays."""
    c = [np.asarray(v).astype(np.uint64) & _LO for v in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c[0]
        p1 = _M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), p1 & _LO, (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), p0 & _LO]
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return tuple(v.astype(np.uint32) for v in c)


def _key(seed):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return seed & 0xFFFFFFFF, seed >> 32


def _uniform(word):
    """24 high bits -> (0, 1): ((w >> 8) + 0.5) / 2^24, exact in float32."""
    return ((word >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def dither_noise(seed, utt, frames, dither=1.0, win=400):
    """(frames, win) float32: the noise k_mfcc.hip adds to sample n of frame f of (global) utterance `utt`.
    counter = (n, f, utt lo, utt hi), key = seed; u = max(uniform, eps); sqrt(-2 ln u) * cos(2 pi u) * dither."""
    f, n = np.meshgrid(np.arange(frames), np.arange(win), indexing="ij")
    k0, k1 = _key(seed)
    r = philox4x32_10(n, f, int(utt) & 0xFFFFFFFF, (int(utt) >> 32) & 0xFFFFFFFF, k0, k1)[0]
    u = np.maximum(_uniform(r), np.float32(np.finfo(np.float32).eps))
    return (np.sqrt(np.float32(-2.0) * np.log(u)) * np.cos(np.float32(6.283185307179586) * u) * np.float32(dither)).astype(np.float32)


def nes_normal(seed, example, pair, T):
    """(T,) float32 standard normals of antithetic pair `pair` of (global) example `example`: Box-Muller on the first
    two words of philox(counter = (t, pair, example lo, example hi), key = seed)."""
    t = np.arange(T)
    k0, k1 = _key(seed)
    r0, r1, _, _ = philox4x32_10(t, int(pair), int(example) & 0xFFFFFFFF, (int(example) >> 32) & 0xFFFFFFFF, k0, k1)
    return (np.sqrt(np.float32(-2.0) * np.log(_uniform(r0))) * np.cos(np.float32(6.283185307179586) * _uniform(r1))).astype(np.float32)


def feco_random_init(seed, utt, F, k):
    """Frames that initialise the k centroids of (global) utterance `utt`: every frame f draws the 32-bit key
    philox(counter = (f, 0, utt lo, utt hi), key = seed)[0]; the frames are ranked by (key, f) ascending and
    centroid j starts at the frame of rank j -- k distinct frames, a uniformly random subset in random order
    (what kmeans_pytorch's ``np.random.choice(F, k, replace=False)`` draws from numpy's global generator)."""
    k0, k1 = _key(seed)
    keys = philox4x32_10(np.arange(F), 0, int(utt) & 0xFFFFFFFF, (int(utt) >> 32) & 0xFFFFFFFF, k0, k1)[0]
    order = np.lexsort((np.arange(F), keys))
    return order[:k].astype(np.int64)
