"""The engine's counter-based noise streams are pinned through oracle/philox.py: the generator itself against the
Random123 known-answer vectors (kat_vectors of the library that published Philox: philox4x32, 10 rounds), the
derived streams for shape / range / statistics.  The device kernels are compared with the same restatement in
tests/test_gpu_xv.py (NES normals, dither) and tests/test_gpu_feco.py (random k-means initialisation)."""
import numpy as np

from oracle import philox


def _hex(words):
    return ["%08x" % int(w) for w in words]


def test_philox4x32_10_known_answers():
    f = 0xFFFFFFFF
    assert _hex(philox.philox4x32_10(0, 0, 0, 0, 0, 0)) == ["6627e8d5", "e169c58d", "bc57ac4c", "9b00dbd8"]
    assert _hex(philox.philox4x32_10(f, f, f, f, f, f)) == ["408f276d", "41c83b0e", "a20bc7c6", "6d5451fd"]
    # counter and key = digits of pi
    assert _hex(philox.philox4x32_10(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0)) == [
        "d16cfe09", "94fdcceb", "5001e420", "24126ea1"]


def test_vectorised_counters_equal_scalar_calls():
    c0 = np.arange(5)
    got = philox.philox4x32_10(c0, 7, 1 << 40, 3, 0x1234, 0xABCD)  # words are taken mod 2^32
    for i in range(5):
        one = philox.philox4x32_10(i, 7, 0, 3, 0x1234, 0xABCD)
        assert [int(g[i]) for g in got] == [int(o) for o in one]


def test_streams_are_keyed_by_position():
    z = philox.nes_normal(11, example=3, pair=2, T=20000)
    assert z.dtype == np.float32 and abs(float(z.mean())) < 0.03 and abs(float(z.std()) - 1) < 0.03
    assert np.array_equal(z, philox.nes_normal(11, 3, 2, 20000))
    assert not np.array_equal(z, philox.nes_normal(11, 4, 2, 20000)) and not np.array_equal(z, philox.nes_normal(12, 3, 2, 20000))
    d = philox.dither_noise(5, utt=1, frames=50, dither=1.0)
    assert d.shape == (50, 400) and np.isfinite(d).all()
    # torchaudio 0.6.0 feeds ONE uniform to both Box-Muller factors: |noise| <= sqrt(-2 ln u) with the same u
    assert float(np.abs(d).max()) < 6 and not np.array_equal(d, philox.dither_noise(5, 2, 50))


def test_feco_random_init_is_a_random_subset():
    a = philox.feco_random_init(9, utt=0, F=300, k=150)
    assert a.shape == (150,) and len(set(a.tolist())) == 150 and a.min() >= 0 and a.max() < 300
    assert not np.array_equal(a, np.sort(a))  # random order, not the evenly spaced default
    assert not np.array_equal(a, philox.feco_random_init(9, 1, 300, 150))
    hits = np.zeros(300)
    for s in range(200):
        hits[philox.feco_random_init(s, 0, 300, 150)] += 1
    assert 70 < hits.min() and hits.max() < 130  # every frame is picked about half of the time
