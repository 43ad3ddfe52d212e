"""oracle/post.py against the reference's own save_audio / metric / set_threshold outputs (tests/golden/post_formats.npz)."""
import numpy as np

from conftest import load_golden
from oracle import post


def test_pcm_rounding_is_the_reference_one():
    g = load_golden("post_formats.npz")
    for i in range(len(g["adver"])):
        assert np.array_equal(post.save_audio_pcm(g["adver"][i]), g["pcm"][i]), i
    assert g["pcm"][2, 0] == -32768 and g["pcm"][2, 2] == 32767  # 1.0 wraps, 0.999985 does not


def test_metrics_match_reference():
    g = load_golden("post_formats.npz")
    for i in range(len(g["adver"])):
        got = np.array(post.all_metrics(g["benign"][i], g["adver"][i]), dtype=np.float64)
        want = g["metrics"][i]
        assert np.array_equal(np.isinf(got), np.isinf(want)), i
        fin = np.isfinite(want)
        np.testing.assert_allclose(got[fin], want[fin], rtol=1e-6, atol=0, err_msg=str(i))


def test_set_threshold_matches_reference():
    g = load_golden("post_formats.npz")
    for name in ("plain", "ties", "separable", "single"):
        thr, frr, far = post.set_threshold(g[name + "_target"].astype(np.float64), g[name + "_untarget"].astype(np.float64))
        assert [thr, frr, far] == g[name + "_out"].tolist(), name
