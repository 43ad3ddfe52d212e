"""SURVEY 8(f) N3/N4 through the C-ABI (sg_wav_finalize, sg_eer_threshold) against fixtures the REFERENCE's own
functions produced: int16 PCM and the integer-valued results bit-exact, floating-point metrics to 1e-6."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_pcm_bit_exact_and_wav_files(tmp_path):
    from scipy.io import wavfile
    from speakerguard_amd.audio_io import quantize_pcm, save_audio
    g = load_golden("post_formats.npz")
    adver = torch.from_numpy(g["adver"]).unsqueeze(1).cuda()
    assert np.array_equal(quantize_pcm(adver), g["pcm"])
    names = ["id%02d-utt%d" % (i, i) for i in range(len(g["adver"]))]
    save_audio(adver, names, str(tmp_path))
    for i, n in enumerate(names):
        fs, data = wavfile.read(os.path.join(str(tmp_path), n.split("-")[0], n + ".wav"))
        assert fs == 16000 and data.dtype == np.int16 and np.array_equal(data, g["pcm"][i])


def test_metrics_match_reference():
    from speakerguard_amd.metric import metric
    g = load_golden("post_formats.npz")
    got = metric.batch_metrics(torch.from_numpy(g["benign"]).unsqueeze(1), torch.from_numpy(g["adver"]).unsqueeze(1))
    want = g["metrics"]
    assert np.array_equal(got[:, 1], want[:, 1])                               # L0: exact count
    assert np.array_equal(got[:, 3].astype(np.float32), want[:, 3].astype(np.float32))  # Linf: exact in fp32
    assert np.array_equal(np.isinf(got), np.isinf(want))                       # zero perturbation -> SNR inf
    np.testing.assert_allclose(got[:, :4], want[:, :4], rtol=1e-6)             # reference sums in float32
    fin = np.isfinite(want[:, 4])
    # SNR in dB: the reference's float32 power sums carry ~1e-7 relative error, i.e. ~5e-7 dB absolute,
    # which is a large RELATIVE error when the ratio is near 1 (row 7: 2.7e-4 dB)
    np.testing.assert_allclose(got[fin, 4], want[fin, 4], rtol=1e-6, atol=1e-5)
    i = 0
    one = metric.get_all_metric(torch.from_numpy(g["benign"][i:i + 1]), torch.from_numpy(g["adver"][i:i + 1]))
    np.testing.assert_allclose(one, want[i], rtol=1e-6, atol=1e-5)
    assert metric.Linf(torch.from_numpy(g["benign"][5:6]), torch.from_numpy(g["adver"][5:6])) == pytest.approx(want[5, 3], rel=1e-7)
    with pytest.raises(NotImplementedError):
        metric.PESQ(None, None)


def test_full_size_properties():
    """BASELINE size (64 x 48000): PCM of a float-domain batch equals trunc(x * 2^15) mod 2^16, metrics of a
    PGD-style perturbation obey Linf <= eps and L2 <= eps * sqrt(T), L0 <= T."""
    from speakerguard_amd import synth
    from speakerguard_amd.audio_io import quantize_pcm
    from speakerguard_amd.metric import metric
    x = torch.from_numpy(synth.make_waveforms(64, 48000, seed=3)).cuda()
    eps = 0.002
    adv = torch.clamp(x + eps * torch.sign(torch.randn_like(x)), -1, 1)
    pcm = quantize_pcm(adv)
    want = np.trunc(adv.squeeze(1).cpu().numpy() * np.float32(32768)).astype(np.int64)
    assert np.array_equal(pcm.astype(np.int64), ((want + 32768) % 65536) - 32768)
    m = metric.batch_metrics(x, adv)
    assert (m[:, 3] <= eps + 1e-7).all() and (m[:, 0] <= (eps + 1e-7) * np.sqrt(48000)).all()  # fp32 rounding of x + eps
    assert (m[:, 1] <= 48000).all() and (m[:, 4] > 20).all()


def test_eer_threshold_matches_reference():
    from speakerguard_amd.set_threshold import set_threshold
    g = load_golden("post_formats.npz")
    for name in ("plain", "ties", "separable", "single"):
        got = list(set_threshold(g[name + "_target"], g[name + "_untarget"]))
        assert got == g[name + "_out"].tolist(), (name, got, g[name + "_out"].tolist())


def test_eer_threshold_large_is_consistent():
    """10^4 x 3*10^4 scores: the returned threshold is one of the targets, FRR/FAR recompute exactly, and no
    other target has a strictly smaller |FRR - FAR| before it."""
    from speakerguard_amd.set_threshold import set_threshold
    rs = np.random.RandomState(1)
    st = (rs.randn(10000) * 2 + 1.5).astype(np.float32)
    su = (rs.randn(30000) * 2 - 1.5).astype(np.float32)
    thr, frr, far = set_threshold(st, su)
    i = int(np.nonzero(st.astype(np.float64) == thr)[0][0])
    assert frr == (st < st[i]).sum() * 100 / st.size and far == (su >= st[i]).sum() * 100 / su.size
    ss, us = np.sort(st), np.sort(su)
    d = np.abs(np.searchsorted(ss, st, side="left") * 100 / st.size - (su.size - np.searchsorted(us, st, side="left")) * 100 / su.size)
    assert d[i] == d.min() and int(np.argmin(d)) == i


def test_enroll_helpers_match_oracle(xv_weights):
    """enroll.py:49-92 on the native forward path vs the oracle model: mean embedding and z-norm statistics."""
    from oracle.xv_plda import XvPlda
    from speakerguard_amd import synth
    from speakerguard_amd.enroll import enroll_speaker, speaker_model_line, znorm_stats
    from speakerguard_amd.model.xv_plda import xv_plda
    hm = xv_plda.from_weights(xv_weights, device=torch.device("cuda:0"), dither=0.0)
    om = XvPlda(xv_weights, threshold=None)
    utts = [torch.from_numpy(synth.make_waveforms(1, n, seed=50 + k))[0] for k, n in enumerate((16000, 20000, 12000))]
    tests = [torch.from_numpy(synth.make_waveforms(1, 16000, seed=60 + k))[0] for k in range(3)]
    before = hm.make_decision(utts[0].unsqueeze(0).cuda())[1].cpu()
    emb = enroll_speaker(hm, utts)
    with torch.no_grad():
        oemb = sum(om.embedding(u.unsqueeze(0)) for u in utts) / len(utts)
        oscores = [float(om.score(t.unsqueeze(0), enroll_embs=oemb).flatten()[0]) for t in tests]
    assert emb.shape == (1, oemb.shape[-1])
    assert (emb.cpu() - oemb).abs().max().item() < 2e-3 * oemb.abs().max().item()
    mean, std = znorm_stats(hm, emb, tests)
    assert abs(mean - np.mean(oscores)) < 5e-3 * abs(np.mean(oscores)) + 1e-2 and abs(std - np.std(oscores)) < 5e-2
    after = hm.make_decision(utts[0].unsqueeze(0).cuda())[1].cpu()
    assert torch.equal(before, after), "znorm_stats must restore the enrolled speakers"
    assert speaker_model_line("id1", "/p/id1.xv", 1.5, 0.25) == "id1 /p/id1.xv 1.5 0.25"
