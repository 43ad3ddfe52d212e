"""The AudioNet CNN alone (feature-level pass: no log-mel front-end): in-loop launch times from the library's stage trace,
fused kernels (default) against the per-layer sequence (SG_AN_FUSED=0).  python tools/audionet_cnn_bench.py [B ...]"""
import os, sys
os.environ.setdefault("SG_TUNE", "1")  # the knobs below count only behind this gate, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import synth
from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
from speakerguard_amd.model.audionet_csine import audionet_csine
dev = torch.device("cuda:0")
m = audionet_csine.from_weights(synth.make_audionet_state_dict(seed=0, num_class=251), device=dev)
ce = SEC4SR_CrossEntropy()
PEAK = 157.3
for B in [int(a) for a in sys.argv[1:]] or [64, 512]:
    x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=5)).to(dev)
    feats = m.compute_feat(x)
    y = m.make_decision(x)[0]
    for fused in ("1", "0"):
        os.environ["SG_AN_FUSED"] = fused
        for _ in range(3):
            m.loss_grad(feats, y, ce, flag=1)
        recs = m.trace_stages(lambda: [m.loss_grad(feats, y, ce, flag=1) for _ in range(10)], max_records=2048)
        by = {}
        for t, ms in recs:
            by.setdefault(t, []).append(ms)
        tot = sum(sum(v) for k, v in by.items() if k != "an_tail") / 10
        print("B=%d %s: CNN launches %.1f us per pass (fwd + bwd, tail excluded) = %.2f of the f32-MFMA peak; " % (
            B, "fused" if fused == "1" else "per-layer", 1e3 * tot, B * 2 * 0.053e9 / (tot * 1e-3) / 1e12 / PEAK) +
            ", ".join("%s %.1f" % (k, 1e3 * statistics.mean(v)) for k, v in by.items() if fused == "1" or k == "an_tail"))
