"""AudioNet CNN + head per pass: separate launches (SG_AN_HEAD=0) / head inside the backward / one launch; in-loop stage times."""
import os, sys, statistics
os.environ.setdefault("SG_TUNE", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import synth
from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
from speakerguard_amd.model.audionet_csine import audionet_csine
dev = torch.device("cuda:0")
m = audionet_csine.from_weights(synth.make_audionet_state_dict(seed=0, num_class=251), device=dev)
ce = SEC4SR_CrossEntropy()
for B in [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512]:
    x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=5)).to(dev)
    feats = m.compute_feat(x)
    y = m.make_decision(x)[0]
    for name, head, one in (("separate", "0", "0"), ("head in bwd", "1", "0"), ("one launch", "1", "1")):
        os.environ["SG_AN_HEAD"], os.environ["SG_AN_ONE"] = head, one
        for _ in range(3):
            m.loss_grad(feats, y, ce, flag=1)
        recs = m.trace_stages(lambda: [m.loss_grad(feats, y, ce, flag=1) for _ in range(10)], max_records=2048)
        by = {}
        for t, ms in recs:
            by.setdefault(t, []).append(ms)
        tot = sum(sum(v) for v in by.values()) / 10
        print("B=%d %-12s: %.1f us per pass; " % (B, name, 1e3 * tot) + ", ".join("%s %.1f" % (k, 1e3 * statistics.mean(v)) for k, v in by.items()), flush=True)
