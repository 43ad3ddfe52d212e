"""PGD-20 on AudioNet: separate launches / head inside the backward / one launch, same process, alternating."""
import os, sys, time
os.environ.setdefault("SG_TUNE", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import synth
from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
from speakerguard_amd.model.audionet_csine import audionet_csine
dev = torch.device("cuda:0")
m = audionet_csine.from_weights(synth.make_audionet_state_dict(seed=0, num_class=251), device=dev)
spec = SEC4SR_CrossEntropy()
for B in [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512]:
    x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=5)).to(dev)
    y = m.make_decision(x)[0]
    lo, hi = torch.clamp(x - 0.002, min=-1), torch.clamp(x + 0.002, max=1)
    res = {}
    for rep in range(3):
        for name, head, one in (("separate", "0", "0"), ("head in bwd", "1", "0"), ("one launch", "1", "1")):
            os.environ["SG_AN_HEAD"], os.environ["SG_AN_ONE"] = head, one
            m.pgd_run(x, y, lo, hi, spec, 0.0004, 3, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.pgd_run(x, y, lo, hi, spec, 0.0004, 20, 1)
            torch.cuda.synchronize()
            res.setdefault(name, []).append((time.perf_counter() - t0) / 20 * 1e3)
    print("B=%3d " % B + "  ".join("%s %.3f" % (k, min(v)) for k, v in res.items()) + "  ms per PGD step", flush=True)
