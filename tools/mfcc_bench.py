"""Times the MFCC forward / backward kernels alone (B=64 x 3 s) via torch events around the C-ABI calls."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import synth
from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
from speakerguard_amd.model.xv_plda import xv_plda
dev = torch.device("cuda:0")
m = xv_plda.from_weights(synth.make_xv_weights(), device=dev, dither=0.0)
x = torch.from_numpy(synth.make_waveforms(64, 48000, seed=1)).to(dev)
y = (torch.arange(64) % 10).to(dev)
for name, fn in (("mfcc fwd (compute_feat flag=1)", lambda: m.compute_feat(x, 1)),
                 ("full loss_grad", lambda: m.loss_grad(x, y, SEC4SR_CrossEntropy())),
                 ("forward only", lambda: m.make_decision(x))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-34s %.3f ms" % (name, e0.elapsed_time(e1) / 10))
