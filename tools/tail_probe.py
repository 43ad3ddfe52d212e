import os, sys
sys.path.insert(0, "/root/repo")
import torch
from speakerguard_amd import synth
from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
from speakerguard_amd.model.xv_plda import xv_plda
dev = torch.device("cuda:0")
m = xv_plda.from_weights(synth.make_xv_weights(), device=dev, dither=0.0)
for B in (64, 8):
    x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=1)).to(dev)
    y = (torch.arange(B) % 10).to(dev)
    sys.stderr.write("B=%d\n" % B)
    for _ in range(4):
        m.loss_grad(x, y, SEC4SR_CrossEntropy())
    torch.cuda.synchronize()
