"""One fused PGD run at a given batch, for `rocprofv3 --kernel-trace --stats -- python3 tools/step_profile.py B [K]`."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import synth
from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
from speakerguard_amd.model.xv_plda import xv_plda
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
m = xv_plda.from_weights(synth.make_xv_weights(), device=dev, dither=0.0)
spec = SEC4SR_CrossEntropy()
x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=1)).to(dev)
y = (torch.arange(B) % 10).to(dev)
lo, hi = torch.clamp(x - 0.002, min=-1), torch.clamp(x + 0.002, max=1)
m.pgd_run(x, y, lo, hi, spec, 0.0004, 2, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
m.pgd_run(x, y, lo, hi, spec, 0.0004, K, 1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("batch %d: %.3f ms per step, %.0f utterance-steps/s" % (B, 1e3 * dt / K, B * K / dt))
