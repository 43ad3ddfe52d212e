"""CW2 targeted SV, batch 32 (BASELINE configs[2]) for rocprofv3; prints wall time per iteration."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import synth
from speakerguard_amd.attack.CW2 import CW2
from speakerguard_amd.model.xv_plda import xv_plda
dev = torch.device("cuda:0")
w = dict(synth.make_xv_weights()); w["enroll"] = w["enroll"][:1]
sv = xv_plda.from_weights(w, threshold=-10.0, device=dev, dither=0.0)
x = torch.from_numpy(synth.make_waveforms(32, 48000, seed=2)).to(dev)
y = torch.zeros(32, dtype=torch.int64, device=dev)
atk = CW2(sv, task="SV", targeted=True, initial_const=1e-3, binary_search_steps=2, max_iter=30, stop_early=False, lr=1e-2, batch_size=32, verbose=0)
atk.attack(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
atk.attack(x, y)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("CW2 batch 32: %.3f ms per iteration" % (1e3 * dt / 62))
