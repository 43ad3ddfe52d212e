"""FAKEBOB / NES on xv_plda OSI, 8 examples x 51 queries per iteration (BASELINE configs[4]) for rocprofv3."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import synth
from speakerguard_amd.attack.FAKEBOB import FAKEBOB
from speakerguard_amd.model.xv_plda import xv_plda
dev = torch.device("cuda:0")
osi = xv_plda.from_weights(synth.make_xv_weights(), threshold=-10.0, device=dev, dither=0.0)
xq = torch.from_numpy(synth.make_waveforms(8, 48000, seed=4)).to(dev)
yq = (osi.make_decision(xq)[0].clamp(min=0) + 3) % 10
fb = FAKEBOB(osi, threshold=-10.0, task="OSI", targeted=True, epsilon=0.002, max_iter=5, samples_per_draw=50, samples_per_draw_batch_size=50,
             stop_early=False, batch_size=8, verbose=0)
fb.attack(xq, yq)
torch.cuda.synchronize(); t0 = time.perf_counter()
fb.attack(xq, yq)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("FAKEBOB: %.2f ms per iteration of 408 queries (%.0f queries/s)" % (1e3 * dt / 6, 8 * 51 * 6 / dt))
