"""xv_plda fused PGD throughput vs batch per GPU (the BASELINE metric is quoted at 64; informational)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import synth
from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
from speakerguard_amd.model.xv_plda import xv_plda
dev = torch.device("cuda:0")
m = xv_plda.from_weights(synth.make_xv_weights(), device=dev, dither=0.0)
spec = SEC4SR_CrossEntropy()
for B in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
    x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=1)).to(dev)
    y = (torch.arange(B) % 10).to(dev)
    lo, hi = torch.clamp(x - 0.002, min=-1), torch.clamp(x + 0.002, max=1)
    # PGD-20 (20 steps + the final forward-only pass, as the metric counts them); the warm-up attack is long enough for
    # the chip to reach its clock at the smallest batches (>= ~15 ms of work), then the median of three timed attacks
    K = 20
    m.pgd_run(x, y, lo, hi, spec, 0.0004, max(K, int(40 / max(B, 1)) * K), 1)
    torch.cuda.synchronize()
    dts = []
    for _ in range(3):
        t0 = time.perf_counter()
        m.pgd_run(x, y, lo, hi, spec, 0.0004, K, 1)
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[1]
    print("batch %3d: %7.2f ms per step  %8.0f utterance-steps/s  %6.1f model TFLOP/s  (torch-visible memory %.0f MB)" % (
        B, 1e3 * dt / K, B * K / dt, B * K / dt * 4.70e9 / 1e12, torch.cuda.memory_allocated() / 1e6))
