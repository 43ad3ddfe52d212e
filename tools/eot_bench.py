"""PGD with EOT over the front-end's random dither (the reference's default front-end, xv_plda.py:119): time per gradient
step for EOT_size repeats, which the device loop runs as ONE batch of EOT_size x B rows, against EOT_size times the
single-repeat step (what running the repeats one after the other costs)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import synth
from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
from speakerguard_amd.model.xv_plda import xv_plda
dev = torch.device("cuda:0")
m = xv_plda.from_weights(synth.make_xv_weights(), device=dev, dither=1.0, dither_seed=1)
spec, K = SEC4SR_CrossEntropy(), 10


def step_ms(B, eot):
    x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=5)).to(dev)
    y = (torch.arange(B) % 10).to(dev)
    lo, hi = torch.clamp(x - 0.002, min=-1), torch.clamp(x + 0.002, max=1)
    run = lambda: m.pgd_run(x, y, lo, hi, spec, 0.0004, K, 1, eot_size=eot, eot_batch_size=eot)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / K


for B in (1, 8, 64):
    one = step_ms(B, 1)
    for eot in (2, 4, 8):
        ms = step_ms(B, eot)
        print("batch %2d, EOT %d: %6.2f ms per step = %6.0f utterance-passes/s   (%d separate passes: %6.2f ms -> x%.2f)" % (
            B, eot, ms, 1e3 * B * eot / ms, eot, eot * one, eot * one / ms))
