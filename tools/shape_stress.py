"""Runs the fused PGD loop and the step API over unusual (B, T) shapes, in one process (workspace re-allocation),
and checks fused == stepwise bit for bit plus basic invariants.  Not a parity test: a crash / hang / NaN detector."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import synth
from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy, SEC4SR_MarginLoss
from speakerguard_amd.model.xv_plda import xv_plda
dev = torch.device("cuda:0")
m = xv_plda.from_weights(synth.make_xv_weights(), device=dev, dither=0.0)
eps, step = 0.002, 0.0004
for B, T in ((1, 16001), (130, 8000), (7, 160000), (64, 48000), (3, 5600), (257, 16000), (2, 48000), (64, 48000)):
    x = torch.from_numpy(synth.make_waveforms(B, T, seed=B + T)).to(dev)
    y = m.make_decision(x)[0]
    lo, hi = torch.clamp(x - eps, min=-1), torch.clamp(x + eps, max=1)
    spec = SEC4SR_CrossEntropy()
    out = m.pgd_run(x, y, lo, hi, spec, step, 2, 1)
    xs = x.clone()
    for _ in range(2):
        _, _, _, g = m.loss_grad(xs, y, spec)
        m.pgd_update(xs, g, lo, hi, step, 1)
    ok = torch.equal(out[0], xs)
    fin = bool(torch.isfinite(out[0]).all()) and bool(torch.isfinite(out[3]).all())
    print("B=%3d T=%6d  fused==stepwise %s  finite %s  max|dx| %.6f  successes %d" % (
        B, T, ok, fin, (out[0] - x).abs().max().item(), int(out[1].sum())))
    assert ok and fin and (out[0] - x).abs().max().item() <= eps + 1e-7
print("mem allocated by torch: %.1f MB" % (torch.cuda.memory_allocated() / 1e6))
