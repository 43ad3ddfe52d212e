"""Per-stage time of one gradient pass (the library's own HIP-event stage trace, median over passes).

  python tools/stage_times.py --batch 64 [--only mfcc]     # SG_MFCC_ABLATE / SG_ABLATE are read by the library as usual
"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import synth
from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
from speakerguard_amd.model.xv_plda import xv_plda

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--passes", type=int, default=12)
ap.add_argument("--only", default="")
a = ap.parse_args()
dev = torch.device("cuda:0")
m = xv_plda.from_weights(synth.make_xv_weights(), device=dev, dither=0.0)
x = torch.from_numpy(synth.make_waveforms(a.batch, 48000, seed=1)).to(dev)
y = (torch.arange(a.batch) % 10).to(dev)
loss = SEC4SR_CrossEntropy()
for _ in range(3):
    m.loss_grad(x, y, loss)
torch.cuda.synchronize()


def run():
    for _ in range(a.passes):
        m.loss_grad(x, y, loss)
    torch.cuda.synchronize()


recs = m.trace_stages(run)
per, order = {}, []
for name, ms in recs:
    if name not in per:
        per[name] = []
        order.append(name)
    per[name].append(ms)
tot = 0.0
for name in order:
    med = 1e3 * statistics.median(per[name])
    tot += med
    if a.only in name:
        print("  %-14s %8.1f us  (n=%d)" % (name, med, len(per[name])))
print("  sum of medians %.1f us" % tot)
