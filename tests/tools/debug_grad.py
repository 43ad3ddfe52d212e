# Debug aid (imports the oracle as the checker, hence it lives under tests/, not tools/).
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speakerguard_amd import synth
from speakerguard_amd.model.xv_plda import xv_plda
from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
from oracle.xv_plda import XvPlda
from oracle import attacks as oatk
dev = torch.device("cuda:0")
w = synth.make_xv_weights()
hm = xv_plda.from_weights(w, device=dev, dither=0.0)
om = XvPlda(w)
x = torch.from_numpy(synth.make_waveforms(3, 48000, seed=22))
with torch.no_grad():
    y = om.make_decision(x)[0]
xin = x.clone().requires_grad_(True)
_, sc = om.make_decision(xin)
oatk.cross_entropy_loss(sc, y).backward(torch.ones(3))
want = xin.grad.numpy()[:, 0]
_, _, _, g = hm.loss_grad(x.to(dev), y.to(dev), SEC4SR_CrossEntropy())
got = g.cpu().numpy()[:, 0]
m64 = XvPlda(w).double()
x64 = x.double().requires_grad_(True)
_, s64 = m64.make_decision(x64)
torch.nn.functional.cross_entropy(s64, y, reduction="none").backward(torch.ones(3, dtype=torch.float64))
g64 = x64.grad.numpy()[:, 0]
for b in range(3):
    mis = np.nonzero(np.sign(got[b]) != np.sign(g64[b]))[0]
    print("utt", b, "mismatches", len(mis), "max|g64|", np.abs(g64[b]).max(), "oracle32 mismatches", int((np.sign(want[b]) != np.sign(g64[b])).sum()))
    if len(mis) == 0:
        continue
    print("  first idx", mis[:20])
    print("  idx mod 160 histogram top", np.bincount(mis % 160, minlength=160).argsort()[-5:], np.sort(np.bincount(mis % 160, minlength=160))[-5:])
    print("  |g64| at mismatches: median %.3e max %.3e ; |got-g64| median %.3e" % (np.median(np.abs(g64[b][mis])), np.abs(g64[b][mis]).max(), np.median(np.abs(got[b][mis]-g64[b][mis]))))
    print("  region histogram (per 4800):", np.histogram(mis, bins=10, range=(0, 48000))[0])
    err = np.abs(got[b] - g64[b])
    print("  overall |err| median %.3e p99 %.3e ; |g64| median %.3e" % (np.median(err), np.quantile(err, 0.99), np.median(np.abs(g64[b]))))
    e32 = np.abs(want[b] - g64[b])
    print("  oracle32 |err| median %.3e p99 %.3e" % (np.median(e32), np.quantile(e32, 0.99)))
np.savez("gpurun_out/grad_debug.npz", got=got, want=want, g64=g64)
