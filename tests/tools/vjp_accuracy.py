"""Where does the fp32 noise of a score-VJP gradient come from?  HIP and the fp32 oracle against the fp64 model, for a
cross-entropy cotangent, a one-hot cotangent and a random +-1 cotangent, from the waveform (flag 0) and from CMVN
features (flag 2).  Debug aid (uses the oracle as the checker)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.xv_plda import XvPlda
from speakerguard_amd import synth
from speakerguard_amd.attack.utils import ScoreVJP
from speakerguard_amd.model.xv_plda import xv_plda
dev = torch.device("cuda:0")
w = synth.make_xv_weights()
hip = xv_plda.from_weights(w, device=dev, dither=0.0)
o32, o64 = XvPlda(w), XvPlda(w).double()
x = torch.from_numpy(synth.make_waveforms(3, 32000, seed=75))
rs = np.random.RandomState(12)
y0 = torch.zeros(3, dtype=torch.int64, device=dev)
with torch.no_grad():
    sc = o32.make_decision(x)[1]
p = torch.softmax(sc, 1)
cots = {"random normal": torch.from_numpy(rs.randn(3, 10).astype(np.float32)),
        "one-hot (speaker 2)": torch.nn.functional.one_hot(torch.tensor([2, 2, 2]), 10).float(),
        "softmax - onehot (CE-like)": p - torch.nn.functional.one_hot(sc.argmax(1), 10).float(),
        "all ones": torch.ones(3, 10)}
for flag in (0, 2):
    xin = x if flag == 0 else o32.compute_feat(x, flag=2).detach()
    for name, c in cots.items():
        g = hip.loss_grad(xin.to(dev), y0, ScoreVJP(c.to(dev)), flag=flag)[3].cpu().numpy()
        a = xin.clone().requires_grad_(True)
        (c * o32.make_decision(a, flag=flag)[1]).sum().backward()
        b = xin.double().requires_grad_(True)
        (c.double() * o64.make_decision(b, flag=flag)[1]).sum().backward()
        g64 = b.grad.numpy()
        rms = lambda v: float(np.sqrt(((v - g64) ** 2).mean() / (g64 ** 2).mean()))
        print("flag %d  %-28s rms err vs fp64: hip %.2e  oracle-fp32 %.2e   (|g64| rms %.3e)" % (
            flag, name, rms(g), rms(a.grad.numpy()), float(np.sqrt((g64 ** 2).mean()))))
