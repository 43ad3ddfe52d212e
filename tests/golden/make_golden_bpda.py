"""tests/golden/bpda.npz: the reference's own BPDA wrapper (adaptive_attack/BPDA.py:7-65) and its BPDA-wrapped
quantisation defense (defense/time_domain.py:10-48 QT, BDR), executed unmodified in the build container.
Forward outputs and the gradients torch.autograd sends through them for given upstream gradients."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
np.infty = np.inf
import torch  # noqa: E402
from adaptive_attack.BPDA import BPDA  # noqa: E402
from defense.time_domain import BDR, QT  # noqa: E402

rs = np.random.RandomState(5)
x = (0.3 * rs.randn(3, 1, 257)).astype(np.float32)
g = rs.randn(3, 1, 257).astype(np.float32)
out = {"x": x, "g": g}


def run(name, f):
    xin = torch.from_numpy(x).clone().requires_grad_(True)
    y = f(xin)
    y.backward(torch.from_numpy(g))
    out[name + "_out"] = y.detach().numpy()
    out[name + "_grad"] = xin.grad.numpy()


run("qt128", lambda t: QT(t, 128))                       # straight-through: gradient == g
run("bdr8", lambda t: BDR(t, 8))
ori = lambda t, s: torch.round(t * s) / s                 # noqa: E731  a generic non-differentiable transform
sub = lambda t, s: t + 0.1 * torch.sin(t * s)             # noqa: E731  and a differentiable substitute
run("generic", lambda t: BPDA(ori, sub)(t, 7.0))
np.savez_compressed(os.path.join(HERE, "bpda.npz"), meta=json.dumps({
    "generator": "tests/golden/make_golden_bpda.py", "reference": "adaptive_attack/BPDA.py, defense/time_domain.py (unmodified)",
    "accommodations": ["numpy.infty alias"], "torch": torch.__version__}), **out)
print({k: v.shape for k, v in out.items()})
