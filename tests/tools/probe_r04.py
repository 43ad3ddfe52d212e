"""Which parameters of the round-4 full-size tests leave BOTH outcomes (HIP path alone; the oracle checks them in the tests)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speakerguard_amd import synth
from speakerguard_amd.attack.CW2 import CW2
from speakerguard_amd.model.xv_plda import xv_plda
DEV = torch.device("cuda:0")
xvw = synth.make_xv_weights(seed=0, D=200, n_spk=10)
w = dict(xvw); w["enroll"] = xvw["enroll"][:1].copy()
x = torch.from_numpy(synth.make_waveforms(32, 48000, seed=35)).to(DEV)
probe = xv_plda.from_weights(w, device=DEV, dither=0.0)
clean = probe.make_decision(x)[1][:, 0].cpu()
y = torch.zeros(32, dtype=torch.long, device=DEV)
for lr in (2e-3, 5e-4, 2e-4):
    for iters in (10,):
        hm = xv_plda.from_weights(w, threshold=1e9, device=DEV, dither=0.0)
        kw = dict(task="SV", targeted=True, confidence=0.0, initial_const=1e-2, binary_search_steps=1, max_iter=iters,
                  stop_early=True, stop_early_iter=5, lr=lr, batch_size=32)
        adv, succ = CW2(hm, verbose=0, **kw).attack(x, y)
        print("lr %g iters %d unreachable threshold: final scores" % (lr, iters), [round(float(v), 1) for v in hm.make_decision(adv)[1][:, 0].cpu()])
    for thr in (0.0, 20.0, 40.0, 55.0):
        hm = xv_plda.from_weights(w, threshold=thr, device=DEV, dither=0.0)
        adv, succ = CW2(hm, verbose=0, **dict(kw, lr=lr)).attack(x, y)
        print("  lr %g thr %.0f: %d/32" % (lr, thr, sum(succ)))
