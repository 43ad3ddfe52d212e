"""Scratch probe (HIP side only, seconds): which attack parameters leave SOME utterances un-fooled, so that the parity tests
compare both outcomes of the success predicate instead of all-False / all-True flag lists."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from speakerguard_amd import synth
from speakerguard_amd.attack.PGD import PGD
from speakerguard_amd.attack.FAKEBOB import FAKEBOB
from speakerguard_amd.model.xv_plda import xv_plda
dev = torch.device("cuda:0")
w = synth.make_xv_weights(seed=0, D=200, n_spk=10)
hip = xv_plda.from_weights(w, device=dev, dither=0.0)
x = torch.from_numpy(synth.make_waveforms(64, 48000, seed=1234)).to(dev)
y = hip.make_decision(x)[0]
for eps, step, k in ((0.0001, 0.00002, 5), (0.0002, 0.00004, 5), (0.0005, 0.0001, 5), (0.001, 0.0002, 5)):
    adv, succ = PGD(hip, task="CSI", epsilon=eps, step_size=step, max_iter=k, batch_size=64, verbose=0).attack(x, y)
    top = hip.make_decision(adv)[1].topk(2, 1)[0]
    m = (top[:, 0] - top[:, 1])
    print("xv PGD-%d eps %.5f y=clean decisions: success %d/64, margins of the final scores: min %.3f, 5 smallest %s" % (
        k, eps, sum(succ), float(m.min()), [round(float(v), 2) for v in m.sort()[0][:5]]))
for th in (0.0, -5.0):
    hm = xv_plda.from_weights(w, threshold=th, device=dev, dither=0.0)
    idx = [0, 1, 4, 5]
    xs = x[idx]
    d0, s0 = hm.make_decision(xs)
    yt = s0.argmax(1)
    print("threshold %.1f: decisions %s, top scores %s, targets %s" % (th, d0.tolist(), [round(float(v), 1) for v in s0.max(1)[0]], yt.tolist()))
    for iters in (4, 8):
        g = torch.Generator().manual_seed(5)
        atk = FAKEBOB(hm, task="OSI", targeted=True, threshold=th, epsilon=0.002, max_iter=iters, max_lr=0.001, min_lr=1e-6, samples_per_draw=50,
                      samples_per_draw_batch_size=50, sigma=0.001, stop_early=True, stop_early_iter=100, batch_size=4, verbose=0,
                      noise_fn=lambda shape: torch.randn(shape, generator=g))
        adv, succ = atk.attack(xs, yt)
        d1, s1 = hm.make_decision(adv)
        print("  FAKEBOB OSI targeted S=50 iters %d: success %s decisions %s top %s" % (iters, succ, d1.tolist(), [round(float(v), 1) for v in s1.max(1)[0]]))
