"""Parity at the sizes of BASELINE.json configs[2] and configs[4] (not part of the test suite: minutes of oracle CPU time).

  configs[2]  CW2 L2 targeted on the xv_plda SV task, batch 32 x 3 s, Adam inner optimiser
  configs[4]  FAKEBOB / NES on xv_plda OSI, samples_per_draw 50 (51 queries per example and iteration), 8 examples x 3 s

through the HIP path and through the oracle (the checker) on the same seeded inputs; the NES noise of both sides comes
from the same seeded CPU generator.  Prints success flags / decisions (must be equal) and the difference of the audio.

    python tests/tools/other_configs_parity.py
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import attacks as oatk  # noqa: E402
from oracle.xv_plda import XvPlda  # noqa: E402
from speakerguard_amd import synth  # noqa: E402
from speakerguard_amd.attack.CW2 import CW2  # noqa: E402
from speakerguard_amd.attack.FAKEBOB import FAKEBOB  # noqa: E402
from speakerguard_amd.model.xv_plda import xv_plda  # noqa: E402

dev = torch.device("cuda:0")
w = synth.make_xv_weights(seed=0, D=200, n_spk=10)

# ---- configs[2]: CW2, SV (one enrolled speaker), every utterance starts rejected, targeted = get it accepted
w1 = dict(w)
w1["enroll"] = w["enroll"][:1].copy()
x = torch.from_numpy(synth.make_waveforms(32, 48000, seed=35))
probe = xv_plda.from_weights(w1, threshold=None, device=dev, dither=0.0)
thr = float(probe.make_decision(x.to(dev))[1][:, 0].max()) + 2.0
om, hm = XvPlda(w1, threshold=thr), xv_plda.from_weights(w1, threshold=thr, device=dev, dither=0.0)
y = torch.zeros(32, dtype=torch.long)
kw = dict(task="SV", targeted=True, confidence=0.0, initial_const=1e-2, binary_search_steps=2, max_iter=20, stop_early=True,
          stop_early_iter=10, lr=2e-3, batch_size=32)
t0 = time.perf_counter()
adv, succ = CW2(hm, verbose=0, **kw).attack(x.to(dev), y.to(dev))
torch.cuda.synchronize()
t_h = time.perf_counter() - t0
t0 = time.perf_counter()
oadv, osucc = oatk.CW2(om, **kw).attack(x.clone(), y)
t_o = time.perf_counter() - t0
d = (adv.cpu() - oadv).abs().numpy()
with torch.no_grad():
    odec = om.make_decision(oadv)[0].tolist()
hdec = hm.make_decision(adv)[0].cpu().tolist()
l2h, l2o = (adv.cpu() - x).flatten(1).norm(dim=1), (oadv - x).flatten(1).norm(dim=1)
print("configs[2] at full size: CW2 targeted SV, 32 utterances x 3 s, 2 search steps x 20 iterations (HIP %.2f s, oracle on %d CPU threads %.0f s)"
      % (t_h, torch.get_num_threads(), t_o))
print("  success flags: HIP %d/32, oracle %d/32, equal per utterance: %s; decisions on own audio equal: %s"
      % (sum(succ), sum(osucc), list(map(bool, succ)) == list(map(bool, osucc)), hdec == odec))
print("  audio: max |x_adv - oracle's| %.3e, samples differing by > 2e-4: %.4f %%; L2 of the perturbation HIP %.4f oracle %.4f (mean)"
      % (d.max(), 100 * float((d > 2e-4).mean()), float(l2h.mean()), float(l2o.mean())))

# ---- configs[4]: FAKEBOB / NES, OSI with a finite threshold, targeted at another speaker
oo, ho = XvPlda(w, threshold=-10.0), xv_plda.from_weights(w, threshold=-10.0, device=dev, dither=0.0)
xq = torch.from_numpy(synth.make_waveforms(8, 48000, seed=4))
yq = (ho.make_decision(xq.to(dev))[0].cpu().clamp(min=0) + 3) % 10
kq = dict(threshold=-10.0, task="OSI", targeted=True, epsilon=0.002, max_iter=4, samples_per_draw=50, samples_per_draw_batch_size=50,
          batch_size=8, stop_early=False)
g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
t0 = time.perf_counter()
qadv, qsucc = FAKEBOB(ho, verbose=0, noise_fn=lambda shape: torch.randn(shape, generator=g1), **kq).attack(xq.to(dev), yq.to(dev))
torch.cuda.synchronize()
t_h = time.perf_counter() - t0
t0 = time.perf_counter()
oqadv, oqsucc = oatk.FAKEBOB(oo, noise_fn=lambda shape: torch.randn(shape, generator=g2), **kq).attack(xq.clone(), yq)
t_o = time.perf_counter() - t0
dq = (qadv.cpu() - oqadv).abs()
with torch.no_grad():
    oqdec = oo.make_decision(oqadv)[0].tolist()
print("configs[4] at full size: FAKEBOB (NES 50 + 1 queries per example and iteration), 8 utterances x 3 s, 5 iterations = %d queries "
      "(HIP %.2f s, oracle %.0f s)" % (8 * 51 * 5, t_h, t_o))
print("  success flags equal: %s (HIP %s); decisions on own audio equal: %s; samples differing %.2f %%, max |diff| %.6f (<= 2 eps: %s)"
      % (list(map(bool, qsucc)) == list(map(bool, oqsucc)), list(map(bool, qsucc)), ho.make_decision(qadv)[0].cpu().tolist() == oqdec,
         100 * float((dq > 1e-7).float().mean()), dq.max().item(), dq.max().item() <= 2 * 0.002 + 1e-6))
