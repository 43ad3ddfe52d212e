"""Parity at the FULL size of BASELINE.json configs[1] (not part of the test suite: the oracle needs minutes of CPU).

PGD-20, L-inf eps 0.002, step 0.0004, cross-entropy, untargeted, CSI-E, 64 utterances x 3 s @ 16 kHz, dither off -- the
exact workload bench.py times -- through the HIP path and through the oracle (vectorised CPU restatement, pinned against
the reference run) on the same seeded inputs.  Prints what north_star asks for: success flags and
predicted speaker ids on the adversarial audio (must be equal) and the perturbation difference (stated tolerance: see
tests/test_gpu_xv.py).  The oracle is used as the checker only.

    python tests/tools/full_config_parity.py [n_utterances=64] [steps=20]
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
from speakerguard_amd.attack.PGD import PGD  # noqa: E402
from speakerguard_amd.model.xv_plda import xv_plda  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
EPS, STEP, T = 0.002, 0.0004, 48000
dev = torch.device("cuda:0")
w = synth.make_xv_weights(seed=0, D=200, n_spk=10)
x = torch.from_numpy(synth.make_waveforms(B, T, seed=1234))
y = torch.arange(B) % 10

hip = xv_plda.from_weights(w, device=dev, dither=0.0)
t0 = time.perf_counter()
adv, succ = PGD(hip, task="CSI", epsilon=EPS, step_size=STEP, max_iter=K, batch_size=B, verbose=0).attack(x.to(dev), y.to(dev))
torch.cuda.synchronize()
t_hip = time.perf_counter() - t0

ora = XvPlda(w, faithful=False, freeze=True)
t0 = time.perf_counter()
oadv, osucc = oatk.PGD(ora, task="CSI", epsilon=EPS, step_size=STEP, max_iter=K, batch_size=B).attack(x, y)
t_ora = time.perf_counter() - t0

with torch.no_grad():
    odec_on_oadv, osc = ora.make_decision(oadv)
    odec_on_hadv, osc_on_hadv = ora.make_decision(adv.cpu())
hdec_on_hadv, hsc = hip.make_decision(adv)
hdec_on_oadv, _ = hip.make_decision(oadv.to(dev))
diff = (adv.cpu() - oadv).abs()
print("configs[1] at full size: PGD-%d, %d utterances x 3 s, eps %.4g, step %.4g (HIP %.2f s incl. first-call set-up, oracle on %d CPU threads %.1f s)"
      % (K, B, EPS, STEP, t_hip, torch.get_num_threads(), t_ora))
print("  success flags          : HIP %d/%d, oracle %d/%d, equal per utterance: %s"
      % (sum(succ), B, sum(osucc), B, [bool(a) for a in succ] == [bool(a) for a in osucc]))
print("  predicted speaker ids  : HIP ids on HIP audio == oracle ids on oracle audio: %s; both models agree on the HIP audio: %s, on the oracle audio: %s"
      % (hdec_on_hadv.cpu().tolist() == odec_on_oadv.tolist(), hdec_on_hadv.cpu().tolist() == odec_on_hadv.tolist(),
         hdec_on_oadv.cpu().tolist() == odec_on_oadv.tolist()))
print("  perturbation           : max |x_adv - x| HIP %.6f oracle %.6f (eps %.4g); samples that differ between the two %.2f %%, max |diff| %.6f (<= 2 eps: %s)"
      % ((adv.cpu() - x).abs().max().item(), (oadv - x).abs().max().item(), EPS, 100 * float((diff > 1e-7).float().mean()),
         diff.max().item(), diff.max().item() <= 2 * EPS + 1e-6))
print("  scores, SAME audio     : max |HIP - oracle| %.4f on |score| up to %.1f (both models on the HIP path's adversarial audio)"
      % ((hsc.cpu() - osc_on_hadv).abs().max().item(), osc_on_hadv.abs().max().item()))
print("  scores, own audio      : max |HIP - oracle| %.2f -- the two trajectories end at different points of the eps-ball (sign() feeds "
      "round-off on near-zero gradient entries back, DESIGN.md section 2), both fool the model on every utterance"
      % (hsc.cpu() - osc).abs().max().item())
