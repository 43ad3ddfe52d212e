"""Parity at the size of BASELINE.json configs[3] on one GPU's shard (not part of the test suite):
PGD-10 against the FeCo-defended AudioNet (FeCo at the log-mel level, cl_r 0.5, deterministic clustering), 64 utterances x
3 s, through the ONE device loop (sg_an_pgd_run_feco) and through the oracle: reference-pinned AudioNet restatement +
oracle.feco (its OWN cluster ids from its own features, contract restatement) + torch autograd, driven by the oracle's PGD.
The oracle is the checker only.

    python tests/tools/config3_parity.py [n_utterances=64] [steps=10]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import attacks as oatk  # noqa: E402
from oracle import feco  # noqa: E402
from oracle.audionet import AudioNet  # noqa: E402
from speakerguard_amd import synth  # noqa: E402
from speakerguard_amd.attack.PGD import PGD  # noqa: E402
from speakerguard_amd.defense.feature_level import FeCoDefense  # noqa: E402
from speakerguard_amd.model.audionet_csine import audionet_csine  # noqa: E402
from speakerguard_amd.model.defended_model import defended_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
EPS, STEP, RATIO = 0.002, 0.0004, 0.5
dev = torch.device("cuda:0")
sd = synth.make_audionet_state_dict(seed=0, num_class=251)
hip, ora = audionet_csine.from_weights(sd, device=dev), AudioNet(sd)


class OracleDefended:
    """defended_model(AudioNet, [(1, FeCo)]) on the oracle side: differentiable through the cluster means, ids from the
    contract restatement on the oracle's own features."""
    threshold = -np.inf
    same = []

    def make_decision(self, x):
        feats = ora.compute_feat(x, flag=1)
        k = int(feats.shape[1] * RATIO)
        comp = []
        for b in range(feats.shape[0]):
            ids = feco.kmeans_ids(feats[b].detach().numpy(), k)
            comp.append(feco.compress_from_ids(feats[b], ids, k, force=True))
        return ora.make_decision(torch.stack(comp), flag=1)


x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=3))
dm = defended_model(hip, defense=[(1, FeCoDefense(RATIO))])
y = dm.make_decision(x.to(dev))[0].cpu()
om = OracleDefended()
with torch.no_grad():
    oy = om.make_decision(x)[0]
t0 = time.perf_counter()
atk = PGD(dm, task="CSI", epsilon=EPS, step_size=STEP, max_iter=K, batch_size=B, verbose=0)
adv, succ = atk.attack(x.to(dev), y.to(dev))
torch.cuda.synchronize()
t_h = time.perf_counter() - t0
t0 = time.perf_counter()
oadv, osucc = oatk.PGD(om, task="CSI", epsilon=EPS, step_size=STEP, max_iter=K, batch_size=B).attack(x, y)
t_o = time.perf_counter() - t0
hdec = dm.make_decision(adv)[0].cpu()
with torch.no_grad():
    odec = om.make_decision(oadv)[0]
    odec_h = om.make_decision(adv.cpu())[0]
diff = (adv.cpu() - oadv).abs()
print("configs[3], one GPU's shard: PGD-%d vs FeCo-defended AudioNet, %d utterances x 3 s (device loop: %s; HIP %.2f s, oracle %.0f s)"
      % (K, B, atk._fused_feco(B) is not None, t_h, t_o))
print("  clean decisions equal: %s" % (y.tolist() == oy.tolist()))
print("  success flags: HIP %d/%d, oracle %d/%d, equal per utterance: %d/%d; decisions on own audio equal: %d/%d; the oracle's decisions on "
      "the HIP audio equal the HIP model's: %d/%d" % (sum(succ), B, sum(osucc), B, sum(bool(a) == bool(b) for a, b in zip(succ, osucc)), B,
                                                    int((hdec == odec).sum()), B, int((hdec == odec_h).sum()), B))
print("  perturbation: samples that differ %.2f %%, max |diff| %.6f (<= 2 eps: %s)"
      % (100 * float((diff > 1e-7).float().mean()), diff.max().item(), diff.max().item() <= 2 * EPS + 1e-6))
