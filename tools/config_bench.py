"""Throughput of the other BASELINE.json configurations on one GPU (informational: the metric is configs[1], bench.py).

    configs[2]  CW2 L2 targeted on xv_plda SV task, batch 32, Adam inner optimiser
    configs[3]  PGD + EOT vs FeCo-defended AudioNet, 64 utterances (= one GPU's shard of the batch of 512)
    configs[4]  FAKEBOB / NES on xv_plda OSI, samples_per_draw 50: queries per second
"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import synth
from speakerguard_amd.attack.CW2 import CW2
from speakerguard_amd.attack.FAKEBOB import FAKEBOB
from speakerguard_amd.attack.PGD import PGD
from speakerguard_amd.defense.feature_level import FeCoDefense
from speakerguard_amd.model.audionet_csine import audionet_csine
from speakerguard_amd.model.defended_model import defended_model
from speakerguard_amd.model.xv_plda import xv_plda

dev = torch.device("cuda:0")
w = synth.make_xv_weights()


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


# ---- configs[0]: FGSM 1-step on AudioNet CSI-NE, a single 3 s utterance (the reference's CPU-runnable plumbing case)
from speakerguard_amd.attack.FGSM import FGSM
an0 = audionet_csine.from_weights(synth.make_audionet_state_dict(seed=0, num_class=251), device=dev)
x0 = torch.from_numpy(synth.make_waveforms(1, 48000, seed=1)).to(dev)
y0 = an0.make_decision(x0)[0]
fgsm = FGSM(an0, task="CSI", epsilon=0.002, batch_size=1, verbose=0)
fgsm.attack(x0, y0)
(_, succ0), dt = timed(lambda: [fgsm.attack(x0, y0) for _ in range(20)][-1])
print("configs[0] FGSM (1 step + final pass) on AudioNet, ONE 3 s utterance: %.2f ms per attack, success %s" % (1e3 * dt / 20, succ0))

# ---- configs[2]: CW2, SV (one enrolled speaker, finite threshold), batch 32
w1 = dict(w); w1["enroll"] = w["enroll"][:1]
sv = xv_plda.from_weights(w1, threshold=-10.0, device=dev, dither=0.0)
x = torch.from_numpy(synth.make_waveforms(32, 48000, seed=2)).to(dev)
y = torch.zeros(32, dtype=torch.int64, device=dev)
iters, steps = 30, 2
atk = CW2(sv, task="SV", targeted=True, initial_const=1e-3, binary_search_steps=steps, max_iter=iters, stop_early=False,
          lr=1e-2, batch_size=32, verbose=0)
atk.attack(x[:32], y)  # warm
(_, succ), dt = timed(lambda: atk.attack(x, y))
n_it = steps * (iters + 1)
print("configs[2] CW2 targeted SV, batch 32 x 3 s: %d iterations in %.2f s -> %.1f iterations/s (%.0f utterance-iterations/s), success %d/32"
      % (n_it, dt, n_it / dt, 32 * n_it / dt, sum(succ)))

# ---- configs[3]: PGD + EOT vs FeCo-defended AudioNet, 64 utterances per GPU
an = audionet_csine.from_weights(synth.make_audionet_state_dict(seed=0, num_class=251), device=dev)
xa = torch.from_numpy(synth.make_waveforms(64, 48000, seed=3)).to(dev)
K = 20


def feco_run(label, init, eot, fused, passes_per_step):
    dm = defended_model(an, defense=[(1, FeCoDefense(0.5, init=init, seed=1))])
    ya = defended_model(an, defense=[(1, FeCoDefense(0.5))]).make_decision(xa)[0]
    pgd = PGD(dm, epsilon=0.002, step_size=0.0004, max_iter=K, batch_size=64, EOT_size=eot, EOT_batch_size=1, verbose=0)
    pgd.fuse_defended = fused
    pgd.attack(xa, ya)
    (_, succ), dt = timed(lambda: pgd.attack(xa, ya))
    n_pass = passes_per_step * K + 1
    print("configs[3] PGD-%d + EOT %d vs FeCo-defended AudioNet (%s), batch 64 x 3 s: %.2f ms per step (%.2f ms per model pass), "
          "%.0f utterance-passes/s, success %d/64" % (K, eot, label, 1e3 * dt / K, 1e3 * dt / n_pass, 64 * n_pass / dt, sum(succ)))


# the randomised defense (fresh random initial frames per pass): EOT repeats are distinct passes
feco_run("random-init k-means, host-chained gradient", "random", 2, False, 2)
feco_run("random-init k-means, ONE device loop sg_an_pgd_run_feco", "random", 2, True, 2)
# the deterministic defense: its EOT repeats coincide -- the host loop still runs them, the device loop runs one
feco_run("evenly started k-means, host-chained gradient", "even", 2, False, 2)
feco_run("evenly started k-means, ONE device loop", "even", 2, True, 1)
ya = an.make_decision(xa)[0]
plain = PGD(an, epsilon=0.002, step_size=0.0004, max_iter=K, batch_size=64, verbose=0)
plain.attack(xa, ya)
(_, _), dt0 = timed(lambda: plain.attack(xa, ya))
print("           undefended AudioNet, fused loop: %.2f ms per step" % (1e3 * dt0 / K))

# ---- configs[4]: FAKEBOB / NES, OSI, samples_per_draw 50
osi = xv_plda.from_weights(w, threshold=-10.0, device=dev, dither=0.0)
xq = torch.from_numpy(synth.make_waveforms(8, 48000, seed=4)).to(dev)
yq = (osi.make_decision(xq)[0].clamp(min=0) + 3) % 10   # targeted at another speaker: no example finishes in 5 iterations
iters = 5
fb = FAKEBOB(osi, threshold=-10.0, task="OSI", targeted=True, epsilon=0.002, max_iter=iters, samples_per_draw=50, samples_per_draw_batch_size=50,
             stop_early=False, batch_size=8, verbose=0)
fb.attack(xq[:8], yq)
(_, succ), dt = timed(lambda: fb.attack(xq, yq))
q = 8 * 51 * (iters + 1)
print("configs[4] FAKEBOB (NES 50 + 1 queries per example and iteration), batch 8 x 3 s: %d queries in %.2f s -> %.0f queries/s, success %s" % (q, dt, q / dt, succ))
