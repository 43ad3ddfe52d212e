"""Where the per-attack fixed time of the headline goes (host side of PGD.attack on the fused loop).

    python tools/attack_overhead.py

T(K) = a + b K from attacks of K = 20 and 100 steps, then the pieces of `a` timed one by one (each bracketed by a
device synchronise, so the pieces overlap less than in the real call: an upper bound of their sum).
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import synth  # noqa: E402
from speakerguard_amd.attack.PGD import PGD  # noqa: E402
from speakerguard_amd.model.xv_plda import xv_plda  # noqa: E402


def med(fn, n=7):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    model = xv_plda.from_weights(synth.make_xv_weights(seed=0, D=200, n_spk=10), device=dev, dither=0.0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=1234)).to(dev)
    y = (torch.arange(B) % 10).to(dev)

    def pgd(k):
        return PGD(model, task="CSI", epsilon=0.002, step_size=0.0004, max_iter=k, loss="Entropy", targeted=False,
                   batch_size=64, EOT_size=1, EOT_batch_size=1, verbose=0)
    pgd(20).attack(x, y)
    t20 = med(lambda: pgd(20).attack(x, y), 5)
    t100 = med(lambda: pgd(100).attack(x, y), 3)
    t0 = med(lambda: pgd(0).attack(x, y), 7)
    b = (t100 - t20) / 80
    print("B=%d: attack(20) %.3f ms, attack(100) %.3f ms -> per step %.4f ms, fixed %.3f ms; attack(0 steps = the final pass only) %.3f ms" % (
        B, t20, t100, b, t20 - 20 * b, t0))
    a = pgd(20)
    print("  _check_inputs (x.max round trip)       %.3f ms" % med(lambda: a._check_inputs(x, y)))
    print("  two clamps (the epsilon ball)          %.3f ms" % med(lambda: (torch.clamp(x + 0.002, max=1), torch.clamp(x - 0.002, min=-1))))
    lower, upper = torch.clamp(x - 0.002, min=-1), torch.clamp(x + 0.002, max=1)
    base = model
    from speakerguard_amd.attack.utils import resolve_loss  # noqa: F401
    a._begin_attack()
    print("  forward-only pass (model.make_decision) %.3f ms" % med(lambda: model.make_decision(x)))
    print("  _run_batches, 0 steps                  %.3f ms" % med(lambda: pgd(0)._run_batches(x, y, lower, upper)))
    print("  _run_batches, 1 step                   %.3f ms" % med(lambda: pgd(1)._run_batches(x, y, lower, upper)))
    print("  _run_batches, 20 steps                 %.3f ms" % med(lambda: a._run_batches(x, y, lower, upper), 5))
    adv = x.clone()
    print("  torch.cat of one batch                 %.3f ms" % med(lambda: torch.cat([adv], 0)))
    print("  flags to the host (64 bools)           %.3f ms" % med(lambda: torch.zeros(B, dtype=torch.uint8, device=dev).bool().tolist()))
    print("  check_health                           %.3f ms" % med(lambda: base.check_health()))


if __name__ == "__main__":
    main()
