"""Throughput of the other BASELINE.json configurations on one GPU (the metric is configs[1]: bench.py).

    configs[0]  FGSM 1-step on AudioNet CSI-NE, a single 3 s utterance: ms per attack
    configs[2]  CW2 L2 targeted on xv_plda SV task, batch 32, Adam inner optimiser: iterations/s
    configs[3]  PGD + EOT 2 vs FeCo-defended AudioNet (randomised defense), 64 utterances = one GPU's shard of the
                batch of 512: ms per step; the undefended AudioNet loop at 64 and 512 next to it
    configs[4]  FAKEBOB / NES on xv_plda OSI, samples_per_draw 50: queries/s

``measure()`` is what ``bench.py`` puts into its JSON line as ``other_configs`` (every entry: workload string, median of
``reps`` timed runs after one warm-up, and the fraction of the f32-MFMA peak its algorithmic FLOPs amount to);
``python tools/config_bench.py`` prints the same plus the host-chained / deterministic-defense variants of configs[3].
"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PEAK_F32_MFMA_TFLOPS = 157.3
XV_FWD_GFLOP, XV_STEP_GFLOP = 2.351, 4.70   # per utterance: forward only / forward + data gradient
AN_FWD_GFLOP, AN_STEP_GFLOP = 0.078, 0.156  # AudioNet, same source
T = 48000
PEAK_HBM_TBS = 8.0  # MI355X_MICROARCH.md
# AudioNet HBM bytes per utterance and gradient step (3 s, 300 frames; VERDICT r4 item 2d: the AudioNet step is front-end
# bound, its MFMA fraction alone says little).  MANDATORY = what a fully fused step must move: read x, lower, upper, write x.
# AS BUILT = every tensor the launch sequence hands from one kernel to the next, once each way: waveform in / bounds / out,
# log-mel + mel energies (forward -> backward), the packed spectrum cache (300 x 512 complex float32, written and read), the
# per-frame sample gradients (300 x 800, written and read), the CNN activations the backward masks need (written, read) and
# the log-mel gradient.
AN_BYTES_MANDATORY = 4 * T * 4
AN_BYTES_AS_BUILT = (4 * T * 4 + 2 * 2 * 300 * 32 * 4 + 2 * 300 * 512 * 8 + 2 * 300 * 800 * 4 +
                     2 * 4 * (300 * 64 + 150 * 64 + 150 * 128 * 3 + 75 * 128 + 75 * 64 + 37 * 64 + 35 * 32 + 300 * 32) + 2 * 300 * 32 * 4)
# large batches (sg_an_configure's automatic choice, >= ~200 utterances of 3 s): the overlap-add runs inside the adjoint and
# the per-frame sample gradients never reach HBM
AN_BYTES_AS_BUILT_FUSED = AN_BYTES_AS_BUILT - 2 * 300 * 800 * 4


def _hbm(bytes_per_utt_step, utt_steps, seconds):
    return bytes_per_utt_step * utt_steps / seconds / 1e12 / PEAK_HBM_TBS


def _timed(fn, reps):
    """one warm-up, then `reps` timed runs; (last result, median seconds, all seconds)"""
    fn()
    out, dts = None, []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    return out, statistics.median(dts), dts


def _frac(gflop, seconds):
    return gflop / seconds / 1e3 / PEAK_F32_MFMA_TFLOPS


def measure(dev, reps=3, xv_weights=None, extras=False):
    from speakerguard_amd import synth
    from speakerguard_amd.attack.CW2 import CW2
    from speakerguard_amd.attack.FAKEBOB import FAKEBOB
    from speakerguard_amd.attack.FGSM import FGSM
    from speakerguard_amd.attack.PGD import PGD
    from speakerguard_amd.defense.feature_level import FeCoDefense
    from speakerguard_amd.model.audionet_csine import audionet_csine
    from speakerguard_amd.model.defended_model import defended_model
    from speakerguard_amd.model.xv_plda import xv_plda

    out = {}
    w = xv_weights if xv_weights is not None else synth.make_xv_weights()
    an = audionet_csine.from_weights(synth.make_audionet_state_dict(seed=0, num_class=251), device=dev)

    # ---- configs[0]: FGSM on AudioNet, ONE 3 s utterance (the reference's CPU-runnable plumbing case)
    x0 = torch.from_numpy(synth.make_waveforms(1, T, seed=1)).to(dev)
    y0 = an.make_decision(x0)[0]
    fgsm = FGSM(an, task="CSI", epsilon=0.002, batch_size=1, verbose=0)
    n0 = 20
    (_, succ0), dt, dts = _timed(lambda: [fgsm.attack(x0, y0) for _ in range(n0)][-1], reps)
    out["configs[0]"] = {"workload": "FGSM (1 step + final forward pass) on AudioNet CSI-NE, ONE 3 s utterance, attack() called %d times" % n0,
                         "value": 1e3 * dt / n0, "unit": "ms/attack", "samples": [1e3 * d / n0 for d in dts],
                         "frac_f32_mfma_peak": _frac((AN_STEP_GFLOP + AN_FWD_GFLOP) * n0, dt),
                         "roofline_hbm": _hbm(AN_BYTES_MANDATORY, n0, dt), "roofline_hbm_as_built": _hbm(AN_BYTES_AS_BUILT, n0, dt),
                         "bound": "launch latency (one utterance: ~40 dependent launches of a few microseconds)"}

    # ---- configs[2]: CW2, SV (one enrolled speaker, finite threshold), batch 32
    w1 = dict(w)
    w1["enroll"] = w["enroll"][:1]
    sv = xv_plda.from_weights(w1, threshold=-10.0, device=dev, dither=0.0)
    x = torch.from_numpy(synth.make_waveforms(32, T, seed=2)).to(dev)
    y = torch.zeros(32, dtype=torch.int64, device=dev)
    iters, steps = 30, 2
    atk = CW2(sv, task="SV", targeted=True, initial_const=1e-3, binary_search_steps=steps, max_iter=iters, stop_early=False,
              lr=1e-2, batch_size=32, verbose=0)
    (_, succ), dt, dts = _timed(lambda: atk.attack(x, y), reps)
    n_it = steps * (iters + 1)
    out["configs[2]"] = {"workload": "CW2 L2 targeted on xv_plda SV, batch 32 x 3 s, Adam lr 1e-2, %d search steps x (%d iterations + final pass)" % (steps, iters),
                         "value": n_it / dt, "unit": "iterations/s", "samples": [n_it / d for d in dts],
                         "utt_iterations_per_s": 32 * n_it / dt, "success": "%d/32" % sum(succ),
                         "frac_f32_mfma_peak": _frac(32 * steps * (iters * XV_STEP_GFLOP + XV_FWD_GFLOP), dt), "bound": "mfma"}

    # ---- configs[3]: PGD + EOT vs FeCo-defended AudioNet, 64 utterances per GPU
    xa = torch.from_numpy(synth.make_waveforms(64, T, seed=3)).to(dev)
    K = 20

    def feco_run(init, eot, fused):
        dm = defended_model(an, defense=[(1, FeCoDefense(0.5, init=init, seed=1))])
        ya = defended_model(an, defense=[(1, FeCoDefense(0.5))]).make_decision(xa)[0]
        pgd = PGD(dm, epsilon=0.002, step_size=0.0004, max_iter=K, batch_size=64, EOT_size=eot, EOT_batch_size=1, verbose=0)
        pgd.fuse_defended = fused
        return _timed(lambda: pgd.attack(xa, ya), reps)

    (_, succ), dt, dts = feco_run("random", 2, True)
    # the front-end runs once per step, the CNN once per EOT repeat (the defense is the only random part)
    an_cnn, an_front = 0.053, 0.025  # GFLOP forward per utterance
    g3 = 64 * (K * 2 * (an_front + 2 * an_cnn) + an_front + an_cnn)
    out["configs[3]"] = {"workload": "PGD-%d + EOT 2 vs FeCo-defended AudioNet (k-means from fresh random frames per repeat), 64 x 3 s = one GPU's "
                                     "shard of the batch of 512, ONE device loop (sg_an_pgd_run_feco)" % K,
                         "value": 1e3 * dt / K, "unit": "ms/step", "samples": [1e3 * d / K for d in dts], "success": "%d/64" % sum(succ),
                         "frac_f32_mfma_peak": _frac(g3, dt),
                         "roofline_hbm": _hbm(AN_BYTES_MANDATORY, 64 * K, dt), "roofline_hbm_as_built": _hbm(AN_BYTES_AS_BUILT, 64 * K, dt),
                         "bound": "latency: k-means (two CUs per utterance and repeat, ~5 Lloyd iterations of ~13 us + 16 us set-up) + the fused CNN at 128 rows"}
    for b in (64, 512):
        xb = torch.from_numpy(synth.make_waveforms(b, T, seed=5)).to(dev)
        yb = an.make_decision(xb)[0]
        plain = PGD(an, epsilon=0.002, step_size=0.0004, max_iter=K, batch_size=b, verbose=0)
        (_, _), dt, dts = _timed(lambda: plain.attack(xb, yb), reps)
        # HBM side of the same launch sequence: what one step has to move if every activation crosses HBM exactly once each way
        out["audionet_pgd_b%d" % b] = {"workload": "PGD-%d on the undefended AudioNet, %d x 3 s, ONE device loop (sg_an_pgd_run)" % (K, b),
                                       "value": 1e3 * dt / K, "unit": "ms/step", "samples": [1e3 * d / K for d in dts],
                                       "utt_steps_per_s": b * K / dt,
                                       "frac_f32_mfma_peak": _frac(b * (K * AN_STEP_GFLOP + AN_FWD_GFLOP), dt),
                                       "roofline_hbm": _hbm(AN_BYTES_MANDATORY, b * K, dt),
                                       "roofline_hbm_as_built": _hbm(AN_BYTES_AS_BUILT_FUSED if b >= 256 else AN_BYTES_AS_BUILT, b * K, dt),
                                       "hbm_bytes_per_utt_step": {"mandatory": AN_BYTES_MANDATORY,
                                                                  "as_built": AN_BYTES_AS_BUILT_FUSED if b >= 256 else AN_BYTES_AS_BUILT},
                                       "bound": "front-end VALU + LDS issue (float32 STFT, one wave per frame) and the CNN's per-layer serial parts"}

    # ---- configs[4]: FAKEBOB / NES, OSI, samples_per_draw 50
    osi = xv_plda.from_weights(w, threshold=-10.0, device=dev, dither=0.0)
    xq = torch.from_numpy(synth.make_waveforms(8, T, seed=4)).to(dev)
    yq = (osi.make_decision(xq)[0].clamp(min=0) + 3) % 10   # targeted at another speaker: no example finishes in 5 iterations
    iters = 5
    fb = FAKEBOB(osi, threshold=-10.0, task="OSI", targeted=True, epsilon=0.002, max_iter=iters, samples_per_draw=50,
                 samples_per_draw_batch_size=50, stop_early=False, batch_size=8, verbose=0)
    (_, succ), dt, dts = _timed(lambda: fb.attack(xq, yq), reps)
    q = 8 * 51 * (iters + 1)
    out["configs[4]"] = {"workload": "FAKEBOB OSI targeted, NES 50 + 1 queries per example and iteration, 8 x 3 s, %d iterations + final pass "
                                     "(forward-only model passes of 408 queries)" % iters,
                         "value": q / dt, "unit": "queries/s", "samples": [q / d for d in dts],
                         "frac_f32_mfma_peak": _frac(q * XV_FWD_GFLOP, dt), "bound": "mfma"}
    if extras:
        ex = {}
        for label, init, fused in (("random-init k-means, host-chained gradient", "random", False),
                                   ("evenly started k-means, host-chained gradient", "even", False),
                                   ("evenly started k-means, ONE device loop", "even", True)):
            (_, succ), dt, _ = feco_run(init, 2, fused)
            ex[label] = {"ms_per_step": 1e3 * dt / K, "success": "%d/64" % sum(succ)}
        out["configs[3]_variants"] = ex
    return out


if __name__ == "__main__":
    import json
    res = measure(torch.device("cuda:0"), reps=3, extras=True)
    for k, v in res.items():
        print(k, json.dumps(v))
