"""Headline benchmark: PGD attack steps/s on the x-vector/PLDA system (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload per GPU: PGD, L-inf eps 0.002, step 0.0004, cross-entropy, untargeted, EOT 1/1, CSI-E,
batch 64 x 3 s @ 16 kHz synthetic utterances, synthetic seeded weights (D=200, 10 enrolled
speakers), dither off.  One "step" = forward + hand-coded backward to d loss/d waveform + fused
sign/project/clamp update for the whole batch of 64.  The timed region is ONE ``sg_xv_pgd_run``
call with max_iter = K, i.e. K steps plus the attack's final forward-only evaluation pass
(reference attack/FGSM.py:44-47) -- the pass is part of every real attack, so it is charged to
the K steps rather than hidden.  Inputs are resident in HBM before the clock starts.

N > 1: weak scaling, every rank attacks its own batch of 64 (no data-path collective); the only
exchange is one RCCL all-gather of the per-utterance success flags at the end of the attack,
inside the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
B_PER_GPU, T_SAMPLES = 64, 48000
EPS, STEP = 0.002, 0.0004
# algorithmic MACs per utterance of the TDNN contractions, forward; data-gradient = same
FLOP_PER_UTT_STEP = 4.70e9


def cpu_baseline(weights, budget_utts=32, steps=3, gpu_model=None):
    """Reference-equivalent CPU path = the oracle in its structure-faithful form (per-utterance
    MFCC/TDNN loops, Python CMVN loop, autograd with parameters requiring grad), timed on this
    host's cores on a bounded sample of the same workload."""
    from oracle import attacks as oatk
    from oracle.xv_plda import XvPlda
    from speakerguard_amd import synth
    cores = torch.get_num_threads()
    model = XvPlda(weights, faithful=True, freeze=False)
    x = torch.from_numpy(synth.make_waveforms(budget_utts, T_SAMPLES, seed=1234))
    y = torch.arange(budget_utts) % 10
    atk = oatk.PGD(model, task="CSI", epsilon=EPS, step_size=STEP, max_iter=steps, batch_size=budget_utts)
    t0 = time.perf_counter()
    oadv, osucc = atk.attack(x, y)
    dt = time.perf_counter() - t0
    utt_steps = budget_utts * steps  # (+ one forward-only pass, charged like on the GPU side)
    # second row SURVEY.md section 8(d) asks for: the same oracle vectorised (batched, closed-form CMVN, parameters
    # frozen) -- fairer to the CPU, still not the optimisation target
    vmodel = XvPlda(weights, faithful=False, freeze=True)
    vatk = oatk.PGD(vmodel, task="CSI", epsilon=EPS, step_size=STEP, max_iter=steps, batch_size=budget_utts)
    t1 = time.perf_counter()
    vatk.attack(x, y)
    dtv = time.perf_counter() - t1
    parity = None
    if gpu_model is not None:
        # the checker role of the oracle: the same PGD-%d on the same utterances through the HIP path
        from speakerguard_amd.attack.PGD import PGD
        dev = gpu_model.device
        adv, succ = PGD(gpu_model, task="CSI", epsilon=EPS, step_size=STEP, max_iter=steps, batch_size=budget_utts,
                        verbose=0).attack(x.to(dev), y.to(dev))
        diff = (adv.cpu() - oadv).abs()
        with torch.no_grad():
            odec = model.make_decision(oadv)[0]
        parity = {"utterances": budget_utts, "pgd_steps": steps,
                  "success_flags_equal": [bool(a) for a in succ] == [bool(a) for a in osucc],
                  "decisions_on_adversarial_audio_equal": gpu_model.make_decision(adv)[0].cpu().tolist() == odec.tolist(),
                  "perturbation_samples_differing": float((diff > 1e-7).float().mean()),
                  "perturbation_max_abs_diff": float(diff.max())}
    return {
        "parity_vs_oracle": parity,
        "vectorised_value": utt_steps / dtv / B_PER_GPU,
        "vectorised_sample": "same sample, batched oracle with frozen parameters: %.1f s" % dtv,
        "value": utt_steps / dt / B_PER_GPU,
        "unit": "steps/s",
        "cores": cores,
        "kind": "port",
        "sample": "oracle faithful path (per-utterance loops, autograd incl. weight grads), PGD-%d on %d of the 64 "
                  "utterances = %d utterance-steps in %.1f s; scaled to batch-64 steps" % (steps, budget_utts, utt_steps, dt),
        "utt_steps_per_s": utt_steps / dt,
    }


def gather_flags(mine, dist, world):
    """The attack's only exchange: every rank's per-utterance success flags, in rank order (uint8: RCCL has no bool)."""
    if dist is None:
        return mine
    mine = mine.to(torch.uint8)
    flags = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(flags, mine)
    return torch.cat(flags)


def timed_region(attack, steps, warmup, dist, sync, dev):
    """W untimed warm-up steps, then exactly `steps` steps bracketed by barrier + device synchronise on both sides;
    returns (attack result, seconds = MAX over ranks).  Factored out so the N > 1 protocol is covered by a
    world-size-2 gloo test on the CPU (tests/test_bench_protocol.py)."""
    if warmup > 0:
        attack(warmup)
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    result = attack(steps)
    sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return result, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves as a CHILD torchrun (nothing in this process
        # has touched the GPU yet) and leave with its exit code -- never silently report a 1-GPU number for N > 1
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from speakerguard_amd import synth
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    from speakerguard_amd.model.xv_plda import xv_plda

    weights = synth.make_xv_weights(seed=0, D=200, n_spk=10)
    model = xv_plda.from_weights(weights, device=dev, dither=0.0)
    x = torch.from_numpy(synth.make_waveforms(B_PER_GPU, T_SAMPLES, seed=1234 + rank)).to(dev)
    y = (torch.arange(B_PER_GPU) % 10).to(dev)
    lower, upper = torch.clamp(x - EPS, min=-1), torch.clamp(x + EPS, max=1)
    spec = SEC4SR_CrossEntropy()

    def attack(k):
        out = model.pgd_run(x, y, lower, upper, spec, STEP, k, 1)
        return out, gather_flags(out[1], dist, world)  # inside the timed region

    (out, flags), dt = timed_region(attack, args.steps, args.warmup, dist, torch.cuda.synchronize, dev)

    # roofline of the dominant kernel: TDNN layer 3 forward contraction (stream-K, 128x128 quad-fed tiles),
    # HIP events on the launch stream inside the library
    ms, flops, _ = model.time_layer(3, B_PER_GPU, T_SAMPLES, iters=20)
    achieved = flops / (ms * 1e-3) / 1e12
    # memory-side bytes of that launch come from the committed rocprofv3 --pmc passes (cannot be collected
    # from inside this process); null if the summary is missing
    traffic, traffic_src = None, None
    for name in ("r02_pmc_tdnn3.json", "r01_pmc_tdnn3.json"):  # latest committed passes of this kernel
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
        if os.path.exists(pmc):
            with open(pmc) as f:
                traffic = json.load(f).get("traffic_bytes_per_launch")
            traffic_src = "profiles/%s (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; fabric bytes incl. Infinity-Cache hits)" % name
            break
    roofline = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "conv_gemm_streamk_kernel<BIAS_RELU, KIND 2: 16 waves, 256x128 quad-fed> tdnn3 forward "
                          "(B=64: M=17280 N=512 K=3584); 8 of the 10 contractions of a step run this kernel (86 % of the step)",
                "ms_per_launch": ms, "flop_per_launch": flops,
                # whole step against the same peak: algorithmic TDNN FLOPs of the job / wall time (front-end, pooling, tail and
                # the final forward-only pass of the attack included in the time, not in the FLOPs)
                "end_to_end_frac": world * args.steps * B_PER_GPU * FLOP_PER_UTT_STEP / dt / 1e12 / (PEAK_F32_MFMA_TFLOPS * world)}

    steps_per_s = world * args.steps / dt
    line = {
        "metric": "PGD attack steps/sec (xv_plda, 3s@16kHz, batch 64)",
        "value": steps_per_s,
        "unit": "steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "PGD L-inf eps=0.002 step=0.0004 CE untargeted on xv_plda CSI-E, batch 64 x 3 s @ 16 kHz "
                               "per GPU, EOT 1/1, dither off; timed = K steps + final forward-only pass",
                   "batch_per_gpu": B_PER_GPU, "samples": T_SAMPLES, "global_batch": B_PER_GPU * world},
        "utt_steps_per_s": steps_per_s * B_PER_GPU,
        "model_tflops": steps_per_s * B_PER_GPU * FLOP_PER_UTT_STEP / 1e12 / world,
        "success_count": int(flags.sum().item()),
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(weights, gpu_model=model)
        line["gpu_vs_cpu"] = steps_per_s / line["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
