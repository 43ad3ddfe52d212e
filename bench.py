"""Headline benchmark: PGD attack steps/s on the x-vector/PLDA system (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reps R] [--scaling weak|strong] [--batch-per-gpu B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload: PGD, L-inf eps 0.002, step 0.0004, cross-entropy, untargeted, EOT 1/1, CSI-E, 3 s @ 16 kHz synthetic
utterances, synthetic seeded weights (D=200, 10 enrolled speakers), dither off.  One "step" = forward + hand-coded
backward to d loss/d waveform + fused sign/project/clamp update for a whole batch of 64.  A timed region is ONE
``sg_xv_pgd_run`` call with max_iter = K, i.e. exactly K steps plus the attack's final forward-only evaluation pass
(reference attack/FGSM.py:44-47) -- the pass is part of every real attack, so it is charged to the K steps rather
than hidden.  Inputs are resident in HBM before the clock starts.

Protocol: W untimed warm-up steps, then R timed attacks of exactly K steps, each bracketed by
barrier + device synchronise on both sides and timed as the MAX over ranks; `value` is computed from the MEDIAN attack,
all samples are in the line.

Every timed attack is the PRODUCT's call: ``speakerguard_amd.attack.PGD(model, ..., batch_size=64).attack(x, y)``
(reference attack/PGD.py:40-79: input checks, epsilon box, one chunk -> the fused device loop, flags to the host), and
for N > 1 ``speakerguard_amd.shard.ShardedAttack`` around it -- not a bench-private partition.

N > 1, two partitions of the work, both measured in the same run:
  * ``--scaling weak`` (default, the line's `value`): every rank attacks its own batch of 64; no data-path collective,
    one RCCL all-gather of the per-utterance success flags at the end of the attack, inside the timed region;
  * ``strong`` (the line's ``value_metric_partition`` / ``strong_scaling``; `value` with ``--scaling strong``): the
    metric's own partition -- ONE batch of 64, ``ShardedAttack(PGD(batch_size=64)).attack(x64, y64)`` on every rank, i.e.
    contiguous shards of 64/N utterances (reference attack/PGD.py:62-73 chunks one batch; BASELINE.md section 3 "batch 64
    sharded B/G per GPU"), flags all-gathered by the wrapper.  On one GPU the shard sizes 32 / 16 / 8 are timed as well
    (``shard_points``): what a rank of a 2 / 4 / 8-GPU strong-scaling run does.

``other_configs``: the other BASELINE.json configurations (configs[0], [2], [3], [4]) on this GPU, a few seconds in
all (tools/config_bench.py); ``--no-other-configs`` skips them.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
GLOBAL_BATCH, T_SAMPLES = 64, 48000
EPS, STEP = 0.002, 0.0004
# algorithmic MACs per utterance of the TDNN contractions, forward; data-gradient = same
FLOP_PER_UTT_STEP = 4.70e9
LAYER_MACS = {1: 22732800, 2: 377487360, 3: 495452160, 4: 70778880, 5: 207360000}  # per utterance, one direction
STREAMK_STAGES = ["tdnn%d_fwd" % l for l in (2, 3, 4, 5)] + ["tdnn%d_dgrad" % l for l in (5, 4, 3, 2)]


def _timed(fn):
    t0 = time.perf_counter()
    out = fn()
    return out, time.perf_counter() - t0


def cpu_baseline(weights, budget_utts=32, steps=5, gpu_model=None):
    """Reference-equivalent CPU path = the oracle in its structure-faithful form (per-utterance MFCC/TDNN loops, Python
    CMVN loop, autograd with parameters requiring grad), timed on this host's cores on a bounded sample of the same
    workload: thread count picked by a quick sweep, one warm-up, median of three."""
    from oracle import attacks as oatk
    from oracle.xv_plda import XvPlda
    from speakerguard_amd import synth
    avail = os.cpu_count() or 1
    x = torch.from_numpy(synth.make_waveforms(budget_utts, T_SAMPLES, seed=1234))
    y = torch.arange(budget_utts) % 10
    model = XvPlda(weights, faithful=True, freeze=False)
    vmodel = XvPlda(weights, faithful=False, freeze=True)
    kw = dict(task="CSI", epsilon=EPS, step_size=STEP)

    # thread sweep on a small piece of the workload (batch-1 convolutions oversubscribe easily; BASELINE.md section 3 said
    # "N = physical cores" -- 128 threads run this path 8x slower than 16, so the count is measured, and stated): PGD-1 on 8
    # utterances, best of two runs per candidate (round 3 timed 4 utterances once: ~0.1 s, noisy ground for a choice)
    sweep = {}
    for nt in sorted({n for n in (8, 16, 32, 64, 128) if n <= avail} | {min(avail, 8)}):
        torch.set_num_threads(nt)
        atk = oatk.PGD(model, max_iter=1, batch_size=8, **kw)
        atk.attack(x[:2], y[:2])  # touch
        sweep[nt] = min(_timed(lambda: atk.attack(x[:8], y[:8]))[1] for _ in range(2))
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)

    def run(m):
        atk = oatk.PGD(m, max_iter=steps, batch_size=budget_utts, **kw)
        atk.attack(x, y)  # warm-up
        outs = [_timed(lambda: atk.attack(x, y)) for _ in range(3)]
        return outs[-1][0], [dt for _, dt in outs]

    (oadv, osucc), dts = run(model)
    _, vdts = run(vmodel)
    dt, dtv = statistics.median(dts), statistics.median(vdts)
    utt_steps = budget_utts * steps  # (+ one forward-only pass, charged like on the GPU side)
    parity = None
    if gpu_model is not None:
        # the checker role of the oracle: the same PGD on the same utterances through the HIP path
        from speakerguard_amd.attack.PGD import PGD
        dev = gpu_model.device
        adv, succ = PGD(gpu_model, max_iter=steps, batch_size=budget_utts, verbose=0, **kw).attack(x.to(dev), y.to(dev))
        diff = (adv.cpu() - oadv).abs()
        with torch.no_grad():
            odec = model.make_decision(oadv)[0]
        parity = {"utterances": budget_utts, "pgd_steps": steps,
                  "success_flags_equal": [bool(a) for a in succ] == [bool(a) for a in osucc],
                  "decisions_on_adversarial_audio_equal": gpu_model.make_decision(adv)[0].cpu().tolist() == odec.tolist(),
                  "perturbation_samples_differing": float((diff > 1e-7).float().mean()),
                  "perturbation_max_abs_diff": float(diff.max())}
    return {
        "value": utt_steps / dt / GLOBAL_BATCH,
        "unit": "steps/s",
        "cores": cores,
        "kind": "port",
        "sample": "oracle faithful path (per-utterance loops, autograd incl. weight grads), PGD-%d on %d of the 64 utterances = "
                  "%d utterance-steps, one warm-up + median of 3 runs (%s s), %d threads picked by a sweep; scaled to batch-64 steps"
                  % (steps, budget_utts, utt_steps, ", ".join("%.1f" % d for d in dts), cores),
        "utt_steps_per_s": utt_steps / dt,
        "samples_s": dts,
        "thread_sweep_s": {str(k): v for k, v in sweep.items()},
        "host_cores": avail,
        "vectorised_value": utt_steps / dtv / GLOBAL_BATCH,
        "vectorised_sample": "same sample, batched oracle with frozen parameters: median of 3 runs (%s s)" % ", ".join("%.1f" % d for d in vdts),
        "parity_vs_oracle": parity,
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


def timed_reps(attack, steps, warmup, reps, dist, sync, dev):
    """`reps` timed regions of exactly `steps` steps (the warm-up before the first only); returns (last result, [seconds])."""
    result, samples = None, []
    for r in range(max(1, reps)):
        result, dt = timed_region(attack, steps, warmup if r == 0 else 0, dist, sync, dev)
        samples.append(dt)
    return result, samples


def summarise(samples, steps):
    ms = sorted(1e3 * s / steps for s in samples)
    return {"ms_per_step": statistics.median(ms), "ms_per_step_min": ms[0], "ms_per_step_max": ms[-1],
            "ms_per_step_samples": [1e3 * s / steps for s in samples]}


def stage_roofline(model, run_attack, B, steps):
    """In-loop per-launch times from HIP-event pairs on the launch stream (sg_trace_begin / sg_trace_end) over one traced
    attack of the timed workload: flop-weighted fraction of the f32-MFMA peak over the 8 stream-K contraction launches of
    a step, plus the per-stage table.  The events cost a few microseconds per launch: its own run, not a timed one."""
    recs = model.trace_stages(lambda: run_attack(steps), max_records=64 * (steps + 2))
    by = {}
    for name, ms in recs:
        by.setdefault(name, []).append(ms)
    stages = {k: {"launches": len(v), "avg_us": 1e3 * statistics.mean(v), "min_us": 1e3 * min(v)} for k, v in by.items()}
    for k, st in stages.items():
        if k.startswith("tdnn"):
            st["tflops"] = 2.0 * LAYER_MACS[int(k[4])] * B / (st["avg_us"] * 1e-6) / 1e12
    sk_time = sum(stages[k]["avg_us"] for k in STREAMK_STAGES)                       # us per step (dgrad launches: K of K+1 passes)
    sk_flop = sum(2.0 * LAYER_MACS[int(k[4])] * B for k in STREAMK_STAGES)
    fwd = [ms for k in STREAMK_STAGES[:4] for ms in by[k]]
    bwd = [ms for k in STREAMK_STAGES[4:] for ms in by[k]]
    return {"achieved": sk_flop / (sk_time * 1e-6) / 1e12, "us_per_step_in_kernel": sk_time, "flop_per_step": sk_flop,
            "avg_us_forward_instantiation": 1e3 * statistics.mean(fwd), "avg_us_dgrad_instantiation": 1e3 * statistics.mean(bwd),
            "stages": stages, "traced_records": len(recs)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps", type=int, default=5, help="timed attacks of --steps steps; the line reports the median")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--batch-per-gpu", type=int, default=0, help="utterances per GPU (default: 64 weak, 64 / N strong)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shard-points", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    # test hooks (tests/test_gpu_multiproc.py runs the N > 1 code path as two ranks on the ONE GPU of the test box)
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)
    ap.add_argument("--one-gpu", action="store_true", help=argparse.SUPPRESS)
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
        if args.one_gpu:
            local = 0
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    from speakerguard_amd.model.xv_plda import xv_plda
    from speakerguard_amd.shard import ShardedAttack

    weights = synth.make_xv_weights(seed=0, D=200, n_spk=10)
    model = xv_plda.from_weights(weights, device=dev, dither=0.0)
    sync = torch.cuda.synchronize

    def pgd(k):
        """the product's attack object for K steps (PGD-K: K steps + the final forward-only pass)"""
        return PGD(model, task="CSI", epsilon=EPS, step_size=STEP, max_iter=k, loss="Entropy", targeted=False,
                   batch_size=GLOBAL_BATCH, EOT_size=1, EOT_batch_size=1, verbose=0)

    def make_attack(x_host, y_host, sharded=False):
        """Resident inputs + the attack closure of one partition.  sharded: ShardedAttack cuts the batch over the ranks and
        all-gathers the flags itself; otherwise the rank attacks all of (x, y) and the flags are gathered here -- inside
        the timed region either way."""
        x = torch.from_numpy(np.ascontiguousarray(x_host)).to(dev)
        y = torch.from_numpy(np.ascontiguousarray(y_host)).to(dev)

        def attack(k, gather=True):
            if sharded:
                adv, succ = ShardedAttack(pgd(k), gather_audio=False).attack(x, y)
                return adv, torch.tensor(succ, dtype=torch.uint8, device=dev)
            adv, succ = pgd(k).attack(x, y)
            flags = torch.tensor(succ, dtype=torch.uint8, device=dev if world > 1 else "cpu")  # (one rank: nothing to exchange)
            return adv, (gather_flags(flags, dist, world) if gather else flags)
        return attack

    labels = np.arange(GLOBAL_BATCH) % 10
    # weak: every rank its own 64 utterances; strong: ONE batch of 64 (seed 1234), resident on every rank, cut by ShardedAttack
    weak_b = args.batch_per_gpu if (args.batch_per_gpu > 0 and args.scaling == "weak") else GLOBAL_BATCH
    weak = make_attack(synth.make_waveforms(weak_b, T_SAMPLES, seed=1234 + rank), np.arange(weak_b) % 10)
    global_x = synth.make_waveforms(GLOBAL_BATCH, T_SAMPLES, seed=1234)
    if args.batch_per_gpu > 0 and args.scaling == "strong":
        strong_b = args.batch_per_gpu  # one GPU standing in for a rank of a 64 / B-GPU run (no exchange)
        strong = make_attack(global_x[:strong_b], labels[:strong_b])
    else:
        lo, hi = ShardedAttack(pgd(1)).bounds(GLOBAL_BATCH)[rank] if world > 1 else (0, GLOBAL_BATCH)
        strong_b = hi - lo
        if strong_b < 1:
            sys.exit("bench.py: more ranks (%d) than utterances in the metric's batch" % world)
        strong = make_attack(global_x, labels, sharded=world > 1)

    primary, primary_b = (weak, weak_b) if args.scaling == "weak" else (strong, strong_b)
    (out, flags), samples = timed_reps(primary, args.steps, args.warmup, args.reps, dist, sync, dev)
    prim = summarise(samples, args.steps)
    dt = prim["ms_per_step"] * 1e-3 * args.steps
    # a "step" of the metric is a batch-64 step: weak = N of them per time step of the job, strong = one
    strong_jobs = strong_b / GLOBAL_BATCH if args.batch_per_gpu > 0 else 1.0  # ShardedAttack: the ranks share ONE batch of 64
    jobs_per_step = world * primary_b / GLOBAL_BATCH if args.scaling == "weak" else strong_jobs
    steps_per_s = jobs_per_step * args.steps / dt

    # ---- the other partition, same run (N > 1), or the per-rank shard sizes of 2 / 4 / 8 GPUs (N = 1)
    other = None
    if world > 1:
        sec, sec_b = (strong, strong_b) if args.scaling == "weak" else (weak, weak_b)
        (_, sflags), ssamples = timed_reps(sec, args.steps, min(args.warmup, 2), 3, dist, sync, dev)
        s = summarise(ssamples, args.steps)
        mult = strong_jobs if args.scaling == "weak" else world * sec_b / GLOBAL_BATCH
        other = dict(s, scaling="strong" if args.scaling == "weak" else "weak", batch_per_gpu=sec_b,
                     value=mult * 1e3 / s["ms_per_step"], unit="steps/s", success_count=int(sflags.sum().item()),
                     note="ShardedAttack(PGD(batch_size=64)).attack(x64, y64): one batch of 64 cut into contiguous shards of 64 / N "
                          "utterances (attack/PGD.py:62-73), flags all-gathered inside the timed region" if args.scaling == "weak"
                          else "every rank its own batch")
    shard_points = None
    if world == 1 and not args.no_shard_points:
        shard_points = []
        for b in (32, 16, 8):
            atk = make_attack(global_x[:b], labels[:b])
            # (a full attack as warm-up: at 8 utterances three steps are 1.5 ms, and the chip needs ~10 ms of work to reach its clock)
            # five timed attacks like the headline (three left the median to one disturbed sample: 0.84 / 0.86 / 0.75 ms at 16 on one box)
            _, ss = timed_reps(atk, args.steps, max(args.warmup, args.steps), max(args.reps, 3), None, sync, dev)
            s = summarise(ss, args.steps)
            shard_points.append(dict(s, batch_per_gpu=b, n_gpus_of_a_strong_run=GLOBAL_BATCH // b,
                                     steps_per_s_of_that_run_without_exchange=1e3 / s["ms_per_step"],
                                     utt_steps_per_s=b * 1e3 / s["ms_per_step"],
                                     end_to_end_frac=b * FLOP_PER_UTT_STEP / (s["ms_per_step"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS))

    # ---- roofline of the dominant kernel (conv_gemm_streamk_kernel: 8 of the 10 contractions of a step), in the loop
    roof = stage_roofline(model, lambda k: primary(k, gather=False), primary_b, args.steps) if rank == 0 else None
    if dist is not None:
        dist.barrier()
    line = None
    if rank == 0:
        ms_iso, flops_iso, _ = model.time_layer(3, primary_b, T_SAMPLES, iters=20)
        traffic, traffic_src = None, None
        for name in ("r06_pmc_tdnn3.json", "r05_pmc_tdnn3.json", "r04_pmc_tdnn3.json", "r03_pmc_tdnn3.json", "r02_pmc_tdnn3.json", "r01_pmc_tdnn3.json"):  # latest committed passes of this kernel
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc):
                with open(pmc) as f:
                    traffic = json.load(f).get("traffic_bytes_per_launch")
                traffic_src = "profiles/%s: one tdnn3 forward launch at batch 64 (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; fabric " \
                              "bytes incl. Infinity-Cache hits; cannot be collected from inside this process)" % name
                break
        roofline = {
            "bound": "mfma", "achieved": roof["achieved"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": roof["achieved"] / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": "conv_gemm_streamk_kernel (tdnn2-5 forward, tdnn5-2 data gradient: 8 launches per step, 98 % of a step's FLOPs)",
            "how": "algorithmic FLOPs of the 8 launches / sum of their average in-loop durations; HIP-event pairs on the launch stream "
                   "around every launch of one traced attack of the timed workload (sg_trace_begin / sg_trace_end); compare "
                   "avg_us_*_instantiation with the rocprofv3 --kernel-trace --stats averages of the same command (profiles/)",
            "us_per_step_in_kernel": roof["us_per_step_in_kernel"], "flop_per_step": roof["flop_per_step"],
            "avg_us_forward_instantiation": roof["avg_us_forward_instantiation"],
            "avg_us_dgrad_instantiation": roof["avg_us_dgrad_instantiation"],
            "stages": roof["stages"],
            "isolated_tdnn3_forward": {"ms_per_launch": ms_iso, "frac": flops_iso / (ms_iso * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                       "note": "the best layer launched back to back with a warm L2 (sg_xv_time_layer): an upper bound, not `frac`"},
            # whole step against the same peak: algorithmic TDNN FLOPs of the job / wall time (front-end, pooling, tail and the
            # final forward-only pass of the attack included in the time, not in the FLOPs)
            "end_to_end_frac": primary_b * FLOP_PER_UTT_STEP / (prim["ms_per_step"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
        }
        line = {
            "metric": "PGD attack steps/sec (xv_plda, 3s@16kHz, batch 64)",
            "value": steps_per_s,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": prim["ms_per_step"],
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "PGD L-inf eps=0.002 step=0.0004 CE untargeted on xv_plda CSI-E, %d x 3 s @ 16 kHz per GPU (%s), "
                                   "EOT 1/1, dither off; timed = K steps + final forward-only pass; median of %d attacks"
                                   % (primary_b, "every rank its own batch" if args.scaling == "weak" else "one batch of 64 cut over the ranks",
                                      len(samples)),
                       "batch_per_gpu": primary_b, "samples": T_SAMPLES,
                       "global_batch": primary_b * world},
            "timed_attacks": len(samples),
            "ms_per_step_min": prim["ms_per_step_min"], "ms_per_step_max": prim["ms_per_step_max"],
            "ms_per_step_samples": prim["ms_per_step_samples"],
            "utt_steps_per_s": world * primary_b * args.steps / dt,
            "model_tflops_per_gpu": primary_b * args.steps / dt * FLOP_PER_UTT_STEP / 1e12,
            "success_count": int(flags.sum().item()),
            "roofline": roofline,
        }
        # the metric's own partition (GPU-N = the batch of 64 cut B/N per GPU), next to `value`
        if args.scaling == "strong" or world == 1:
            line["value_metric_partition"] = steps_per_s
        elif other is not None:
            line["value_metric_partition"] = other["value"]
        line["value_metric_partition_note"] = ("steps/s of ONE batch of 64 cut over the %d GPUs by the product's ShardedAttack (strong "
                                               "scaling); `value` is %s" % (world, "the same number" if (args.scaling == "strong" or world == 1)
                                                                            else "weak scaling: every rank its own batch of 64"))
        if other is not None:
            line["strong_scaling" if args.scaling == "weak" else "weak_scaling"] = other
        if shard_points is not None:
            line["shard_points"] = shard_points
        if world == 1 and not args.no_other_configs:
            import contextlib
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import config_bench
            with contextlib.redirect_stdout(sys.stderr):  # the attack classes print like the reference's; the line stays alone
                t0 = time.perf_counter()
                line["other_configs"] = config_bench.measure(dev, reps=3, xv_weights=weights)
                line["other_configs"]["seconds_spent"] = time.perf_counter() - t0
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(weights, gpu_model=model)
            line["gpu_vs_cpu"] = steps_per_s / line["cpu_baseline"]["value"]
        order = ["metric", "value", "value_metric_partition"]
        line = {**{k: line[k] for k in order if k in line}, **{k: v for k, v in line.items() if k not in order}}
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
