"""bench.py's N > 1 protocol (warm-up, barrier + synchronise on both sides of exactly K steps, MAX over ranks, rank-ordered
all-gather of the success flags) on a world-size-2 gloo group on the CPU; the GPU ranks run the same two functions over
RCCL.  Also: the launcher contract (--gpus must match the world size)."""
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import bench
    calls = []

    def attack(k):
        calls.append(k)
        time.sleep(0.05 * k * (1 + rank))                      # rank 1 is twice as slow
        mine = torch.tensor([rank == 0, True, rank == 1])       # per-"utterance" success flags of this shard
        return k, bench.gather_flags(mine, dist, world)

    (k, flags), dt = bench.timed_region(attack, 4, 2, dist, lambda: None, torch.device("cpu"))
    if rank == 0:
        torch.save({"calls": calls, "flags": flags, "dt": dt, "k": k}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_protocol(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "r.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["calls"] == [2, 4] and r["k"] == 4                 # W warm-up steps, then exactly K
    assert r["flags"].tolist() == [1, 1, 0, 0, 1, 1]            # rank order
    assert 0.38 <= r["dt"] < 1.0, r["dt"]                       # the slower rank's 0.4 s, not rank 0's own 0.2 s


def test_gpus_flag_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_repeated_regions_and_median_summary():
    """warm-up once, then R timed attacks of exactly K steps; the line reports the median and keeps
    every sample."""
    sys.path.insert(0, ROOT)
    import bench
    calls = []

    def attack(k):
        calls.append(k)
        time.sleep(0.01 * (3 if len(calls) == 2 else 1))  # the first timed attack is the slow one (clock ramp)
        return k, None

    (k, _), samples = bench.timed_reps(attack, 4, 2, 5, None, lambda: None, torch.device("cpu"))
    assert calls == [2, 4, 4, 4, 4, 4] and k == 4 and len(samples) == 5
    s = bench.summarise(samples, 4)
    assert s["ms_per_step_min"] <= s["ms_per_step"] <= s["ms_per_step_max"]
    assert s["ms_per_step"] < 1e3 * samples[0] / 4          # the median is not the slow first sample
    assert len(s["ms_per_step_samples"]) == 5


def test_stage_roofline_arithmetic():
    """`roofline.achieved` = FLOPs of the 8 stream-K launches of a step / sum of their average durations."""
    sys.path.insert(0, ROOT)
    import bench

    class Fake:
        def trace_stages(self, fn, max_records):
            fn()
            recs = []
            for it in range(3):  # two gradient steps + the forward-only pass
                recs += [("mfcc_fwd", 0.05)] + [("tdnn%d_fwd" % l, 0.1 * l) for l in (1, 2, 3, 4, 5)] + [("tail", 0.03)]
                if it < 2:
                    recs += [("tdnn%d_dgrad" % l, 0.2 * l) for l in (5, 4, 3, 2, 1)]
            return recs
    r = bench.stage_roofline(Fake(), lambda k: None, 64, 2)
    t_us = 1e3 * (0.1 * (2 + 3 + 4 + 5) + 0.2 * (5 + 4 + 3 + 2))
    fl = 2 * 2.0 * 64 * sum(bench.LAYER_MACS[l] for l in (2, 3, 4, 5))
    assert abs(r["us_per_step_in_kernel"] - t_us) < 1e-6 and abs(r["flop_per_step"] - fl) < 1
    assert abs(r["achieved"] - fl / (t_us * 1e-6) / 1e12) < 1e-9
    assert r["stages"]["tdnn3_fwd"]["launches"] == 3 and r["stages"]["tdnn3_dgrad"]["launches"] == 2
