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
