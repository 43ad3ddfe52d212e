"""The N > 1 path on the REAL engine: two processes (one rank each, torch.distributed over gloo -- the GPU box has one
GPU, so both ranks drive cuda:0; RCCL itself needs one GPU per rank and is exercised by the driver's multi-GPU bench) run
speakerguard_amd.shard.ShardedAttack (batch of utterances cut over the ranks) and QueryShardedModel (rows of every
model call cut over the ranks) and must reproduce the single-process result bit for bit.  Small utterances and batches on
purpose: every contraction runs as an ordinary tile launch, none as a chip-wide persistent one, so the two processes
can share the GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _setup():
    from speakerguard_amd import synth
    from speakerguard_amd.model.xv_plda import xv_plda
    dev = torch.device("cuda:0")
    model = xv_plda.from_weights(synth.make_xv_weights(), device=dev, dither=1.0, dither_seed=9)  # random front-end on
    x = torch.from_numpy(synth.make_waveforms(4, 16000, seed=51)).to(dev)
    y = (torch.arange(4) % 10).to(dev)
    return model, x, y


def _pgd(model):
    from speakerguard_amd.attack.PGD import PGD
    return PGD(model, epsilon=0.002, step_size=0.0005, max_iter=3, batch_size=2, EOT_size=2, EOT_batch_size=2, verbose=0)


def _fakebob(model):
    from speakerguard_amd.attack.FAKEBOB import FAKEBOB
    return FAKEBOB(model, task="CSI", epsilon=0.002, max_iter=2, samples_per_draw=6, samples_per_draw_batch_size=6, batch_size=2,
                   stop_early=False, verbose=0)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from speakerguard_amd.shard import QueryShardedModel, ShardedAttack
    model, x, y = _setup()
    adv, succ = ShardedAttack(_pgd(model)).attack(x, y)
    model._noise_epoch = 0
    qadv, qsucc = _fakebob(QueryShardedModel(model)).attack(x[:2], y[:2])
    torch.save({"adv": adv.cpu(), "succ": succ, "qadv": qadv.cpu(), "qsucc": qsucc}, out % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_processes_on_the_engine_match_one(tmp_path):
    model, x, y = _setup()
    ref_adv, ref_succ = _pgd(model).attack(x, y)
    model._noise_epoch = 0
    ref_qadv, ref_qsucc = _fakebob(model).attack(x[:2], y[:2])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank%d.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    for rank in (0, 1):
        got = torch.load(out % rank)
        assert got["succ"] == list(ref_succ) and torch.equal(got["adv"], ref_adv.cpu()), rank
        assert got["qsucc"] == list(ref_qsucc) and torch.equal(got["qadv"], ref_qadv.cpu()), rank


def _rccl_worker(rank, port, out):
    """One rank over the `nccl` backend (= RCCL): communicator creation on the GPU box, then every collective the product
    issues -- ShardedAttack's flag all-gather and restart all-reduce, QueryShardedModel's score / gradient all-gathers,
    bench.py's uint8 flag exchange and barrier -- on DEVICE tensors.  A single GPU admits a single RCCL rank, so the
    exchange itself is degenerate; what this pins is that the calls, dtypes (RCCL has no bool) and tensor placement the
    N > 1 path uses are accepted by RCCL."""
    import sys
    from conftest import ROOT
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=1, device_id=torch.device("cuda", 0))
    from speakerguard_amd.attack.PGD import PGD
    from speakerguard_amd.shard import QueryShardedModel, ShardedAttack
    model, x, y = _setup()
    adv, succ = ShardedAttack(_pgd(model)).attack(x, y)
    restarts = PGD(model, epsilon=0.002, step_size=0.0005, max_iter=2, num_random_init=2, batch_size=2, verbose=0)
    import numpy as np
    np.random.seed(5)
    radv, rsucc = ShardedAttack(restarts).attack(x, y)
    model._noise_epoch = 0
    qadv, qsucc = _fakebob(QueryShardedModel(model)).attack(x[:2], y[:2])
    sys.path.insert(0, ROOT)
    import bench
    flags = bench.gather_flags(torch.tensor(list(succ), device="cuda:0"), dist, 1)
    (k, _), dt = bench.timed_region(lambda k: (k, None), 2, 1, dist, torch.cuda.synchronize, torch.device("cuda", 0))
    torch.save({"adv": adv.cpu(), "succ": succ, "radv": radv.cpu(), "rsucc": rsucc, "qadv": qadv.cpu(), "qsucc": qsucc,
                "flags": flags.cpu(), "backend": dist.get_backend(), "k": k, "dt": dt}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rccl_backend_runs_the_exchange_on_device_tensors(tmp_path):
    import numpy as np
    from speakerguard_amd.attack.PGD import PGD
    model, x, y = _setup()
    ref_adv, ref_succ = _pgd(model).attack(x, y)
    np.random.seed(5)
    ref_radv, ref_rsucc = PGD(model, epsilon=0.002, step_size=0.0005, max_iter=2, num_random_init=2, batch_size=2,
                              verbose=0).attack(x, y)
    model._noise_epoch = 0
    ref_qadv, ref_qsucc = _fakebob(model).attack(x[:2], y[:2])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rccl.pt")
    mp.spawn(_rccl_worker, args=(port, out), nprocs=1, join=True)
    got = torch.load(out)
    assert got["backend"] == "nccl"
    assert got["succ"] == list(ref_succ) and torch.equal(got["adv"], ref_adv.cpu())
    assert got["rsucc"] == list(ref_rsucc) and torch.equal(got["radv"], ref_radv.cpu())
    assert got["qsucc"] == list(ref_qsucc) and torch.equal(got["qadv"], ref_qadv.cpu())
    assert got["flags"].tolist() == [int(v) for v in ref_succ] and got["flags"].dtype == torch.uint8
    assert got["k"] == 2 and got["dt"] >= 0


@pytest.mark.timeout(900)
def test_bench_n2_code_path_on_one_gpu(tmp_path):
    """bench.py's N > 1 path end to end -- the launcher contract, ShardedAttack(PGD(batch_size=64)) as the strong partition,
    the weak partition, MAX-over-ranks timing, the one JSON line from rank 0 -- as two ranks over gloo that share the test
    box's one GPU (hidden --backend / --one-gpu hooks; RCCL needs a GPU per rank: the driver's SCALE run).  Two steps, and
    SG_STREAMK=0: two processes must not both run chip-wide persistent kernels with intra-launch hand-offs on one GPU."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SG_TUNE="1", SG_STREAMK="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reps", "2",
           "--backend", "gloo", "--one-gpu"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 128
    ss = line["strong_scaling"]
    assert ss["scaling"] == "strong" and ss["batch_per_gpu"] == 32 and "ShardedAttack" in ss["note"]
    assert line["value_metric_partition"] == ss["value"] and 0 <= ss["success_count"] <= 64
    assert line["success_count"] <= 128 and line["roofline"]["frac"] > 0
