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
