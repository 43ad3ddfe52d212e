"""N > 1 path on CPU: world_size-2 gloo processes run speakerguard_amd.shard.ShardedAttack over the
CPU engine double and must reproduce the single-process attack exactly (shard invariance)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from engine_doubles import AutogradEngine
from speakerguard_amd.attack.CW2 import CW2
from speakerguard_amd.attack.FAKEBOB import FAKEBOB
from speakerguard_amd.attack.PGD import PGD
from speakerguard_amd.shard import QueryShardedModel, ShardedAttack, row_slices, shard_bounds
from toy_model import ToyModel, toy_inputs


def _make(kind):
    toy = ToyModel().eval()
    for p in toy.parameters():
        p.requires_grad_(False)
    model = AutogradEngine(toy)
    if kind == "pgd":
        return PGD(model, epsilon=0.01, step_size=0.002, max_iter=6, batch_size=2, verbose=0)
    if kind == "pgd_rand":
        return PGD(model, epsilon=0.01, step_size=0.002, max_iter=3, batch_size=2, num_random_init=3, verbose=0)
    return CW2(model, initial_const=0.5, binary_search_steps=2, max_iter=12, stop_early=True, stop_early_iter=5,
               lr=5e-3, batch_size=2, verbose=0)


def _data():
    x = toy_inputs(B=6, T=800, seed=5)
    toy = ToyModel().eval()
    with torch.no_grad():
        y = toy.make_decision(x)[0]
    return x, y


def _worker(rank, world, port, kind, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    x, y = _data()
    np.random.seed(77)
    adv, succ = ShardedAttack(_make(kind)).attack(x, y)
    if rank == 0:
        torch.save({"adv": adv, "succ": succ}, out)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_bounds():
    assert shard_bounds(64, 8) == [(i * 8, i * 8 + 8) for i in range(8)]
    assert shard_bounds(6, 2, granule=2) == [(0, 4), (4, 6)]
    assert shard_bounds(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]


def test_two_rank_gloo_matches_single_process(tmp_path):
    for kind in ("pgd", "pgd_rand", "cw2"):
        x, y = _data()
        np.random.seed(77)
        ref_adv, ref_succ = _make(kind).attack(x, y)
        out = str(tmp_path / ("%s.pt" % kind))
        mp.spawn(_worker, args=(2, _free_port(), kind, out), nprocs=2, join=True)
        got = torch.load(out)
        assert got["succ"] == list(ref_succ), kind
        assert torch.equal(got["adv"], ref_adv), kind


# ---- query sharding (BASELINE.json configs[4]): the rows of every model call split over the ranks --------------
def _fakebob(model, n_queries):
    gen = torch.Generator().manual_seed(99)  # same NES noise on every rank: the attack state is replicated
    rn = lambda shape: torch.randn(shape, generator=gen)
    return FAKEBOB(model, task="CSI", epsilon=0.01, max_iter=5, max_lr=0.002, samples_per_draw=n_queries,
                   samples_per_draw_batch_size=n_queries, sigma=0.002, stop_early=False, batch_size=2, verbose=0,
                   noise_fn=rn)


def _toy_engine():
    toy = ToyModel().eval()
    for p in toy.parameters():
        p.requires_grad_(False)
    return AutogradEngine(toy)


def _query_worker(rank, world, port, n_queries, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    x, y = _data()
    x, y = x[:3], y[:3]
    model = QueryShardedModel(_toy_engine())
    adv, succ = _fakebob(model, n_queries).attack(x, y)
    # the white-box form of the same proxy: rows of a gradient call split over the ranks
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    dec, sc, ls, g = model.loss_grad(x, y, SEC4SR_CrossEntropy(), want_grad=True)
    torch.save({"adv": adv, "succ": succ, "dec": dec, "sc": sc, "ls": ls, "g": g}, out % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_row_slices_keep_every_rank_busy():
    run, keep = row_slices(5, 2)
    assert run == keep == [(0, 3), (3, 5)]
    run, keep = row_slices(1, 4)  # fewer rows than ranks: the idle ranks re-score row 0, nothing of theirs is kept
    assert run == [(0, 1)] * 4 and keep == [(0, 1), (1, 1), (1, 1), (1, 1)]


def test_query_sharded_fakebob_two_rank_gloo_matches_single_process(tmp_path):
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    for n_queries in (6, 4):  # 3 examples x 7 rows (uneven split) and x 5 rows
        x, y = _data()
        x, y = x[:3], y[:3]
        ref_adv, ref_succ = _fakebob(_toy_engine(), n_queries).attack(x, y)
        ref = _toy_engine().loss_grad(x, y, SEC4SR_CrossEntropy(), want_grad=True)
        out = str(tmp_path / ("q%d_rank%%d.pt" % n_queries))
        mp.spawn(_query_worker, args=(2, _free_port(), n_queries, out), nprocs=2, join=True)
        for rank in (0, 1):  # every rank ends with the full, identical result
            got = torch.load(out % rank)
            assert got["succ"] == list(ref_succ)
            assert torch.equal(got["adv"], ref_adv)
            # the CPU double's matmuls round differently for 3 rows and for 2 + 1 rows (the HIP engine does not:
            # tests/test_gpu_xv.py checks the sliced call bit for bit); rows, order and content must be the same
            assert torch.equal(got["dec"], ref[0])
            torch.testing.assert_close(got["sc"], ref[1], rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(got["ls"], ref[2], rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(got["g"], ref[3], rtol=1e-4, atol=1e-6)
