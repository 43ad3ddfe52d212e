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
    model = AutogradEngine(toy, per_row=True)  # an utterance's arithmetic must not depend on its batch, as on the engine
    if kind == "pgd64":  # the metric's shape: ONE chunk of 64 ("batch 64 sharded B/G per GPU")
        return PGD(model, epsilon=0.01, step_size=0.002, max_iter=4, batch_size=64, verbose=0)
    if kind == "cw2_64":  # one chunk whose early stop is a mean over utterances that sit on different ranks
        return CW2(model, initial_const=0.5, binary_search_steps=2, max_iter=12, stop_early=True, stop_early_iter=3,
                   lr=5e-3, batch_size=64, verbose=0)
    if kind == "pgd":
        return PGD(model, epsilon=0.01, step_size=0.002, max_iter=6, batch_size=2, verbose=0)
    if kind == "pgd_rand":
        return PGD(model, epsilon=0.01, step_size=0.002, max_iter=3, batch_size=2, num_random_init=3, verbose=0)
    return CW2(model, initial_const=0.5, binary_search_steps=2, max_iter=12, stop_early=True, stop_early_iter=5,
               lr=5e-3, batch_size=2, verbose=0)


def _data(n=6):
    x = toy_inputs(B=n, T=800, seed=5)
    toy = ToyModel().eval()
    with torch.no_grad():
        y = toy.make_decision(x)[0]
    return x, y


class _Recording:
    """Attacker proxy that notes which utterances this rank was handed (chunk by chunk)."""

    def __init__(self, attacker):
        object.__setattr__(self, "_a", attacker)
        object.__setattr__(self, "seen", [])
        inner = attacker.attack_batch

        def attack_batch(x, y, lower, upper, batch_id):
            self.seen.append(int(x.shape[0]))
            return inner(x, y, lower, upper, batch_id)
        attacker.attack_batch = attack_batch

    def __getattr__(self, name):
        return getattr(self._a, name)

    def __setattr__(self, name, value):
        setattr(self._a, name, value)


def _worker(rank, world, port, kind, out, n=6):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    x, y = _data(n)
    np.random.seed(77)
    rec = _Recording(_make(kind))
    adv, succ = ShardedAttack(rec).attack(x, y)
    chunks = [None] * world
    dist.all_gather_object(chunks, rec.seen)
    if rank == 0:
        torch.save({"adv": adv, "succ": succ, "chunks": chunks}, out)
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


def test_metric_configuration_is_cut_over_all_ranks(tmp_path):
    """N = 64 with batch_size = 64 (what INTEGRATION.md tells users to pass, and BASELINE.json's metric) on 8 and on 3
    ranks: every rank attacks its 64 / world utterances as ONE chunk, and the result is the unsharded one bit for
    bit.  Round 3 cut on multiples of batch_size, which handed rank 0 the whole batch."""
    assert ShardedAttack(_make("pgd64")).granule() == 1 and ShardedAttack(_make("cw2_64")).granule() == 1
    x, y = _data(64)
    for kind, worlds in (("pgd64", (8, 3)), ("cw2_64", (3,))):
        np.random.seed(77)
        ref_adv, ref_succ = _make(kind).attack(x, y)
        for world in worlds:
            out = str(tmp_path / ("%s_%d.pt" % (kind, world)))
            mp.spawn(_worker, args=(world, _free_port(), kind, out, 64), nprocs=world, join=True)
            got = torch.load(out)
            sizes = [e - s for s, e in shard_bounds(64, world)]
            assert [sum(c) for c in got["chunks"]] == sizes and max(sizes) - min(sizes) <= 1, (kind, world, got["chunks"])
            assert all(len(c) == 1 for c in got["chunks"]), "a rank's shard runs as one chunk of min(batch_size, shard)"
            assert got["succ"] == list(ref_succ) and len(got["succ"]) == 64, (kind, world)
            assert torch.equal(got["adv"], ref_adv), (kind, world)
        assert any(ref_succ) and not all(ref_succ), "both outcomes of the success predicate: %s" % kind


def test_fakebob_keeps_its_chunks_together():
    """FAKEBOB's plateau history aliases over the examples of a chunk (attack/FAKEBOB.py:56): the batch cut stays on
    chunk boundaries; what scales it is QueryShardedModel (below)."""
    assert ShardedAttack(FAKEBOB(_toy_engine(), batch_size=4, verbose=0)).granule() == 4
    assert shard_bounds(10, 2, 4) == [(0, 8), (8, 10)]


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


# ---- batch-coupled defense (FeCo, reference defense/feature_level.py:33): a one-utterance model call behaves differently ----
class _CoupledEngine(AutogradEngine):
    """CPU stand-in for a FeCo-defended model: a model call that holds ONE utterance scores differently from a larger one
    (the reference drops empty clusters only when feat.shape[0] == 1)."""
    batch_coupled = True

    def _md(self, x):
        dec, sc = super()._md(x)
        if x.shape[0] == 1:
            sc = sc * 0.5 + 0.25 * sc.flip(1)
            dec = torch.argmax(sc, dim=1)
        return dec, sc


def _coupled_worker(rank, world, port, n, bs, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    x, y = _data(n)
    rec = _Recording(_coupled_attack(bs))
    adv, succ = ShardedAttack(rec).attack(x, y)
    chunks = [None] * world
    dist.all_gather_object(chunks, rec.seen)
    if rank == 0:
        torch.save({"adv": adv, "succ": succ, "chunks": chunks}, out)
    dist.barrier()
    dist.destroy_process_group()


def _coupled_attack(bs):
    toy = ToyModel().eval()
    for p in toy.parameters():
        p.requires_grad_(False)
    return PGD(_CoupledEngine(toy, per_row=True), epsilon=0.01, step_size=0.002, max_iter=4, batch_size=bs, verbose=0)


def test_batch_coupled_model_is_cut_without_new_single_utterance_calls(tmp_path):
    """with a batch-coupled model the cut must not change which utterances sit alone in a model call
    (shard.coupled_plan), over real gloo ranks: whole chunks per rank (5 utterances, batch_size 2 / 4 on 2 ranks), a cut inside
    the one chunk (9 utterances, batch_size 64), and the unsharded run's trailing one-utterance call kept as a second call
    on the last busy rank with an idle rank behind it (5 utterances, batch_size 4, 3 ranks)."""
    for n, bs, world, want in ((5, 2, 2, [[2, 2], [1]]), (5, 4, 2, [[4], [1]]), (9, 64, 2, [[5], [4]]), (5, 4, 3, [[2], [2, 1], []])):
        x, y = _data(n)
        ref_adv, ref_succ = _coupled_attack(bs).attack(x, y)
        plain_adv, _ = PGD(AutogradEngine(ToyModel().eval(), per_row=True), epsilon=0.01, step_size=0.002, max_iter=4, batch_size=bs,
                           verbose=0).attack(x, y)
        out = str(tmp_path / ("coupled_%d_%d_%d.pt" % (n, bs, world)))
        mp.spawn(_coupled_worker, args=(world, _free_port(), n, bs, out), nprocs=world, join=True)
        got = torch.load(out)
        assert got["chunks"] == want, (n, bs, world, got["chunks"])
        assert got["succ"] == list(ref_succ) and torch.equal(got["adv"], ref_adv), (n, bs, world)
        if n % bs == 1:  # the coupling is real: the lone utterance's result differs from the uncoupled model's
            assert not torch.equal(ref_adv[-1], plain_adv[-1])


def _coupled_cw2(bs):
    toy = ToyModel().eval()
    for p in toy.parameters():
        p.requires_grad_(False)
    return CW2(_CoupledEngine(toy, per_row=True), initial_const=0.5, binary_search_steps=2, max_iter=9, stop_early=True,
               stop_early_iter=3, lr=5e-3, batch_size=bs, verbose=0)


def _coupled_cw2_worker(rank, world, port, n, bs, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    x, y = _data(n)
    rec = _Recording(_coupled_cw2(bs))
    adv, succ = ShardedAttack(rec).attack(x, y)
    chunks = [None] * world
    dist.all_gather_object(chunks, rec.seen)
    if rank == 0:
        torch.save({"adv": adv, "succ": succ, "chunks": chunks}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_batch_coupled_model_under_the_mean_coupled_cut(tmp_path):
    """CW2 with stop_early takes the mean-coupled path (every chunk cut over the ranks), which used to hand single
    utterances of a larger chunk to ranks -- a FeCo-defended model then flips `force = feat.shape[0] > 1`.  With
    row_slices_coupled every call of a chunk of >= 2 holds >= 2 utterances (5 utterances on 3 ranks: 3 + 2 + a dropped re-run of
    two), a chunk of one stays a call of one, and the sharded attack equals the unsharded one."""
    from speakerguard_amd.shard import row_slices_coupled
    assert row_slices_coupled(5, 3) == ([(0, 3), (3, 5), (0, 2)], [(0, 3), (3, 5), (5, 5)])
    assert row_slices_coupled(1, 2) == ([(0, 1), (0, 1)], [(0, 1), (1, 1)])
    assert row_slices_coupled(4, 2) == ([(0, 2), (2, 4)], [(0, 2), (2, 4)])
    for n, bs, world, want in ((5, 64, 3, [[3], [2], [2]]), (5, 4, 2, [[2, 1], [2, 1]])):
        x, y = _data(n)
        ref_adv, ref_succ = _coupled_cw2(bs).attack(x, y)
        out = str(tmp_path / ("coupled_cw2_%d_%d_%d.pt" % (n, bs, world)))
        mp.spawn(_coupled_cw2_worker, args=(world, _free_port(), n, bs, out), nprocs=world, join=True)
        got = torch.load(out)
        assert got["chunks"] == want, (n, bs, world, got["chunks"])
        assert got["succ"] == list(ref_succ) and torch.allclose(got["adv"], ref_adv, atol=0, rtol=0), (n, bs, world)
