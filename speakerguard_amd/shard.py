"""Batch-shard data parallelism for the attacks (one process per GPU, torch.distributed).

Every example's attack trajectory is independent (per-example loss vector, eval-mode model, no
cross-example op: reference adaptive_attack/EOT.py:33-35), so a batch is split into contiguous
slices, one per rank, with NO collective on the data path.  The only exchanges are:

  * one all-gather of the per-utterance success flags (and optionally the adversarial audio) at
    the end of ``attack``;
  * for PGD with ``num_random_init > 0`` one scalar all-reduce per restart, because the reference
    keeps the restart with the best WHOLE-BATCH success rate (attack/PGD.py:74-77).

Batch-coupled details that are preserved by construction rather than by communication:
  * ``check_input_range`` (model/utils.py:11) decides the int16 rescale from the batch max/min; attack
    inputs are asserted to be in [-1, 1), so every shard takes the same branch;
  * CW2's early stop uses the mean loss of the chunk it is processing (attack/CW2.py:96-100): with ``stop_early``
    the cut goes INSIDE every chunk (each rank iterates its slice of the chunk) and the per-utterance losses of the
    chunk are all-gathered at every early-stop test -- B floats every ``stop_early_iter`` iterations, the path's
    one exchange inside a loop -- so every rank takes the mean over the same values in the same order as the
    unsharded run and stops at the same iteration;
  * FAKEBOB's plateau history leaks between the examples of a chunk (the reference's ``[[]] * n`` aliasing,
    attack/FAKEBOB.py:56, reproduced on purpose), so its batch cut stays on multiples of ``batch_size``; the cut
    that scales it is ``QueryShardedModel`` below;
  * everything else (FGSM / PGD / CWinf, CW2 without early stop) has NO coupling inside a chunk: the batch is cut
    with granule 1 -- N = 64 with ``batch_size=64`` on 8 ranks is 8 utterances per rank ("batch
    64 sharded B/G per GPU"; reference attack/PGD.py:62-73 only chunks for memory) -- and a rank runs its shard in
    chunks of ``min(batch_size, shard)``.  Results do not depend on the chunking: per-utterance arithmetic is
    independent of batch composition (bit for bit on the engine), noise is keyed by the global utterance index;
  * a model with a BATCH-COUPLED defense (FeCo: feature_level.py:33, `force = feat.shape[0] > 1` -- a one-utterance model
    call drops empty clusters, a larger one fills them in; ``model.batch_coupled``) is cut so that no cut changes which
    utterances sit alone in a model call: on multiples of ``batch_size`` when there are at least as many chunks as ranks,
    otherwise inside the chunks with at least two utterances per rank, and a trailing one-utterance chunk of the unsharded
    run (N mod batch_size == 1) stays a call of its own on the rank that holds it (``coupled_plan``);
  * random restarts draw the FULL (N,1,T) noise from the host RNG on every rank (seed all ranks
    alike) and slice it, so the noise an utterance sees does not depend on the shard layout;
  * device-generated noise (MFCC dither, NES queries, FeCo's random start) is keyed by (seed, attack call, restart,
    pass number within the chunk) and, inside the kernels, by the GLOBAL index of the utterance and the EOT repeat
    (model/_engine_ops.py): a rank passes its shard offset as ``attacker.index_offset``, so a sharded run draws
    exactly the noise of the unsharded run wherever the cut falls
    (tests/test_gpu_xv.py::test_device_noise_is_shard_invariant).

``QueryShardedModel`` is the other cut (BASELINE.json configs[4]: "query batch sharded over 8 x MI355X"): the
black-box attacks score n * (samples_per_draw + 1) perturbed copies of FEW utterances per iteration
(adaptive_attack/NES.py:19-34), so the ROWS OF ONE MODEL CALL are split over the ranks and the per-row results
(decision, scores, loss -- and the gradient when one is asked for) are all-gathered: one real exchange per model
call, a few hundred bytes per query.  Everything around the call (query construction, the loss-weighted noise
average, the FAKEBOB step) is cheap elementwise work that every rank repeats on identical data, so the attack
state stays replicated without further communication.

Backend: ``nccl`` (= RCCL over xGMI) on GPUs, ``gloo`` in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n, world, granule=1):
    """Contiguous [start, end) per rank, sizes multiples of `granule` except the tail."""
    units = (n + granule - 1) // granule
    base, extra = divmod(units, world)
    bounds, s = [], 0
    for r in range(world):
        e = min(n, s + (base + (1 if r < extra else 0)) * granule)
        bounds.append((s, e))
        s = e
    return bounds


def coupled_plan(n, world, batch_size):
    """Per rank the list of [start, end) ranges (each one ``attack`` call) for a model whose defense depends on the number
    of utterances in a model call (see the module docstring).  The unsharded run makes calls of ``batch_size`` utterances
    and one of ``n % batch_size``; a call has ONE utterance only if batch_size == 1 or n % batch_size == 1."""
    bs = max(1, min(batch_size, n))
    chunks = (n + bs - 1) // bs
    if bs == 1 or chunks >= world:  # whole chunks per rank: every model call is one of the unsharded run's calls
        return [[b] if b[1] > b[0] else [] for b in shard_bounds(n, world, bs)]
    tail = 1 if n % bs == 1 else 0       # the unsharded run's trailing one-utterance call
    body = n - tail
    active = max(1, min(world, body // 2))  # ranks that get utterances: at least two each (fewer ranks than chunks: every shard < bs)
    plan = [[b] if b[1] > b[0] else [] for b in shard_bounds(body, active)] + [[] for _ in range(world - active)]
    if tail:
        last = active - 1 if body > 0 else 0
        plan[last] = plan[last] + [(n - 1, n)]
    return plan


class ShardedAttack:
    """Wraps an attack object exposing ``attack(x, y) -> (adver_x, success)``.  An attack class states how its chunks couple
    their examples through ``chunk_coupling`` (None: not at all, 'mean': through a batch mean, 'chunk': otherwise); a class
    that does not say is treated as 'chunk' -- cut on multiples of its ``batch_size`` only."""

    def __init__(self, attacker, group=None, gather_audio=True):
        self.attacker = attacker
        self.group = group
        self.gather_audio = gather_audio

    def _world(self):
        if not (dist.is_available() and dist.is_initialized()):
            return 1, 0
        return dist.get_world_size(self.group), dist.get_rank(self.group)

    def _gather_rows(self, local, bounds, n):
        """all-gather of uneven leading-dim shards (padded to the largest)."""
        world, _ = self._world()
        width = max(e - s for s, e in bounds)
        pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=self.group)
        return torch.cat([p[: e - s] for p, (s, e) in zip(parts, bounds)], 0)

    def _local_attack(self, x, y, ranges):
        """This rank's part: one ``attack`` call per range (normally one range; a second one only for the trailing
        one-utterance call of a batch-coupled model).  Returns (adver of the ranges concatenated, success list)."""
        a = self.attacker
        ranges = [r for r in ranges if r[1] > r[0]]
        try:
            return self._local_attack_at(x, y, ranges)
        finally:
            a.index_offset = 0

    def _local_attack_at(self, x, y, ranges):
        a = self.attacker
        restarts = getattr(a, "num_random_init", 0)
        if restarts and restarts > 0:
            if hasattr(a, "_begin_attack"):
                a._begin_attack()
            # PGD.attack (:48-77) with the best-of-restarts criterion evaluated on the whole batch
            world, _ = self._world()
            upper = torch.clamp(x + a.epsilon, max=1)
            lower = torch.clamp(x - a.epsilon, min=-1)
            best_rate, best = -1.0, None
            for init in range(restarts):
                noise = torch.tensor(np.random.uniform(-a.epsilon, a.epsilon, tuple(x.shape)), device=x.device, dtype=x.dtype)
                xi = x + noise
                advs, succ = [], []
                for lo, hi in ranges:
                    a.index_offset = lo  # device-generated noise (dither, NES) is keyed by the GLOBAL chunk position
                    adv_r, succ_r = a._run_batches(xi[lo:hi], y[lo:hi], lower[lo:hi], upper[lo:hi], tag=init)
                    advs.append(adv_r)
                    succ += succ_r
                adv = torch.cat(advs, 0) if advs else xi[0:0]
                cnt = torch.tensor([float(sum(succ))], device=x.device if x.is_cuda else "cpu")
                if world > 1:
                    dist.all_reduce(cnt, group=self.group)
                rate = float(cnt.item()) / x.shape[0]
                if rate > best_rate:
                    best_rate, best = rate, (adv, succ)
            return best
        if not ranges:
            base = getattr(getattr(a, "model", None), "base_model", getattr(a, "model", None))
            if hasattr(base, "begin_attack"):  # an empty shard still counts the attack call: noise keys stay aligned over ranks
                base.begin_attack()
            return x[0:0], []
        advs, succ = [], []
        for i, (lo, hi) in enumerate(ranges):
            a.index_offset = lo
            if i > 0 and hasattr(a, "_begin_attack"):
                a._begin_attack = lambda: None  # the rank's ranges are ONE attack call of the unsharded run: counted once
            try:
                adv_r, succ_r = a.attack(x[lo:hi], y[lo:hi])
            finally:
                if i > 0 and "_begin_attack" in vars(a):
                    del a._begin_attack
            advs.append(adv_r)
            succ += list(succ_r)
        return torch.cat(advs, 0), succ

    def granule(self):
        """Utterances that must stay together on one rank: 1 if the attack declares that nothing couples the examples of a
        chunk (``chunk_coupling = None``: FGSM / PGD / CWinf; 'mean' takes its own path), otherwise -- 'chunk' (FAKEBOB) or an
        attack object that does not declare anything -- its ``batch_size``."""
        a = self.attacker
        if getattr(a, "chunk_coupling", "chunk") in (None, "mean"):
            return 1
        return max(1, getattr(a, "batch_size", 1))

    def _batch_coupled(self):
        return bool(getattr(getattr(self.attacker, "model", None), "batch_coupled", False))

    def plan(self, n):
        """Per rank the [start, end) ranges it attacks (each range one ``attack`` call; contiguous and ascending over ranks)."""
        world, _ = self._world()
        if self._batch_coupled() and self.granule() == 1:
            return coupled_plan(n, world, max(1, getattr(self.attacker, "batch_size", 1)))
        return [[b] if b[1] > b[0] else [] for b in shard_bounds(n, world, self.granule())]

    def bounds(self, n):
        """[start, end) of every rank's contiguous shard of a batch of n utterances."""
        out, s = [], 0
        for ranges in self.plan(n):
            e = ranges[-1][1] if ranges else s
            out.append((ranges[0][0] if ranges else s, e))
            s = e
        return out

    def attack(self, x, y):
        """x (N,1,T), y (N,) identical on every rank -> (adver_x, success list of length N).

        With ``gather_audio=False`` the returned audio is this rank's shard only (the 12 MB all-gather
        of a 64 x 3 s batch is skipped); success flags are always global."""
        world, rank = self._world()
        n = x.shape[0]
        if world > 1 and getattr(self.attacker, "chunk_coupling", None) == "mean":
            return self._attack_mean_coupled(x, y)
        plan = self.plan(n)
        bounds = self.bounds(n)
        adv, succ = self._local_attack(x, y, plan[rank])
        if world == 1:
            return adv, list(succ)
        flags = torch.tensor([bool(s) for s in succ], dtype=torch.uint8, device=x.device)
        all_flags = self._gather_rows(flags, bounds, n)
        if self.gather_audio:
            adv = self._gather_rows(adv.contiguous(), bounds, n)
        return adv, [bool(v) for v in all_flags.tolist()]

    def _attack_mean_coupled(self, x, y):
        """An attack whose chunks are coupled through a batch mean only (CW2 with ``stop_early``, attack/CW2.py:96-100):
        every chunk of ``batch_size`` utterances is cut over the ranks; the attacker's ``batch_mean`` hook all-gathers
        the chunk's per-utterance losses and takes the mean over the full chunk in the unsharded order, so all ranks
        stop where the unsharded run stops.  A rank without utterances in a chunk (chunk smaller than the world)
        re-runs utterance 0 of the chunk and drops the result: every rank makes the same sequence of exchanges.
        With a batch-coupled model (``model.batch_coupled``: FeCo's `force = feat.shape[0] > 1`) a chunk of two or more
        utterances is cut so that every rank's call holds at least two (``row_slices_coupled``), and a one-utterance chunk
        stays a one-utterance call everywhere."""
        a = self.attacker
        world, rank = self._world()
        a._check_inputs(x, y)
        a._begin_attack()
        n = x.shape[0]
        lower = torch.tensor(-1, device=x.device, dtype=x.dtype).expand_as(x)
        upper = torch.tensor(1, device=x.device, dtype=x.dtype).expand_as(x)
        bs = min(max(1, a.batch_size), n)
        adver, success = [], []
        try:
            for batch_id, s in enumerate(range(0, n, bs)):
                e = min(n, s + bs)
                # a batch-coupled model (FeCo) must not see one-utterance calls the unsharded run does not make
                run, keep = row_slices_coupled(e - s, world) if self._batch_coupled() else row_slices(e - s, world)
                lo, hi = s + run[rank][0], s + run[rank][1]

                def mean_hook(loss, run=run, keep=keep):
                    width = max(b - a_ for a_, b in run)
                    pad = torch.zeros(width, dtype=loss.dtype, device=loss.device)
                    pad[: loss.shape[0]] = loss
                    parts = [torch.empty_like(pad) for _ in range(world)]
                    dist.all_gather(parts, pad, group=self.group)
                    return float(torch.cat([p[: b - a_] for p, (a_, b) in zip(parts, keep)]).mean().item())

                a.batch_mean = mean_hook
                a._begin_batch(lo)
                adv_c, succ_c = a.attack_batch(x[lo:hi], y[lo:hi], lower[lo:hi], upper[lo:hi], batch_id)
                mine = keep[rank][1] - keep[rank][0]  # 0: this rank only kept the exchanges company
                flags = torch.tensor([bool(v) for v in succ_c][:mine], dtype=torch.uint8, device=x.device)
                success += self._gather_rows(flags, keep, e - s).tolist()
                adver.append(self._gather_rows(adv_c[:mine].contiguous(), keep, e - s) if self.gather_audio else adv_c[:mine])
        finally:
            a.batch_mean = None
        base = getattr(getattr(a, "model", None), "base_model", getattr(a, "model", None))
        if hasattr(base, "check_health"):
            base.check_health()
        return torch.cat(adver, 0), [bool(v) for v in success]


def row_slices(n, world):
    """[start, end) of the rows of one model call per rank.  A rank that would get nothing (n < world) re-scores
    row 0 and its result is dropped: every rank makes the same sequence of model calls, so per-call state (the
    front-end's noise draw counter) stays aligned over ranks."""
    return [(s, e) if e > s else (0, 1) for s, e in shard_bounds(n, world)], shard_bounds(n, world)


def row_slices_coupled(n, world):
    """``row_slices`` for a batch-coupled model: the model behaves differently on a call that holds ONE utterance, so a chunk
    of n >= 2 utterances goes to min(world, n // 2) ranks with at least two each; the other ranks re-run rows 0 and 1 (a call of
    two, like everybody's) and drop the result.  A one-utterance chunk is a one-utterance call in the unsharded run too:
    plain ``row_slices``."""
    if n < 2:
        return row_slices(n, world)
    active = min(world, n // 2)
    keep = shard_bounds(n, active) + [(n, n)] * (world - active)
    return [(s, e) if e > s else (0, 2) for s, e in keep], keep


class QueryShardedModel:
    """Model proxy that splits the rows of every ``loss_grad`` call over the ranks of `group` and all-gathers the
    per-row results.  Use it as the model of a black-box attack -- ``FAKEBOB(QueryShardedModel(model), ...)`` -- with
    identical (x, y) on every rank: each rank returns the full result.  Row r of a call is computed exactly as in
    the unsharded call (per-row arithmetic does not depend on batch composition; device noise is keyed by the
    row's position in the FULL call through ``_row_base``), so the sharded attack reproduces the single-GPU one
    bit for bit (tests/test_shard_gloo.py on the CPU double, tests/test_gpu_xv.py with emulated ranks)."""

    def __init__(self, model, group=None):
        self.model = model
        self.base_model = getattr(model, 'base_model', model)
        self.group = group

    def __getattr__(self, name):  # threshold, make_decision, nes_queries, fakebob_step, ...: the wrapped model's
        if name in ('model', 'base_model', 'group'):  # not set yet (copy / unpickle): no recursion through self.model
            raise AttributeError(name)
        return getattr(self.model, name)

    def eval(self):
        return self

    def _world(self):
        if not (dist.is_available() and dist.is_initialized()):
            return 1, 0
        return dist.get_world_size(self.group), dist.get_rank(self.group)

    def _call_rows(self, x, y, loss_spec, lo, hi, want_grad, kw):
        base = self.base_model
        had = hasattr(base, '_row_base')
        if had:
            base._row_base = lo
        try:
            return self.model.loss_grad(x[lo:hi], y[lo:hi], loss_spec, want_grad=want_grad, **kw)
        finally:
            if had:
                base._row_base = 0

    def loss_grad(self, x, y, loss_spec, want_grad=True, **kw):
        world, rank = self._world()
        if world == 1:
            return self.model.loss_grad(x, y, loss_spec, want_grad=want_grad, **kw)
        n = x.shape[0]
        run, keep = row_slices(n, world)
        dec, scores, loss, grad = self._call_rows(x, y, loss_spec, run[rank][0], run[rank][1], want_grad, kw)
        # one exchange for (decision, loss, scores): packed as float32 columns (decisions are small integers)
        S = scores.shape[1]
        width = max(e - s for s, e in run)
        pack = torch.zeros(width, S + 2, dtype=torch.float32, device=scores.device)
        m = dec.shape[0]
        pack[:m, 0] = dec.to(torch.float32)
        pack[:m, 1] = loss
        pack[:m, 2:] = scores
        parts = [torch.empty_like(pack) for _ in range(world)]
        dist.all_gather(parts, pack, group=self.group)
        full = torch.cat([p[: e - s] for p, (s, e) in zip(parts, keep)], 0)
        out_dec = full[:, 0].round().to(torch.int64)
        out_loss = full[:, 1].contiguous()
        out_scores = full[:, 2:].contiguous()
        out_grad = None
        if want_grad:
            gpad = torch.zeros((width,) + tuple(grad.shape[1:]), dtype=grad.dtype, device=grad.device)
            gpad[:m] = grad
            gparts = [torch.empty_like(gpad) for _ in range(world)]
            dist.all_gather(gparts, gpad, group=self.group)
            out_grad = torch.cat([p[: e - s] for p, (s, e) in zip(gparts, keep)], 0)
        return out_dec, out_scores, out_loss, out_grad
