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
  * CW2's early stop uses the mean loss of the chunk it is processing (attack/CW2.py:96-100); shards
    are cut on multiples of ``attacker.batch_size`` so chunks are the same ones the unsharded run uses;
  * random restarts draw the FULL (N,1,T) noise from the host RNG on every rank (seed all ranks
    alike) and slice it, so the noise an utterance sees does not depend on the shard layout;
  * device-generated noise (MFCC dither, NES queries) is keyed by (seed, attack call, restart, GLOBAL index of
    the chunk's first utterance, pass number within the chunk) and the row within the chunk
    (model/_engine_ops.py): a rank passes its shard offset as ``attacker.index_offset``, so a sharded run
    draws exactly the noise of the unsharded run (tests/test_gpu_xv.py::test_device_noise_is_shard_invariant).

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


class ShardedAttack:
    """Wraps any attack object exposing ``attack(x, y) -> (adver_x, success)``."""

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

    def _local_attack(self, x, y, lo, hi):
        a = self.attacker
        a.index_offset = lo  # device-generated noise (dither, NES) is keyed by the GLOBAL chunk position
        try:
            return self._local_attack_at(x, y, lo, hi)
        finally:
            a.index_offset = 0

    def _local_attack_at(self, x, y, lo, hi):
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
                if hi > lo:
                    adv, succ = a._run_batches(xi[lo:hi], y[lo:hi], lower[lo:hi], upper[lo:hi], tag=init)
                else:
                    adv, succ = xi[lo:hi], []
                cnt = torch.tensor([float(sum(succ))], device=x.device if x.is_cuda else "cpu")
                if world > 1:
                    dist.all_reduce(cnt, group=self.group)
                rate = float(cnt.item()) / x.shape[0]
                if rate > best_rate:
                    best_rate, best = rate, (adv, succ)
            return best
        if hi > lo:
            return a.attack(x[lo:hi], y[lo:hi])
        base = getattr(getattr(a, "model", None), "base_model", getattr(a, "model", None))
        if hasattr(base, "begin_attack"):  # an empty shard still counts the attack call: noise keys stay aligned over ranks
            base.begin_attack()
        return x[lo:hi], []

    def attack(self, x, y):
        """x (N,1,T), y (N,) identical on every rank -> (adver_x, success list of length N).

        With ``gather_audio=False`` the returned audio is this rank's shard only (the 12 MB all-gather
        of a 64 x 3 s batch is skipped); success flags are always global."""
        world, rank = self._world()
        n = x.shape[0]
        bounds = shard_bounds(n, world, max(1, getattr(self.attacker, "batch_size", 1)))
        lo, hi = bounds[rank]
        adv, succ = self._local_attack(x, y, lo, hi)
        if world == 1:
            return adv, list(succ)
        flags = torch.tensor([bool(s) for s in succ], dtype=torch.uint8, device=x.device)
        all_flags = self._gather_rows(flags, bounds, n)
        if self.gather_audio:
            adv = self._gather_rows(adv.contiguous(), bounds, n)
        return adv, [bool(v) for v in all_flags.tolist()]
