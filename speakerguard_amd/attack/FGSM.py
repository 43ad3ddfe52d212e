"""FGSM and the shared PGD inner loop; mirrors reference attack/FGSM.py.

``attack_batch`` is the hot loop (FGSM.py:38-70).  When the model is the native x-vector engine
and nothing between attack and model needs Python (no defense wrapper), the whole loop, EOT repeats
over the front-end's random dither included -- max_iter x (forward, hand-coded backward, sign step,
projection) + the final forward-only pass -- is ONE C-ABI call (``model.pgd_run``).  Otherwise
the same loop runs step by step over ``model.loss_grad`` / ``model.pgd_update``.
"""
import numpy as np
import torch

from ..adaptive_attack.EOT import EOT
from .Attack import Attack
from .utils import resolve_loss, resolve_prediction


class FGSM(Attack):

    def __init__(self, model, task='CSI', epsilon=0.002, loss='Entropy', targeted=False,
                 batch_size=1, EOT_size=1, EOT_batch_size=1,
                 verbose=1):
        self.model = model  # the engine has no train mode
        self.task = task
        self.epsilon = epsilon
        self.loss_name = loss
        self.targeted = targeted
        self.batch_size = batch_size
        self._init_common(EOT_size, EOT_batch_size, verbose)
        self.max_iter = 1  # FGSM is the single-step case of PGD (FGSM.py:35-36)
        self.step_size = epsilon

    def _init_common(self, EOT_size, EOT_batch_size, verbose):
        EOT_size = max(1, EOT_size)
        EOT_batch_size = max(1, EOT_batch_size)
        assert EOT_size % EOT_batch_size == 0, 'EOT size should be divisible by EOT batch size'
        self.EOT_size = EOT_size
        self.EOT_batch_size = EOT_batch_size
        self.verbose = verbose
        self.threshold = None
        if self.task in ['SV', 'OSI']:
            self.threshold = self.model.threshold
            print('Running white box attack for {} task, directly using the true threshold {}'.format(self.task, self.threshold))
        self.loss, self.grad_sign = resolve_loss(loss_name=self.loss_name, targeted=self.targeted,
                                                 task=self.task, threshold=self.threshold, clip_max=False)
        self.EOT_wrapper = EOT(self.model, self.loss, self.EOT_size, self.EOT_batch_size, True)

    # ---- fused device loop -----------------------------------------------------------------
    def _fused_feco(self, n_audios):
        """The FeCoDefense of ``defended_model(base, [(1, FeCoDefense)])`` when the base model runs the defended loop on
        the device (BASELINE.json configs[3]: audionet_csine.pgd_run_feco), else None."""
        m = self.model
        defense = getattr(m, 'defense', None)
        base = getattr(m, 'base_model', None)
        if not self.fuse_defended or defense is None or base is None or not hasattr(base, 'pgd_run_feco') or n_audios < 2:
            return None  # one utterance: the reference drops empty clusters (variable frame count) -> host path
        if getattr(m, 'order', None) != 'sequential' or len(defense) != 1:
            return None
        flag, d = defense[0]
        from ..defense.feature_level import FeCoDefense
        return d if flag == 1 and isinstance(d, FeCoDefense) else None

    def _can_fuse(self):
        m = self.model
        if getattr(m, 'defense', None) is not None:
            return False
        base = getattr(m, 'base_model', m)
        # EOT repeats of a deterministic model are identical (one pass stands for all of them); with random dither
        # the engine runs the repeats itself and sums their gradients on the device
        return hasattr(base, 'pgd_run')

    fuse_defended = True  # False: PGD against a FeCo-defended model runs the host-chained loop (tests compare the two)

    def _attack_batch_fused(self, x_batch, y_batch, lower, upper, batch_id, feco=None):
        base = getattr(self.model, 'base_model', self.model)
        if feco is not None:
            x_adv, success, dec, scores, loss, ltr, dtr = base.pgd_run_feco(
                x_batch, y_batch, lower, upper, self.loss, self.step_size, self.max_iter, self.grad_sign, feco,
                self.EOT_size, self.EOT_batch_size, trace=bool(self.verbose))
        else:
            x_adv, success, dec, scores, loss, ltr, dtr = base.pgd_run(
                x_batch, y_batch, lower, upper, self.loss, self.step_size, self.max_iter, self.grad_sign,
                self.EOT_size, self.EOT_batch_size, trace=bool(self.verbose))
        if self.verbose:
            ltr, dtr = ltr.cpu().numpy(), dtr.cpu().numpy()
            target = y_batch.detach().cpu().numpy()
            for it in range(self.max_iter + 1):
                print("batch:{} iter:{} loss: {} predict: {}, target: {}".format(batch_id, it, ltr[it].tolist(), dtr[it], target))
        return x_adv, [bool(v) for v in success.tolist()]  # (one device round trip, no conversion launch)

    # ---- step-by-step loop (FGSM.py:38-70) ---------------------------------------------------
    def attack_batch(self, x_batch, y_batch, lower, upper, batch_id):
        if self._can_fuse():
            return self._attack_batch_fused(x_batch, y_batch, lower, upper, batch_id)
        feco = self._fused_feco(x_batch.shape[0])
        if feco is not None:
            return self._attack_batch_fused(x_batch, y_batch, lower, upper, batch_id, feco=feco)
        x_batch = x_batch.clone()
        lower = lower.expand_as(x_batch).contiguous()
        upper = upper.expand_as(x_batch).contiguous()
        base = getattr(self.model, 'base_model', self.model)
        success = None
        for it in range(self.max_iter + 1):
            EOT_num_batches = int(self.EOT_size // self.EOT_batch_size) if it < self.max_iter else 1
            real_EOT_batch_size = self.EOT_batch_size if it < self.max_iter else 1
            use_grad = it < self.max_iter
            scores, loss, grad, decisions = self.EOT_wrapper(x_batch, y_batch, EOT_num_batches, real_EOT_batch_size, use_grad)
            loss = loss / EOT_num_batches
            predict = resolve_prediction(decisions)
            target = y_batch.detach().cpu().numpy()
            success = self.compare(target, predict, self.targeted)
            if self.verbose:
                print("batch:{} iter:{} loss: {} predict: {}, target: {}".format(batch_id, it, loss.cpu().numpy().tolist(), predict, target))
            if it < self.max_iter:
                grad = (grad / EOT_num_batches).contiguous()
                base.pgd_update(x_batch, grad, lower, upper, self.step_size, self.grad_sign)
        return x_batch, success

    index_offset = 0  # global index of x[0] when this object attacks one shard of a larger batch (shard.py)
    chunk_coupling = None  # nothing ties the examples of a chunk together (EOT.py:33-35: per-example loss vector)

    def _begin_attack(self):
        base = getattr(self.model, 'base_model', self.model)
        if hasattr(base, 'begin_attack'):
            base.begin_attack()

    def _begin_batch(self, start, tag=None):
        # noise streams (dither, NES) are keyed by the chunk's GLOBAL position, not by what ran before it
        base = getattr(self.model, 'base_model', self.model)
        if hasattr(base, 'begin_batch'):
            base.begin_batch(self.index_offset + start, 0 if tag is None else int(tag) + 1)

    def _run_batches(self, x, y, lower, upper, tag=None):
        base = getattr(self.model, 'base_model', self.model)
        try:
            return self._run_batches_once(x, y, lower, upper, tag)
        except Exception as e:  # the engine's NativeError (kept generic: the CPU test double has no such class)
            # A stream-K hand-off that timed out (the GPU was shared: include/speakerguard_hip.h, sg_set_streamk): the
            # inputs are untouched (attack_batch works on copies) and the noise keys depend on positions only, so the
            # batches are run again, once, as one block per tile -- the same bits, no residency requirement.
            if 'hand-off' not in str(e) or not hasattr(base, 'set_streamk') or getattr(base, 'streamk', True) is False:
                raise
            import warnings
            warnings.warn('a stream-K hand-off timed out (is another process using this GPU?): this model now runs its contractions '
                          'as one block per tile (same results, a few percent slower); model.set_streamk(True) switches back')
            base.set_streamk(False)
            return self._run_batches_once(x, y, lower, upper, tag)

    def _run_batches_once(self, x, y, lower, upper, tag=None):
        n_audios = x.shape[0]
        batch_size = min(self.batch_size, n_audios)
        n_batches = int(np.ceil(n_audios / float(batch_size)))
        adver, success = [], []
        for batch_id in range(n_batches):
            sl = slice(batch_id * batch_size, (batch_id + 1) * batch_size)
            bid = batch_id if tag is None else '{}-{}'.format(tag, batch_id)
            self._begin_batch(sl.start, tag)
            a, s = self.attack_batch(x[sl], y[sl], lower[sl], upper[sl], bid)
            adver.append(a)
            success += s
        # the success flags are on the host, so every launch of these batches has finished: a kernel that flagged
        # its own output as invalid (engine health word) must not go unnoticed
        base = getattr(self.model, 'base_model', self.model)
        if hasattr(base, 'check_health'):
            base.check_health()
        return (adver[0] if len(adver) == 1 else torch.cat(adver, 0)), success  # (one batch: attack_batch's own output tensor)

    def _check_inputs(self, x, y):
        lower, upper = -1, 1
        peak = float(x.max())  # one device round trip (the reference's chained comparison on the tensor makes two)
        assert lower <= peak < upper, 'generating adversarial examples should be done in [-1, 1) float domain'
        n_audios, n_channels, _ = x.size()
        assert n_channels == 1, 'Only Support Mono Audio'
        assert y.shape[0] == n_audios, 'The number of x and y should be equal'

    def attack(self, x, y):
        self._check_inputs(x, y)
        self._begin_attack()
        lower = torch.tensor(-1, device=x.device, dtype=x.dtype).expand_as(x)
        upper = torch.tensor(1, device=x.device, dtype=x.dtype).expand_as(x)
        return self._run_batches(x, y, lower, upper)
