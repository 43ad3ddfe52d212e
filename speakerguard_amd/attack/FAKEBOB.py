"""FAKEBOB black-box attack (NES gradient + momentum + per-example plateau LR) and its threshold
estimation for black-box SV/OSI; mirrors reference attack/FAKEBOB.py:50-295.  Queries are forward-only
passes of the native engine.

Faithfulness notes:
  * ``last_ls = [[]] * n`` (FAKEBOB.py:56) aliases one list across the batch; the in-place append at
    :95 therefore leaks earlier examples' losses into later examples' plateau history until each
    entry is rebound.  That changes the LR schedule and is reproduced on purpose.
  * ``estimate_threshold`` (:280-295) iterates ``for xx in x.unsqueeze(0)``, i.e. exactly once with the
    WHOLE (N,1,T) batch, and ``estimate_threshold_run`` reads only example 0's decision / top score
    (:214-216, :246-249) while updating all N examples -- reproduced as is.
  * At the convergence check (:106-116) the reference assigns the UNFILTERED ``loss_np`` to
    ``prev_loss`` after examples were dropped, which mis-aligns (or crashes) for batch_size > 1;
    the reference default is batch_size=1 where both behaviours coincide.  Here the surviving
    entries are kept aligned.
"""
import numpy as np
import torch

from ..adaptive_attack.EOT import EOT
from ..adaptive_attack.NES import NES
from .Attack import Attack
from .utils import resolve_loss


class FAKEBOB(Attack):

    def __init__(self, model, threshold=None,
                 task='CSI', targeted=False, confidence=0.,
                 epsilon=0.002, max_iter=1000,
                 max_lr=0.001, min_lr=1e-6,
                 samples_per_draw=50, samples_per_draw_batch_size=50, sigma=0.001, momentum=0.9,
                 plateau_length=5, plateau_drop=2.,
                 stop_early=True, stop_early_iter=100,
                 batch_size=1, EOT_size=1, EOT_batch_size=1, verbose=1, noise_fn=None):
        self.model = model
        self.noise_fn = noise_fn  # optional explicit NES noise source (tests); default: engine generator
        self.threshold = threshold
        self.task = task
        self.targeted = targeted
        self.confidence = confidence
        self.epsilon = epsilon
        self.max_iter = max_iter
        self.max_lr = max_lr
        self.min_lr = min_lr
        self.samples_per_draw = samples_per_draw
        self.samples_per_draw_batch_size = samples_per_draw_batch_size
        self.sigma = sigma
        self.momentum = momentum
        self.plateau_length = plateau_length
        self.plateau_drop = plateau_drop
        self.stop_early = stop_early
        self.stop_early_iter = stop_early_iter
        self.batch_size = batch_size
        self.EOT_size = EOT_size
        self.EOT_batch_size = EOT_batch_size
        self.verbose = verbose

    # the plateau history aliases over the examples of a chunk (``[[]] * n``, see above): a chunk's examples have to
    # be attacked together for the reference's result, so shard.ShardedAttack cuts on multiples of batch_size
    chunk_coupling = 'chunk'

    @staticmethod
    def delete_found(flags, tensors, lists):
        """Drop the examples whose flag is < 0 from every per-example tensor / list (:125-168)."""
        keep = [i for i, f in enumerate(flags) if not (f < 0)]
        if not keep:
            return None, None
        idx = torch.tensor(keep, device=tensors[0].device)
        return [t.index_select(0, idx) for t in tensors], [[l[i] for i in keep] for l in lists]

    def get_grad(self, x, y):
        NES_wrapper = NES(self.samples_per_draw, self.samples_per_draw_batch_size, self.sigma, self.EOT_wrapper,
                          noise_fn=self.noise_fn)
        return NES_wrapper(x, y)

    def attack_batch(self, x_batch, y_batch, lower, upper, batch_id):
        n_audios = x_batch.shape[0]
        last_ls = [[]] * n_audios
        lr = [self.max_lr] * n_audios
        prev_loss = [np.inf] * n_audios
        adver_x = x_batch.clone()
        grad = torch.zeros_like(x_batch)
        best_adver_x = adver_x.clone()
        best_loss = [np.inf] * n_audios
        consider_index = list(range(n_audios))
        lower = lower.expand_as(x_batch)
        upper = upper.expand_as(x_batch)
        base = getattr(self.model, 'base_model', self.model)

        for it in range(self.max_iter + 1):
            prev_grad = grad.clone()
            loss, grad, adver_loss, _, y_pred = self.get_grad(adver_x, y_batch)
            adver_loss_h = adver_loss.cpu().numpy()
            loss_h = [float(l) for l in loss.cpu().numpy()]
            for ii, adver_l in enumerate(adver_loss_h):
                index = consider_index[ii]
                if adver_l < best_loss[index]:
                    best_loss[index] = float(adver_l)
                    best_adver_x[index] = adver_x[ii]
            if self.verbose:
                print("batch: {} iter: {}, loss: {}, y: {}, y_pred: {}, best loss: {}".format(
                    batch_id, it, adver_loss_h, y_batch.cpu().numpy(), y_pred, best_loss))
            ts, ls = self.delete_found(adver_loss_h, [adver_x, y_batch, prev_grad, grad, lower, upper],
                                       [consider_index, last_ls, lr, prev_loss, loss_h])
            if ts is None:  # all found
                break
            adver_x, y_batch, prev_grad, grad, lower, upper = ts
            consider_index, last_ls, lr, prev_loss, loss_h = ls

            if it < self.max_iter:
                for jj, loss_ in enumerate(loss_h):
                    last_ls[jj].append(loss_)
                    last_ls[jj] = last_ls[jj][-self.plateau_length:]
                    if last_ls[jj][-1] > last_ls[jj][0] and len(last_ls[jj]) == self.plateau_length:
                        if lr[jj] > self.min_lr:
                            lr[jj] = max(lr[jj] / self.plateau_drop, self.min_lr)
                        last_ls[jj] = []
                lr_t = torch.tensor(lr, device=adver_x.device, dtype=torch.float)
                # momentum mix (:93) + per-example sign step (:103) + epsilon-ball clamp (:104): one native pass
                adver_x, grad = adver_x.contiguous(), grad.contiguous()
                base.fakebob_step(adver_x, grad, prev_grad.contiguous(), lr_t, lower.contiguous(), upper.contiguous(),
                                  self.momentum, self.grad_sign)

                if self.stop_early and it % self.stop_early_iter == 0:
                    loss_np = np.array(loss_h)
                    converge_loss = np.array(prev_loss) * 0.9999 - loss_np
                    ts, ls = self.delete_found(converge_loss, [adver_x, y_batch, prev_grad, grad, lower, upper],
                                               [consider_index, last_ls, lr, list(loss_np), loss_h])
                    if ts is None:  # all converged
                        break
                    adver_x, y_batch, prev_grad, grad, lower, upper = ts
                    consider_index, last_ls, lr, prev_loss, loss_h = ls

        success = [bl < 0 for bl in best_loss]
        return best_adver_x, success

    def attack(self, x, y):
        if self.task in ['SV', 'OSI'] and self.threshold is None:
            raise NotImplementedError('You are running black box attack for {} task, '
                                      'but the threshold not specified. Consider calling estimate threshold'.format(self.task))
        self.loss, self.grad_sign = resolve_loss('Margin', self.targeted, self.confidence, self.task, self.threshold, False)
        self.EOT_wrapper = EOT(self.model, self.loss, self.EOT_size, self.EOT_batch_size, False)
        lower, upper = -1, 1
        assert lower <= x.max() < upper, 'generating adversarial examples should be done in [-1, 1) float domain'
        n_audios, n_channels, _ = x.size()
        assert n_channels == 1, 'Only Support Mono Audio'
        assert y.shape[0] == n_audios, 'The number of x and y should be equal'
        upper = torch.clamp(x + self.epsilon, max=upper)
        lower = torch.clamp(x - self.epsilon, min=lower)
        batch_size = min(self.batch_size, n_audios)
        n_batches = int(np.ceil(n_audios / float(batch_size)))
        adver, success = [], []
        base = getattr(self.model, 'base_model', self.model)
        if hasattr(base, 'begin_attack'):
            base.begin_attack()
        for batch_id in range(n_batches):
            sl = slice(batch_id * batch_size, (batch_id + 1) * batch_size)
            if hasattr(base, 'begin_batch'):  # NES / dither noise keyed by the chunk's global position (shard-invariant)
                base.begin_batch(getattr(self, 'index_offset', 0) + sl.start, 0)
            a, s = self.attack_batch(x[sl], y[sl], lower[sl], upper[sl], batch_id)
            adver.append(a)
            success += s
        if hasattr(base, 'check_health'):  # see FGSM._run_batches
            base.check_health()
        return torch.cat(adver, 0), success

    # ------------------------------------------------------------------ threshold estimation (:210-295)
    def estimate_threshold_run(self, x, step=0.1):
        """Push a REJECTED voice up a ladder of candidate thresholds until the model accepts it; the top
        score at that moment is the estimate (:210-278).  None if example 0 is already accepted."""
        n_audios = x.shape[0]
        d, s = self.model.make_decision(x)
        if int(d[0]) != -1:
            return None  # already accepted, cannot be used to estimate the threshold (:217-218)
        y = torch.full((n_audios,), -1, dtype=torch.long, device=x.device)
        init_score = float(np.max(s[0].cpu().numpy()))
        delta = np.abs(init_score * step)
        threshold = init_score + delta
        adver_x = x.clone().contiguous()
        grad = torch.zeros_like(x)
        upper = torch.clamp(x + self.epsilon, max=1).contiguous()
        lower = torch.clamp(x - self.epsilon, min=-1).contiguous()
        base = getattr(self.model, 'base_model', self.model)
        iter_outer = 0
        while True:
            self.loss, self.grad_sign = resolve_loss('Margin', False, 0., self.task, threshold, False)
            self.EOT_wrapper = EOT(self.model, self.loss, self.EOT_size, self.EOT_batch_size, False)
            iter_inner = 0
            last_ls = [[]] * n_audios
            lr = [self.max_lr] * n_audios
            while True:
                decision, score = self.model.make_decision(adver_x)
                top = float(np.max(score[0].cpu().numpy()))
                if self.verbose:
                    print(iter_outer, iter_inner, top, getattr(self.model, 'threshold', None))
                if int(decision[0]) != -1:  # accepted: found the threshold (:251-252)
                    return top
                elif top >= threshold:      # candidate exceeded without acceptance: raise it (:253-254)
                    break
                prev_grad = grad.clone()
                loss, grad, _, _, _ = self.get_grad(adver_x, y)
                loss_h = [float(l) for l in loss.cpu().numpy()]
                for jj, loss_ in enumerate(loss_h):
                    last_ls[jj].append(loss_)
                    last_ls[jj] = last_ls[jj][-self.plateau_length:]
                    if last_ls[jj][-1] > last_ls[jj][0] and len(last_ls[jj]) == self.plateau_length:
                        if lr[jj] > self.min_lr:
                            lr[jj] = max(lr[jj] / self.plateau_drop, self.min_lr)
                        last_ls[jj] = []
                lr_t = torch.tensor(lr, device=adver_x.device, dtype=torch.float)
                grad = grad.contiguous()
                base.fakebob_step(adver_x, grad, prev_grad.contiguous(), lr_t, lower, upper, self.momentum,
                                  self.grad_sign)  # :260 momentum mix, :269-270 sign step + clamp
                iter_inner += 1
            threshold += delta
            iter_outer += 1

    def estimate_threshold(self, x, step=0.1):
        if self.task == 'CSI':
            print("--- Warning: no need to estimate threshold for CSI, quitting ---")
            return
        estimated_thresholds = []
        for xx in x.unsqueeze(0):  # sic (:287): one pass over the whole batch
            estimated_threshold = self.estimate_threshold_run(xx, step)
            if estimated_threshold is not None:
                estimated_thresholds.append(estimated_threshold)
        self.threshold = np.mean(estimated_thresholds) if len(estimated_thresholds) > 0 else None
        return self.threshold
