"""Carlini-Wagner L2 attack; mirrors reference attack/CW2.py.

Per iteration (CW2.py:67-111): input_x = tanh(modifier + atanh(0.999999 x)); margin loss with
clip and d loss1 / d input_x from one native ``model.loss_grad`` call; the chain through tanh, the
L2 term, torch.optim.Adam's update and the next input_x are one more native pass (``cw2_step``).  The
per-example bookkeeping (best_l2 / best_score / global best, CW2.py:102-111) is kept on the
device as masks, so the host only synchronises for the early-stop test every
``stop_early_iter`` iterations (the reference synchronises every iteration).
"""
import numpy as np
import torch

from .FGSM import FGSM
from .utils import SEC4SR_MarginLoss


class CW2(FGSM):

    def __init__(self, model, task='CSI',
                 targeted=False,
                 confidence=0.,
                 initial_const=1e-3,
                 binary_search_steps=9,
                 max_iter=10000,
                 stop_early=True,
                 stop_early_iter=1000,
                 lr=1e-2,
                 batch_size=1,
                 verbose=1):
        self.model = model
        self.task = task
        self.targeted = targeted
        self.confidence = confidence
        self.initial_const = initial_const
        self.binary_search_steps = binary_search_steps
        self.max_iter = max_iter
        self.stop_early = stop_early
        self.stop_early_iter = stop_early_iter
        self.lr = lr
        self.batch_size = batch_size
        self.verbose = verbose
        self.threshold = None
        if self.task in ['SV', 'OSI']:
            self.threshold = self.model.threshold
            print('Running white box attack for {} task, directly using the true threshold {}'.format(self.task, self.threshold))
        self.loss = SEC4SR_MarginLoss(targeted=self.targeted, confidence=self.confidence, task=self.task,
                                      threshold=self.threshold, clip_max=True)

    def attack_batch(self, x_batch, y_batch, lower, upper, batch_id):
        n_audios = x_batch.shape[0]
        dev = x_batch.device
        const = torch.full((n_audios,), self.initial_const, dtype=torch.float, device=dev)
        lower_bound = torch.zeros(n_audios, dtype=torch.float, device=dev)
        upper_bound = torch.full((n_audios,), 1e10, dtype=torch.float, device=dev)
        inf = torch.full((n_audios,), float('inf'), device=dev)
        global_best_l2 = inf.clone()
        global_best_adver_x = x_batch.clone()
        global_best_score = torch.full((n_audios,), -2, dtype=torch.int64, device=dev)  # -2: never succeeded (CW2.py:52)
        base = getattr(self.model, 'base_model', self.model)
        x_batch = x_batch.contiguous()

        for _ in range(self.binary_search_steps):
            modifier = torch.zeros_like(x_batch)      # fresh modifier + Adam state per search step (CW2.py:56-57)
            exp_avg = torch.zeros_like(x_batch)
            exp_avg_sq = torch.zeros_like(x_batch)
            best_l2 = inf.clone()
            best_score = torch.full((n_audios,), -2, dtype=torch.int64, device=dev)
            continue_flag = True
            prev_loss = np.inf
            input_x, loss2 = base.cw2_step(modifier, None, None, x_batch, None, None, const, self.lr, 0)
            for n_iter in range(self.max_iter + 1):
                if not continue_flag:
                    break
                want_grad = n_iter < self.max_iter
                decisions, scores, loss1, g1 = self.model.loss_grad(input_x, y_batch, self.loss, want_grad=want_grad)
                loss = const * loss1 + loss2
                if want_grad:  # chain through tanh + L2 term + Adam, and the next input, in one native pass
                    input_next, loss2_next = base.cw2_step(modifier, exp_avg, exp_avg_sq, x_batch, input_x, g1, const,
                                                           self.lr, n_iter + 1)
                if self.verbose:
                    print("batch: {}, c: {}, iter: {}, loss: {}, loss1: {}, loss2: {}, y_pred: {}, y: {}".format(
                        batch_id, const.cpu().numpy(), n_iter, loss.cpu().numpy().tolist(), loss1.cpu().numpy().tolist(),
                        loss2.cpu().numpy().tolist(), decisions.cpu().numpy(), y_batch.cpu().numpy()))
                if self.stop_early and n_iter % self.stop_early_iter == 0:
                    # batch-mean criterion (CW2.py:96-100); a sharded run takes the mean over the whole chunk (shard.py)
                    mean_loss = self.batch_mean(loss) if self.batch_mean is not None else float(loss.mean().item())
                    if mean_loss > 0.9999 * prev_loss:
                        print("Early Stop ! ")
                        continue_flag = False
                    prev_loss = mean_loss
                ok = loss1 <= 0  # attack succeeds with at least kappa confidence
                c1 = ok & (loss2 < best_l2)
                best_l2 = torch.where(c1, loss2, best_l2)
                best_score = torch.where(c1, decisions, best_score)
                c2 = ok & (loss2 < global_best_l2)
                global_best_l2 = torch.where(c2, loss2, global_best_l2)
                global_best_score = torch.where(c2, decisions, global_best_score)
                global_best_adver_x = torch.where(c2.view(-1, 1, 1), input_x, global_best_adver_x)
                if want_grad:
                    input_x, loss2 = input_next, loss2_next

            # binary search on const (CW2.py:113-123)
            succeeded = best_score != -2
            upper_bound = torch.where(succeeded, torch.minimum(upper_bound, const), upper_bound)
            lower_bound = torch.where(succeeded, lower_bound, torch.maximum(lower_bound, const))
            bisect = upper_bound < 1e9
            const = torch.where(bisect, (lower_bound + upper_bound) / 2, torch.where(succeeded, const, const * 10))
            if self.verbose:
                print(const.cpu().numpy(), best_l2.cpu().numpy(), global_best_l2.cpu().numpy())

        success = (global_best_score != -2).tolist()
        return global_best_adver_x, success

    batch_mean = None  # shard.ShardedAttack: mean of the chunk's losses over all ranks

    @property
    def chunk_coupling(self):
        """The examples of a chunk share the early-stop test (their mean loss) and nothing else."""
        return 'mean' if self.stop_early else None

    def attack(self, x, y):
        return super().attack(x, y)
