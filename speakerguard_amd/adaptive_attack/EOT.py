"""Expectation over transformation; mirrors reference adaptive_attack/EOT.py.

``forward`` keeps the reference contract -- it returns SUMS over the EOT batches of the
per-batch means, the caller divides (attack/FGSM.py:50-53) -- but the model call, the loss and
``loss.backward(ones)`` (EOT.py:32-35) are one native ``model.loss_grad`` call.
"""
import torch


class EOT:

    def __init__(self, model, loss, EOT_size=1, EOT_batch_size=1, use_grad=True):
        self.model = model
        self.loss = loss
        self.EOT_size = EOT_size
        self.EOT_batch_size = EOT_batch_size
        self.EOT_num_batches = self.EOT_size // self.EOT_batch_size
        self.use_grad = use_grad

    def forward(self, x_batch, y_batch, EOT_num_batches=None, EOT_batch_size=None, use_grad=None):
        EOT_num_batches = EOT_num_batches if EOT_num_batches else self.EOT_num_batches
        EOT_batch_size = EOT_batch_size if EOT_batch_size else self.EOT_batch_size
        use_grad = use_grad if use_grad else self.use_grad  # (sic) same truthiness rule as EOT.py:19
        n_audios, n_channels, max_len = x_batch.size()
        grad = None
        scores = None
        loss = 0
        decisions = [[] for _ in range(n_audios)]
        base = getattr(self.model, 'base_model', self.model)
        keyed = hasattr(base, 'row_keys')  # the engine keys its noise by (utterance, repeat), not by the row of the call
        for EOT_index in range(EOT_num_batches):
            x_rep = x_batch.repeat(EOT_batch_size, 1, 1)
            y_rep = y_batch.repeat(EOT_batch_size)
            if keyed:
                base._rep_rows = n_audios if EOT_batch_size > 1 else 0
            try:
                dec, sc, ls, g = self.model.loss_grad(x_rep, y_rep, self.loss, want_grad=bool(use_grad))
            finally:
                if keyed:
                    base._rep_rows = 0
            sc_m = sc.view(EOT_batch_size, -1, sc.shape[1]).mean(0)
            ls_m = ls.view(EOT_batch_size, -1).mean(0)
            if EOT_index == 0:
                scores, loss = sc_m, ls_m
            else:
                scores = scores + sc_m
                loss = loss + ls_m
            if use_grad:
                g_m = g.view(EOT_batch_size, -1, n_channels, max_len).mean(0)
                grad = g_m if grad is None else grad + g_m
            dec = dec.view(EOT_batch_size, -1).cpu().numpy()
            for ii in range(n_audios):
                decisions[ii] += list(dec[:, ii])
        return scores, loss, grad, decisions

    __call__ = forward
