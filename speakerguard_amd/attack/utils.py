"""Loss selection and majority-vote prediction; mirrors reference attack/utils.py.

The loss objects are *descriptions* (``.native()`` gives the ``sg_loss_spec`` the HIP tail kernel
evaluates together with d loss / d scores); calling one on a (scores, label) pair evaluates the
same formula with torch ops on whatever device the scores live on, for callers that only need
values (no autograd graph is produced anywhere in this package).
"""
import ctypes as C
import math
import warnings
from collections import Counter

import numpy as np
import torch

from .. import _native as N


class _Loss:
    loss_id = N.SG_LOSS_MARGIN

    def native(self):
        s = N.LossSpec()
        s.loss = self.loss_id
        s.task = N.SG_TASK[self.task]
        s.targeted = int(bool(self.targeted))
        s.clip_max = int(bool(self.clip_max))
        s.confidence = float(self.confidence)
        s.threshold = float(self.threshold) if self.threshold is not None else 0.0
        return s

    def forward(self, scores, label):
        raise NotImplementedError

    def __call__(self, scores, label):
        return self.forward(scores, label)


class SEC4SR_CrossEntropy(_Loss):
    """Per-example cross entropy, imposter rows (label -1) contribute 0 (attack/utils.py:7-29)."""
    loss_id = N.SG_LOSS_ENTROPY

    def __init__(self, reduction='none', task='CSI'):
        assert task == 'CSI'  # CrossEntropy only supports the CSI task (:12)
        self.task, self.targeted, self.clip_max, self.confidence, self.threshold = task, False, False, 0., None

    def forward(self, scores, label):
        label = label.to(scores.device)
        valid = label != -1
        lse = torch.logsumexp(scores, dim=1)
        picked = scores.gather(1, label.clamp(min=0).unsqueeze(1)).squeeze(1)
        return torch.where(valid, lse - picked, torch.zeros_like(lse))


class SEC4SR_MarginLoss(_Loss):
    """CW-style margin for CSI / SV / OSI, targeted or not (attack/utils.py:31-102)."""

    def __init__(self, targeted=False, confidence=0., task='CSI', threshold=None, clip_max=True):
        self.targeted, self.confidence, self.task = targeted, confidence, task
        self.threshold, self.clip_max = threshold, clip_max

    def forward(self, scores, label):
        label = label.to(scores.device)
        conf, thr = self.confidence, self.threshold
        if self.task == 'SV':
            s0 = scores[:, 0]
            enroll = label == 0
            assert bool(((label == 0) | (label == -1)).all()), 'SV task should not have labels out of 0 and -1'
            up = thr + conf - s0    # towards acceptance
            down = s0 + conf - thr  # towards rejection
            loss = torch.where(enroll == bool(self.targeted), up, down)
        else:
            valid = label != -1
            onehot = torch.zeros_like(scores).scatter_(1, label.clamp(min=0).unsqueeze(1), 1.0)
            real = (onehot * scores).sum(1)
            other = ((1 - onehot) * scores - onehot * 10000).max(1)[0]
            top = scores.max(1)[0]
            if self.targeted:
                l_valid = other + conf - real if self.task == 'CSI' else torch.clamp(other, min=thr) + conf - real
            elif self.task == 'CSI':
                l_valid = real + conf - other
            else:
                l_valid = torch.minimum(top + conf - thr, torch.clamp(real, min=thr) + conf - other)
            if self.task == 'OSI':
                l_imp = top + conf - thr if self.targeted else thr + conf - top
            else:
                l_imp = torch.zeros_like(real)
            loss = torch.where(valid, l_valid, l_imp)
        if self.clip_max:
            loss = torch.clamp(loss, min=0)
        return loss.float()


class ScoreVJP(_Loss):
    """``loss[b] = sum_s coef[b, s] * scores[b, s]``: the vector-Jacobian product of the scores (SG_LOSS_LINEAR).

    The reference lets a caller put ANY differentiable function of the scores behind ``loss.backward`` (adaptive_attack/
    EOT.py:33-35); the hand-coded backward knows the two losses of attack/utils.py.  A caller-defined loss L closes the gap
    with one extra line: compute ``coef = dL/dscores`` (B, S) yourself and ask for ``model.loss_grad(x, y, ScoreVJP(coef))``
    -- the returned gradient is dL/dx.  ``defended_model`` uses it for the 'average' order, where the loss is taken of
    the MEAN score of several defended branches."""
    loss_id = N.SG_LOSS_LINEAR

    def __init__(self, coef):
        self.coef = coef.to(torch.float32).contiguous()  # kept alive for as long as the native spec may be used
        if not self.coef.is_cuda or self.coef.dim() != 2:
            raise N.NativeError("ScoreVJP needs a (B, S) tensor on the HIP device")
        self.task, self.targeted, self.clip_max, self.confidence, self.threshold = 'CSI', False, False, 0., None

    def native(self):
        s = super().native()
        s.coef_dev = self.coef.data_ptr()
        return s

    def check(self, B, S):
        """Called by the models before the native pass reads coef as a (B, S) table."""
        if tuple(self.coef.shape) != (B, S):
            raise ValueError("ScoreVJP: coef is %s, the pass scores %d rows against %d speakers" % (tuple(self.coef.shape), B, S))

    def forward(self, scores, label=None):
        return (self.coef.to(scores.device) * scores).sum(1)


def loss_dscores(model, scores, label, loss_spec):
    """(decisions, loss, d loss / d scores) of `loss_spec` on given scores (B, S): the loss stage of the tail kernels
    alone (sg_loss_eval), for scores that no single model pass produced."""
    base = getattr(model, 'base_model', model)
    scores = scores.to(torch.float32).contiguous()
    B, S = scores.shape
    label = label.to(scores.device, torch.int64).contiguous()
    dec = torch.empty(B, device=scores.device, dtype=torch.int64)
    loss = torch.empty(B, device=scores.device, dtype=torch.float32)
    dsc = torch.empty(B, S, device=scores.device, dtype=torch.float32)
    thr = float(base.threshold) if base.threshold is not None and np.isfinite(base.threshold) else -math.inf
    spec = loss_spec.native()
    base.ctx.call("sg_loss_eval", N._ptr(scores), N._ptr(label), B, S, thr, C.byref(spec), N._ptr(dec), N._ptr(loss),
                  N._ptr(dsc), N.current_stream_ptr(scores.device))
    return dec, loss, dsc


def resolve_loss(loss_name='Entropy', targeted=False, confidence=0., task='CSI', threshold=None, clip_max=True):
    """attack/utils.py:104-116: SV/OSI force the margin loss; grad_sign is -1 for Margin."""
    assert loss_name in ['Entropy', 'Margin']
    assert task in ['CSI', 'SV', 'OSI']
    if task == 'SV' or task == 'OSI' or loss_name == 'Margin':
        loss = SEC4SR_MarginLoss(targeted=targeted, confidence=confidence, task=task, threshold=threshold, clip_max=clip_max)
        if (task == 'SV' or task == 'OSI') and loss_name == 'Entropy':
            warnings.warn('You are targeting {} task. Force using Margin Loss.'.format(task))
    else:
        loss = SEC4SR_CrossEntropy(reduction='none', task='CSI')
        loss.targeted = targeted  # the formula ignores it; the fused loop needs it for the success flag
    grad_sign = (1 - 2 * int(targeted)) if loss_name == 'Entropy' else -1
    return loss, grad_sign


def resolve_prediction(decisions):
    """Majority vote over the EOT decisions of each example, first-seen wins ties (:118-125)."""
    return np.array([Counter(d).most_common(1)[0][0] for d in decisions])
