"""Backward-pass differentiable approximation; mirrors reference adaptive_attack/BPDA.py:7-65.

The reference wraps a non-differentiable transform ``ori_f`` in a ``torch.autograd.Function`` whose backward
differentiates a substitute ``sub_f`` instead (``QT = BPDA(QT_Non_Diff, lambda *args: args[0])``,
defense/time_domain.py:44).  The engine has no autograd graph, so the same contract is expressed as the
``fwd`` / ``bwd`` pair ``defended_model`` chains by hand (like ``FeCoDefense``):

    out, saved = d.fwd(x, *args)      # out = ori_f(x, *args)
    gx = d.bwd(saved, g_out)          # g_out . d sub_f / d x ; the identity substitute returns g_out itself

Only the FIRST positional argument is differentiated (the audio), which is what every defense in the reference
needs (its other arguments are python scalars).  A general ``sub_f`` is differentiated with torch.autograd on the
substitute alone -- host-side glue around a user-supplied function, not part of the accelerated path.
"""
import torch


def _identity(*args):
    return args[0]


class BPDA:

    def __init__(self, ori_f, sub_f=None):
        self.ori_f = ori_f
        self.sub_f = sub_f if sub_f is not None else _identity
        self._identity = sub_f is None or sub_f is _identity

    def fwd(self, x, *args, **kwargs):
        with torch.no_grad():
            out = self.ori_f(x, *args, **kwargs)
        return out, (x, args, kwargs)

    def bwd(self, saved, g_out):
        if self._identity:
            return g_out
        x, args, kwargs = saved
        xin = x.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            sub = self.sub_f(xin, *args, **kwargs)
        return torch.autograd.grad(sub, xin, g_out)[0]

    def __call__(self, x, *args, **kwargs):
        return self.fwd(x, *args, **kwargs)[0]


def straight_through(ori_f):
    """``BPDA(ori_f, identity)``: forward = ori_f, backward = identity (the form the reference uses for QT / codecs)."""
    return BPDA(ori_f, None)
