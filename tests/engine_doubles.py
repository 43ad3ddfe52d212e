"""Test doubles implementing the engine protocol (``loss_grad`` / ``pgd_update`` / ``make_decision``)
on the CPU with PyTorch autograd + the oracle losses.  They let the HOST logic of
speakerguard_amd.attack.* be checked without a GPU against the reference-generated fixtures.
Never imported by the product.
"""
import torch

from oracle import attacks as oatk


def oracle_loss_from_spec(spec):
    """speakerguard_amd loss object -> the oracle's callable with the same meaning."""
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    if isinstance(spec, SEC4SR_CrossEntropy):
        return oatk.cross_entropy_loss
    return lambda s, y: oatk.margin_loss(s, y, spec.targeted, spec.confidence, spec.task, spec.threshold, spec.clip_max)


class AutogradEngine:
    """Wraps any differentiable ``make_decision`` model (ToyModel, oracle XvPlda)."""

    def __init__(self, model, flag=None):
        self.model = model
        self.threshold = model.threshold
        self.flag = flag

    def _md(self, x):
        return self.model.make_decision(x) if self.flag is None else self.model.make_decision(x, flag=self.flag)

    def make_decision(self, x):
        with torch.no_grad():
            return self._md(x)

    def loss_grad(self, x, y, loss_spec, flag=0, want_grad=True):
        loss_fn = oracle_loss_from_spec(loss_spec)
        xx = x.detach().clone().requires_grad_(bool(want_grad))
        with torch.set_grad_enabled(bool(want_grad)):
            dec, sc = self._md(xx)
            loss = loss_fn(sc, y)
        grad = None
        if want_grad:
            loss.backward(torch.ones_like(loss))
            grad = xx.grad.detach()
        return dec.detach(), sc.detach(), loss.detach(), grad

    def pgd_update(self, x, grad, lower, upper, step_size, grad_sign):
        x += step_size * torch.sign(grad) * grad_sign
        x.copy_(torch.min(torch.max(x, lower), upper))
        return x
