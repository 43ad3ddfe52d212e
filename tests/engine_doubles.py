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

    def __init__(self, model, flag=None, per_row=False):
        self.model = model
        self.threshold = model.threshold
        self.flag = flag
        # per_row: every utterance goes through the model as a batch of one, so that -- like on the HIP engine -- an
        # utterance's result does not depend on what else is in the batch (CPU BLAS picks kernels by batch size)
        self.per_row = per_row

    def _md1(self, x):
        return self.model.make_decision(x) if self.flag is None else self.model.make_decision(x, flag=self.flag)

    def _md(self, x):
        if not self.per_row or x.shape[0] <= 1:
            return self._md1(x)
        outs = [self._md1(x[i:i + 1]) for i in range(x.shape[0])]
        return torch.cat([o[0] for o in outs], 0), torch.cat([o[1] for o in outs], 0)

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

    # ---- attack-state updates: plain torch restatements of the reference lines (CPU double only)
    def cw2_step(self, modifier, exp_avg, exp_avg_sq, x, input_cur, grad1, const, lr, step_t):
        if grad1 is not None:  # CW2.py:75-82 + torch.optim.Adam single-tensor update
            g = (const.view(-1, 1, 1) * grad1 + 2.0 * (input_cur - x)) * (1.0 - input_cur * input_cur)
            b1, b2, eps = 0.9, 0.999, 1e-8
            exp_avg.lerp_(g, 1 - b1)
            exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (exp_avg_sq.sqrt() / (1 - b2 ** step_t) ** 0.5).add_(eps)
            modifier.addcdiv_(exp_avg, denom, value=-lr / (1 - b1 ** step_t))
        nxt = torch.tanh(modifier + torch.atanh(x * 0.999999))
        return nxt, torch.sum(torch.square(nxt - x), dim=(1, 2))

    def nes_queries(self, x, half, with_clean, sigma, seed, pair_base, noise_in=None, want_noise=False, index_base=0):
        assert noise_in is not None, "the CPU double has no counter-based generator: pass noise_fn"
        noise = torch.cat((noise_in, -noise_in), 1)
        if with_clean:
            noise = torch.cat((torch.zeros_like(x).unsqueeze(1), noise), 1)
        q = (noise * sigma + x.unsqueeze(1)).view(-1, x.shape[1], x.shape[2])
        return q, (noise_in if want_noise else None)

    def nes_grad(self, loss, grad, n, T, half, with_clean, seed, pair_base, noise_in, accumulate, final_sigma, final_batches,
                 index_base=0):
        l = loss[:, 1:] if with_clean else loss
        noise = torch.cat((noise_in, -noise_in), 1)
        g = torch.mean(l.unsqueeze(2).unsqueeze(3) * noise, 1)
        if accumulate:
            g = grad + g
        if final_sigma > 0:
            g = g / final_sigma / final_batches
        grad.copy_(g)
        return grad

    def fakebob_step(self, x, grad, prev_grad, lr, lower, upper, momentum, grad_sign):
        g = momentum * prev_grad + (1.0 - momentum) * grad
        grad.copy_(g)
        x.copy_(torch.min(torch.max(x + grad_sign * lr.view(-1, 1, 1) * torch.sign(g), lower), upper))
        return x, grad
