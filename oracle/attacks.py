"""Losses and attack loops, restated on PyTorch-CPU autograd.  TEST INFRASTRUCTURE.

PINNED by tests/golden/attack_*.npz (trajectories produced by the reference's own
attack/{FGSM,PGD,CWinf,CW2,FAKEBOB}.py and adaptive_attack/{EOT,NES}.py).

Follows, function by function:
  attack/utils.py:7-29 (SEC4SR_CrossEntropy), :31-102 (SEC4SR_MarginLoss), :104-116
  (resolve_loss), :118-125 (resolve_prediction); attack/Attack.py:11-15 (compare);
  adaptive_attack/EOT.py:16-54; attack/FGSM.py:38-98; attack/PGD.py:40-78; attack/CWinf.py;
  attack/CW2.py:41-132; adaptive_attack/NES.py:15-56; attack/FAKEBOB.py:50-208.
"""
from collections import Counter

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- losses
def cross_entropy_loss(scores, label):
    """SEC4SR_CrossEntropy(reduction='none'): imposter rows (label -1) give 0 * sum(scores)."""
    loss = torch.zeros(label.shape[0], dtype=torch.float)
    consider = torch.nonzero(label != -1, as_tuple=True)[0]
    if len(consider) > 0:
        loss = loss.index_put((consider,), F.cross_entropy(scores[consider], label[consider], reduction="none"))
    imposter = torch.nonzero(label == -1, as_tuple=True)[0]
    if len(imposter) > 0:
        loss = loss.index_put((imposter,), (0.0 * torch.sum(scores[imposter])).expand(len(imposter)))
    return loss


def margin_loss(scores, label, targeted=False, confidence=0.0, task="CSI", threshold=None, clip_max=True):
    """SEC4SR_MarginLoss.forward, attack/utils.py:41-102."""
    n, num_class = scores.shape
    rows = []
    conf = torch.tensor(confidence, dtype=torch.float)
    for i in range(n):
        s = scores[i]
        y = int(label[i])
        if task == "SV":
            assert y in (0, -1)
            if (y == 0) == targeted:
                l = threshold + conf - s[0]
            else:
                l = s[0] + conf - threshold
        else:
            if y != -1:
                onehot = torch.zeros(num_class)
                onehot[y] = 1
                real = torch.sum(onehot * s)
                other = torch.max((1 - onehot) * s - onehot * 10000)
                if targeted:
                    l = other + conf - real if task == "CSI" else torch.clamp(other, min=threshold) + conf - real
                elif task == "CSI":
                    l = real + conf - other
                else:
                    f_reject = torch.max(s) + conf - threshold
                    f_mis = torch.clamp(real, min=threshold) + conf - other
                    l = torch.minimum(f_reject, f_mis)
            elif task == "OSI":
                l = torch.max(s) + conf - threshold if targeted else threshold + conf - torch.max(s)
            else:
                l = 0.0 * torch.sum(s)
        rows.append(l)
    loss = torch.stack(rows).float()
    if clip_max:
        loss = torch.max(torch.tensor(0, dtype=torch.float), loss)
    return loss


def resolve_loss(loss_name="Entropy", targeted=False, confidence=0.0, task="CSI", threshold=None, clip_max=True):
    assert loss_name in ["Entropy", "Margin"] and task in ["CSI", "SV", "OSI"]
    if task in ("SV", "OSI") or loss_name == "Margin":
        loss = lambda s, y: margin_loss(s, y, targeted, confidence, task, threshold, clip_max)
    else:
        loss = cross_entropy_loss
    grad_sign = (1 - 2 * int(targeted)) if loss_name == "Entropy" else -1
    return loss, grad_sign


def resolve_prediction(decisions):
    return np.array([Counter(d).most_common(1)[0][0] for d in decisions])


def compare(y, y_pred, targeted):
    return (y_pred == y).tolist() if targeted else (y_pred != y).tolist()


# ----------------------------------------------------------------------------- EOT
def eot_forward(model, loss_fn, x_batch, y_batch, num_batches, batch_size, use_grad):
    """adaptive_attack/EOT.py:16-54; returns SUMS over EOT batches like the reference."""
    n_audios, n_channels, max_len = x_batch.shape
    grad = None
    scores = None
    loss = None
    decisions = [[] for _ in range(n_audios)]
    for _ in range(num_batches):
        x_rep = x_batch.repeat(batch_size, 1, 1)
        if use_grad:
            x_rep.retain_grad()
        y_rep = y_batch.repeat(batch_size)
        dec, sc = model.make_decision(x_rep)
        l = loss_fn(sc, y_rep)
        if use_grad:
            l.backward(torch.ones_like(l))
        s_m = sc.detach().view(batch_size, -1, sc.shape[1]).mean(0)
        l_m = l.detach().view(batch_size, -1).mean(0)
        scores = s_m if scores is None else scores + s_m
        loss = l_m if loss is None else loss + l_m
        if use_grad:
            g = x_rep.grad.view(batch_size, -1, n_channels, max_len).mean(0)
            grad = g.clone() if grad is None else grad + g
            x_rep.grad.zero_()
        dec = dec.view(batch_size, -1).detach().cpu().numpy()
        for ii in range(n_audios):
            decisions[ii] += list(dec[:, ii])
    return scores, loss, grad, decisions


class FGSM:
    def __init__(self, model, task="CSI", epsilon=0.002, loss="Entropy", targeted=False,
                 batch_size=1, EOT_size=1, EOT_batch_size=1, verbose=0):
        self.model, self.task, self.epsilon = model, task, epsilon
        self.targeted, self.batch_size = targeted, batch_size
        self.EOT_size, self.EOT_batch_size = max(1, EOT_size), max(1, EOT_batch_size)
        assert self.EOT_size % self.EOT_batch_size == 0
        self.threshold = model.threshold if task in ("SV", "OSI") else None
        self.loss, self.grad_sign = resolve_loss(loss, targeted, 0.0, task, self.threshold, False)
        self.max_iter, self.step_size = 1, epsilon
        self.trace = None  # optional list collecting per-iteration (loss, predict)

    def attack_batch(self, x_batch, y_batch, lower, upper, batch_id=0):
        x_batch = x_batch.clone()
        x_batch.requires_grad = True
        success = None
        for it in range(self.max_iter + 1):
            nb = self.EOT_size // self.EOT_batch_size if it < self.max_iter else 1
            bs = self.EOT_batch_size if it < self.max_iter else 1
            use_grad = it < self.max_iter
            scores, loss, grad, decisions = eot_forward(self.model, self.loss, x_batch, y_batch, nb, bs, use_grad)
            loss = loss / nb
            predict = resolve_prediction(decisions)
            success = compare(y_batch.numpy(), predict, self.targeted)
            if self.trace is not None:
                self.trace.append((loss.numpy().copy(), predict.copy()))
            if use_grad:
                grad = grad / nb
                x_batch.data += self.step_size * torch.sign(grad) * self.grad_sign
                x_batch.data = torch.min(torch.max(x_batch.data, lower), upper)
        return x_batch.detach(), success

    def _chunks(self, x, y, lower, upper):
        n = x.shape[0]
        bs = min(self.batch_size, n)
        adver, success = [], []
        for b in range(int(np.ceil(n / float(bs)))):
            sl = slice(b * bs, (b + 1) * bs)
            a, s = self.attack_batch(x[sl], y[sl], lower[sl], upper[sl], b)
            adver.append(a)
            success += s
        return torch.cat(adver, 0), success

    def attack(self, x, y):
        assert -1 <= x.max() < 1
        assert x.shape[1] == 1 and y.shape[0] == x.shape[0]
        return self._chunks(x, y, torch.full_like(x, -1.0), torch.full_like(x, 1.0))


class PGD(FGSM):
    def __init__(self, model, task="CSI", epsilon=0.002, step_size=0.0004, max_iter=10, num_random_init=0,
                 loss="Entropy", targeted=False, batch_size=1, EOT_size=1, EOT_batch_size=1, verbose=0):
        super().__init__(model, task, epsilon, loss, targeted, batch_size, EOT_size, EOT_batch_size, verbose)
        self.step_size, self.max_iter, self.num_random_init = step_size, max_iter, num_random_init

    def attack(self, x, y):
        assert -1 <= x.max() < 1
        assert x.shape[1] == 1 and y.shape[0] == x.shape[0]
        upper = torch.clamp(x + self.epsilon, max=1)
        lower = torch.clamp(x - self.epsilon, min=-1)
        x_ori = x.clone()
        best_rate, best_success, best_adver = -1, None, None
        for _ in range(max(1, self.num_random_init)):
            if self.num_random_init > 0:
                x = x_ori + torch.tensor(np.random.uniform(-self.epsilon, self.epsilon, tuple(x.shape)), dtype=x.dtype)
            adver, success = self._chunks(x, y, lower, upper)
            if sum(success) / len(success) > best_rate:
                best_rate, best_success, best_adver = sum(success) / len(success), success, adver
        return best_adver, best_success


class CWinf(PGD):
    def __init__(self, model, **kw):
        kw["loss"] = "Margin"
        super().__init__(model, **kw)


class CW2(FGSM):
    def __init__(self, model, task="CSI", targeted=False, confidence=0.0, initial_const=1e-3,
                 binary_search_steps=9, max_iter=10000, stop_early=True, stop_early_iter=1000, lr=1e-2,
                 batch_size=1, verbose=0):
        self.model, self.task, self.targeted, self.confidence = model, task, targeted, confidence
        self.initial_const, self.binary_search_steps, self.max_iter = initial_const, binary_search_steps, max_iter
        self.stop_early, self.stop_early_iter, self.lr, self.batch_size = stop_early, stop_early_iter, lr, batch_size
        self.threshold = model.threshold if task in ("SV", "OSI") else None
        self.loss = lambda s, y: margin_loss(s, y, targeted, confidence, task, self.threshold, True)

    def attack_batch(self, x_batch, y_batch, lower, upper, batch_id=0):
        n = x_batch.shape[0]
        const = torch.tensor([self.initial_const] * n, dtype=torch.float)
        lower_bound = torch.zeros(n)
        upper_bound = torch.full((n,), 1e10)
        global_best_l2 = [np.inf] * n
        global_best_adver = x_batch.clone()
        global_best_score = [-2] * n
        for _ in range(self.binary_search_steps):
            modifier = torch.zeros_like(x_batch, requires_grad=True)
            opt = torch.optim.Adam([modifier], lr=self.lr)
            best_l2 = [np.inf] * n
            best_score = [-2] * n
            cont = True
            prev_loss = np.inf
            for n_iter in range(self.max_iter + 1):
                if not cont:
                    break
                input_x = torch.tanh(modifier + torch.atanh(x_batch * 0.999999))
                decisions, scores = self.model.make_decision(input_x)
                loss1 = self.loss(scores, y_batch)
                loss2 = torch.sum(torch.square(input_x - x_batch), dim=(1, 2))
                loss = const * loss1 + loss2
                if n_iter < self.max_iter:
                    loss.backward(torch.ones_like(loss))
                    opt.step()
                    modifier.grad.zero_()
                predict = decisions.detach().numpy()
                loss_l = loss.detach().numpy().tolist()
                l1s = loss1.detach().numpy().tolist()
                l2s = loss2.detach().numpy().tolist()
                if self.stop_early and n_iter % self.stop_early_iter == 0:
                    if np.mean(loss_l) > 0.9999 * prev_loss:
                        cont = False
                    prev_loss = np.mean(loss_l)
                for ii, (l2, yp, ax, l1) in enumerate(zip(l2s, predict, input_x, l1s)):
                    if l1 <= 0 and l2 < best_l2[ii]:
                        best_l2[ii], best_score[ii] = l2, yp
                    if l1 <= 0 and l2 < global_best_l2[ii]:
                        global_best_l2[ii], global_best_score[ii] = l2, yp
                        global_best_adver[ii] = ax.detach()
            for jj, yp in enumerate(best_score):
                if yp != -2:
                    upper_bound[jj] = min(upper_bound[jj], const[jj])
                    if upper_bound[jj] < 1e9:
                        const[jj] = (lower_bound[jj] + upper_bound[jj]) / 2
                else:
                    lower_bound[jj] = max(lower_bound[jj], const[jj])
                    if upper_bound[jj] < 1e9:
                        const[jj] = (lower_bound[jj] + upper_bound[jj]) / 2
                    else:
                        const[jj] *= 10
        success = [s != -2 for s in global_best_score]
        return global_best_adver, success


# ----------------------------------------------------------------------------- NES / FAKEBOB
def nes_forward(model, loss_fn, x, y, samples_per_draw, samples_batch, sigma, eot_size=1, eot_batch=1, noise_fn=None):
    """adaptive_attack/NES.py:15-56.  ``noise_fn(shape)`` defaults to torch.randn (global RNG)."""
    noise_fn = noise_fn or (lambda shape: torch.randn(shape))
    n, c, N = x.shape
    num_batches = samples_per_draw // samples_batch
    eot_nb = eot_size // eot_batch
    for i in range(num_batches):
        noise = noise_fn([n, samples_batch // 2, c, N])
        noise = torch.cat((noise, -noise), 1)
        if i == 0:
            noise = torch.cat((torch.zeros_like(x).unsqueeze(1), noise), 1)
        eval_input = (noise * sigma + x.unsqueeze(1)).view(-1, c, N)
        per = samples_batch + 1 if i == 0 else samples_batch
        eval_y = torch.cat([torch.tensor([int(y_)] * per, dtype=torch.long) for y_ in y])
        scores, loss, _, decisions = eot_forward(model, loss_fn, eval_input, eval_y, eot_nb, eot_batch, False)
        loss = (loss / eot_nb).view(n, -1)
        scores = (scores / eot_nb).view(n, -1, scores.shape[1])
        if i == 0:
            adver_loss = loss[..., 0]
            loss = loss[..., 1:]
            adver_score = scores[:, 0, :]
            noise = noise[:, 1:, :, :]
            grad = torch.mean(loss.unsqueeze(2).unsqueeze(3) * noise, 1)
            mean_loss = loss.mean(1)
            predict = resolve_prediction(decisions).reshape(n, -1)[:, 0]
        else:
            grad = grad + torch.mean(loss.unsqueeze(2).unsqueeze(3) * noise, 1)
            mean_loss = mean_loss + loss.mean(1)
    return mean_loss / num_batches, grad / sigma / num_batches, adver_loss, adver_score, predict


class FAKEBOB:
    def __init__(self, model, threshold=None, task="CSI", targeted=False, confidence=0.0, epsilon=0.002,
                 max_iter=1000, max_lr=0.001, min_lr=1e-6, samples_per_draw=50, samples_per_draw_batch_size=50,
                 sigma=0.001, momentum=0.9, plateau_length=5, plateau_drop=2.0, stop_early=True,
                 stop_early_iter=100, batch_size=1, EOT_size=1, EOT_batch_size=1, verbose=0, noise_fn=None):
        self.__dict__.update(locals())
        del self.__dict__["self"]

    def get_grad(self, x, y):
        return nes_forward(self.model, self.loss, x, y, self.samples_per_draw, self.samples_per_draw_batch_size,
                           self.sigma, self.EOT_size, self.EOT_batch_size, self.noise_fn)

    @staticmethod
    def delete_found(flags, tensors, lists):
        """FAKEBOB.delete_found :125-168: drop examples whose flag value is < 0."""
        keep = [i for i, f in enumerate(flags) if not (f < 0)]
        if not keep:
            return None, None
        return [t[keep] for t in tensors], [[l[i] for i in keep] for l in lists]

    def attack_batch(self, x_batch, y_batch, lower, upper, batch_id=0):
        with torch.no_grad():
            n = x_batch.shape[0]
            last_ls = [[]] * n
            lr = [self.max_lr] * n
            prev_loss = [np.inf] * n
            adver_x = x_batch.clone()
            grad = torch.zeros_like(x_batch)
            best_adver_x = adver_x.clone()
            best_loss = [np.inf] * n
            consider = list(range(n))
            for it in range(self.max_iter + 1):
                prev_grad = grad.clone()
                loss, grad, adver_loss, _, _ = self.get_grad(adver_x, y_batch)
                for ii, al in enumerate(adver_loss):
                    idx = consider[ii]
                    if al < best_loss[idx]:
                        best_loss[idx] = al.item()
                        best_adver_x[idx] = adver_x[ii]
                ts, ls = self.delete_found(adver_loss, [adver_x, y_batch, prev_grad, grad, lower, upper],
                                           [consider, last_ls, lr, prev_loss, list(loss)])
                if ts is None:
                    break
                adver_x, y_batch, prev_grad, grad, lower, upper = ts
                consider, last_ls, lr, prev_loss, loss = ls
                if it < self.max_iter:
                    grad = self.momentum * prev_grad + (1.0 - self.momentum) * grad
                    for jj, l_ in enumerate(loss):
                        # NB `last_ls = [[]] * n` (FAKEBOB.py:56) aliases ONE list across examples, so the
                        # in-place append below leaks earlier examples' losses into later histories
                        # until each entry is rebound; kept on purpose -- it changes the LR schedule.
                        last_ls[jj].append(l_)
                        last_ls[jj] = last_ls[jj][-self.plateau_length:]
                        if last_ls[jj][-1] > last_ls[jj][0] and len(last_ls[jj]) == self.plateau_length:
                            if lr[jj] > self.min_lr:
                                lr[jj] = max(lr[jj] / self.plateau_drop, self.min_lr)
                            last_ls[jj] = []
                    lr_t = torch.tensor(lr, dtype=torch.float).view(-1, 1, 1)
                    adver_x = adver_x + self.grad_sign * lr_t * torch.sign(grad)
                    adver_x = torch.min(torch.max(adver_x, lower), upper)
                    if self.stop_early and it % self.stop_early_iter == 0:
                        loss_np = np.array([float(l) for l in loss])
                        conv = np.array(prev_loss) * 0.9999 - loss_np
                        ts, ls = self.delete_found(conv, [adver_x, y_batch, prev_grad, grad, lower, upper],
                                                   [consider, last_ls, lr, list(loss_np), list(loss)])
                        if ts is None:
                            break
                        adver_x, y_batch, prev_grad, grad, lower, upper = ts
                        consider, last_ls, lr, prev_loss, loss = ls
            return best_adver_x, [bl < 0 for bl in best_loss]

    def attack(self, x, y):
        if self.task in ("SV", "OSI") and self.threshold is None:
            raise NotImplementedError("threshold not specified")
        self.loss, self.grad_sign = resolve_loss("Margin", self.targeted, self.confidence, self.task, self.threshold, False)
        assert -1 <= x.max() < 1 and x.shape[1] == 1 and y.shape[0] == x.shape[0]
        upper = torch.clamp(x + self.epsilon, max=1)
        lower = torch.clamp(x - self.epsilon, min=-1)
        n = x.shape[0]
        bs = min(self.batch_size, n)
        adver, success = [], []
        for b in range(int(np.ceil(n / float(bs)))):
            sl = slice(b * bs, (b + 1) * bs)
            a, s = self.attack_batch(x[sl], y[sl], lower[sl], upper[sl], b)
            adver.append(a)
            success += s
        return torch.cat(adver, 0), success

    # FAKEBOB.estimate_threshold_run :210-278 / estimate_threshold :280-295
    def estimate_threshold_run(self, x, step=0.1):
        with torch.no_grad():
            n = x.shape[0]
            d, s = self.model.make_decision(x)
            if int(d[0]) != -1:
                return None
            y = torch.full((n,), -1, dtype=torch.long)
            init_score = float(np.max(s[0].numpy()))
            delta = np.abs(init_score * step)
            threshold = init_score + delta
            adver_x = x.clone()
            grad = torch.zeros_like(x)
            upper = torch.clamp(x + self.epsilon, max=1)
            lower = torch.clamp(x - self.epsilon, min=-1)
            while True:
                self.loss, self.grad_sign = resolve_loss("Margin", False, 0.0, self.task, threshold, False)
                last_ls = [[]] * n  # aliased on purpose, see attack_batch
                lr = [self.max_lr] * n
                while True:
                    decision, score = self.model.make_decision(adver_x)
                    top = float(np.max(score[0].numpy()))
                    if int(decision[0]) != -1:
                        return top
                    elif top >= threshold:
                        break
                    prev_grad = grad.clone()
                    loss, grad, _, _, _ = self.get_grad(adver_x, y)
                    grad = self.momentum * prev_grad + (1.0 - self.momentum) * grad
                    for jj, l_ in enumerate(loss):
                        last_ls[jj].append(l_)
                        last_ls[jj] = last_ls[jj][-self.plateau_length:]
                        if last_ls[jj][-1] > last_ls[jj][0] and len(last_ls[jj]) == self.plateau_length:
                            if lr[jj] > self.min_lr:
                                lr[jj] = max(lr[jj] / self.plateau_drop, self.min_lr)
                            last_ls[jj] = []
                    lr_t = torch.tensor(lr, dtype=torch.float).view(-1, 1, 1)
                    adver_x = adver_x + self.grad_sign * lr_t * torch.sign(grad)
                    adver_x = torch.min(torch.max(adver_x, lower), upper)
                threshold += delta

    def estimate_threshold(self, x, step=0.1):
        if self.task == "CSI":
            return None
        est = []
        for xx in x.unsqueeze(0):  # sic (FAKEBOB.py:287): a single pass with the whole batch
            t = self.estimate_threshold_run(xx, step)
            if t is not None:
                est.append(t)
        self.threshold = np.mean(est) if est else None
        return self.threshold
