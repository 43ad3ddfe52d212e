"""Defense wrapper; mirrors reference model/defended_model.py.

With ``defense=None`` (reference :127-128,156-157) every call is passed straight to the base model,
including the native ``loss_grad`` / ``pgd_run`` entry points the attacks use.  Input- and
feature-level defenses are applied for the forward calls exactly like ``process_sequential``
(:46-65); differentiating THROUGH a defense is not part of this round and
``loss_grad`` raises instead of silently ignoring the defense.
"""
import warnings

import torch

sequential = 'sequential'  # model(d_n(...d_2(d_1(x))))
average = 'average'        # average(model(d_1(x)), ..., model(d_n(x)))


class defended_model:

    def __init__(self, base_model, defense=None, order=sequential):
        self.base_model = base_model
        self.threshold = base_model.threshold
        if defense is not None:
            flag2defense = {flag: [] for flag in self.base_model.allowed_flags}
            assert isinstance(defense, (list, tuple))
            assert order in [sequential, average]
            prev_flag = -1
            for flag_method in defense:
                assert isinstance(flag_method, (list, tuple)) and len(flag_method) == 2
                flag, method = flag_method
                if flag not in self.base_model.allowed_flags:
                    warnings.warn('Unsupported Input Level Flag. Ignore the Defense!')
                    continue
                flag2defense[flag].append(method)
                if order == sequential:
                    if flag < prev_flag:
                        warnings.warn('You want to combine multiple defenses in sequential order, but the order of your defense is wrong. Re-arranged.')
                    prev_flag = flag
            self.order = order
            self.flag2defense = flag2defense
        self.defense = defense

    def eval(self):
        return self

    def process_sequential(self, x):
        if self.defense is None:
            return x
        for flag in sorted(self.flag2defense.keys()):
            if flag == 0:
                xx = x.clone()
            elif flag == 1:
                xx = self.base_model.compute_feat(xx, flag=1)
            else:
                xx = self.base_model.comput_feat_from_feat(xx, ori_flag=flag - 1, des_flag=flag)
            for d in self.flag2defense[flag]:
                xx = d(xx)
        return xx

    def _last_flag(self):
        return sorted(self.flag2defense.keys())[-1]

    def _average(self, fn, x):
        acc = None
        for flag in sorted(self.flag2defense.keys()):
            xx = x.clone() if flag == 0 else self.base_model.compute_feat(x, flag=flag)
            for d in self.flag2defense[flag]:
                out = fn(d(xx), flag)
                acc = out if acc is None else tuple(a + o for a, o in zip(acc, out))
        return tuple(a / len(self.defense) for a in acc)

    def embedding(self, x):
        if self.defense is None:
            return self.base_model.embedding(x, flag=0)
        if self.order == sequential:
            return self.base_model.embedding(self.process_sequential(x), flag=self._last_flag())
        return self._average(lambda xx, f: (self.base_model.embedding(xx, flag=f),), x)[0]

    def forward(self, x, return_emb=False, enroll_embs=None):
        if self.defense is None:
            return self.base_model(x, flag=0, return_emb=return_emb, enroll_embs=enroll_embs)
        if self.order == sequential:
            return self.base_model(self.process_sequential(x), flag=self._last_flag(), return_emb=return_emb, enroll_embs=enroll_embs)
        logits, emb = self._average(lambda xx, f: self.base_model(xx, flag=f, return_emb=True, enroll_embs=enroll_embs), x)
        return (logits, emb) if return_emb else logits

    __call__ = forward

    def score(self, x, enroll_embs=None):
        if self.defense is None:
            return self.base_model.score(x, flag=0, enroll_embs=enroll_embs)
        if self.order == sequential:
            return self.base_model.score(self.process_sequential(x), flag=self._last_flag(), enroll_embs=enroll_embs)
        return self._average(lambda xx, f: (self.base_model.score(xx, flag=f, enroll_embs=enroll_embs),), x)[0]

    def make_decision(self, x, enroll_embs=None):
        scores = self.score(x, enroll_embs=enroll_embs)
        decisions = torch.argmax(scores, dim=1)
        max_scores = torch.max(scores, dim=1)[0]
        decisions = torch.where(max_scores > self.base_model.threshold, decisions, torch.full_like(decisions, -1))
        return decisions, scores

    # ---- engine protocol ---------------------------------------------------------------------
    def loss_grad(self, x, y, loss_spec, want_grad=True):
        if self.defense is None:
            return self.base_model.loss_grad(x, y, loss_spec, flag=0, want_grad=want_grad)
        if want_grad:
            raise NotImplementedError('gradient through defenses is not implemented in this round (SURVEY.md N1)')
        decisions, scores = self.make_decision(x)
        return decisions, scores, loss_spec(scores, y), None

    def pgd_update(self, *a, **k):
        return self.base_model.pgd_update(*a, **k)

    def cw2_step(self, *a, **k):
        return self.base_model.cw2_step(*a, **k)

    def nes_queries(self, *a, **k):
        return self.base_model.nes_queries(*a, **k)

    def nes_grad(self, *a, **k):
        return self.base_model.nes_grad(*a, **k)

    def fakebob_step(self, *a, **k):
        return self.base_model.fakebob_step(*a, **k)
