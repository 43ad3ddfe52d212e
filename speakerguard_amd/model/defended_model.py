"""Defense wrapper; mirrors reference model/defended_model.py.

With ``defense=None`` (reference :127-128,156-157) every call is passed straight to the base model,
including the native ``loss_grad`` / ``pgd_run`` entry points the attacks use.  Input- and
feature-level defenses are applied for the forward calls exactly like ``process_sequential``
(:46-65).  The gradient THROUGH feature-level defenses is chained by hand for
defenses that expose ``fwd`` / ``bwd`` (``speakerguard_amd.defense.feature_level.FeCoDefense`` at the feature levels,
``speakerguard_amd.adaptive_attack.BPDA.BPDA(f, sub_f)`` -- the reference's straight-through wrapper, BPDA.py:7-65 -- around any
input-level transform) on the native xv_plda / audionet_csine base models in sequential order; any other defended
configuration raises in ``loss_grad`` instead of silently ignoring the defense.
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

    @property
    def batch_coupled(self):
        """True if a defense's result depends on how many utterances share a model call (FeCo: feature_level.py:33)."""
        return any(getattr(method, 'batch_coupled', False) for _, method in (self.defense or []))

    def _fwd(self, d, xx):
        """``d.fwd(xx)``; a randomised defense (FeCoDefense(init='random')) takes its generator key from the base
        model's noise bookkeeping (attack call, chunk, call number) instead of its own call counter."""
        if getattr(d, 'init', None) == 'random' and hasattr(self.base_model, 'defense_seed'):
            return d.fwd(xx, seed=self.base_model.defense_seed(d.seed), row_keys=self.base_model.row_keys())
        return d.fwd(xx)

    def _apply(self, d, xx):
        return self._fwd(d, xx)[0] if hasattr(d, 'fwd') else d(xx)

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
                xx = self._apply(d, xx)
        return xx

    def _last_flag(self):
        return sorted(self.flag2defense.keys())[-1]

    def _average(self, fn, x):
        acc = None
        for flag in sorted(self.flag2defense.keys()):
            xx = x.clone() if flag == 0 else self.base_model.compute_feat(x, flag=flag)
            for d in self.flag2defense[flag]:
                out = fn(self._apply(d, xx), flag)
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
        if not want_grad:
            decisions, scores = self.make_decision(x)
            return decisions, scores, loss_spec(scores, y), None
        if self.order == average:
            return self._loss_grad_average(x, y, loss_spec)
        return self._loss_grad_through_defenses(x, y, loss_spec)

    def _loss_grad_average(self, x, y, loss_spec):
        """'average' order (:67-75 `_average`): the model scores every defended copy d_i(x) on its own and the loss is
        taken of the MEAN score.  d loss / d x = sum_i J_i^T (g / n) with g = d loss / d mean-score: the loss stage runs
        once on the mean (sg_loss_eval) and every branch is a vector-Jacobian product of its scores (ScoreVJP) chained
        back through its defense and the front-end stages below its level by hand."""
        from ..attack.utils import ScoreVJP, loss_dscores
        bm = self.base_model
        branches = [(flag, d) for flag in sorted(self.flag2defense.keys()) for d in self.flag2defense[flag]]
        if not (hasattr(bm, 'frontend_forward') and all(hasattr(d, 'fwd') and hasattr(d, 'bwd') for _, d in branches)):
            raise NotImplementedError('gradient through this defense configuration is not built: needs a native base '
                                      'model and defenses exposing fwd/bwd (FeCoDefense at the feature levels; '
                                      'adaptive_attack.BPDA.BPDA(f, sub_f) around any input-level transform)')
        n = len(self.defense)  # :75 divides by len(self.defense)
        x = x.to(bm.device, torch.float32).contiguous()
        feats = saved_front = cm = None
        if any(f >= 1 for f, _ in branches):
            feats, saved_front = bm.frontend_forward(x)  # one front-end pass serves every feature-level branch
        if any(f == 2 for f, _ in branches):
            cm = bm.comput_feat_from_feat(feats, ori_flag=1, des_flag=2)
        tape, mean = [], None
        for flag, d in branches:
            out, sv = self._fwd(d, x if flag == 0 else (feats if flag == 1 else cm))
            # a waveform-level branch runs the (dithered) front-end inside the model: ONE noise realisation serves its
            # scores here and its gradient below, like the single autograd graph of the reference (EOT.py:32-35)
            key = bm.next_dither_seed() if flag == 0 and hasattr(bm, 'next_dither_seed') else None
            sc = bm.forward(out, flag=flag, dither_seed=key) if key is not None else bm.forward(out, flag=flag)
            mean = sc if mean is None else mean + sc
            tape.append((flag, d, out, sv, key))
        mean = mean / n
        decisions, loss, g = loss_dscores(bm, mean, y, loss_spec)
        vjp = ScoreVJP(g / n)
        y0 = torch.zeros(x.shape[0], device=bm.device, dtype=torch.int64)
        grad, dfeat = None, None
        for flag, d, out, sv, key in tape:
            kw = {} if key is None else {'dither_seed': key}
            gi = d.bwd(sv, bm.loss_grad(out, y0, vjp, flag=flag, want_grad=True, **kw)[3])
            if flag == 0:
                grad = gi if grad is None else grad + gi
            else:
                if flag == 2:
                    gi = bm.cmvn_backward(gi)
                dfeat = gi if dfeat is None else dfeat + gi  # the front-end is linear in its cotangent: one adjoint below
        if dfeat is not None:
            gw = bm.frontend_backward(saved_front, dfeat)
            grad = gw if grad is None else grad + gw
        return decisions, mean, loss, grad

    def _loss_grad_through_defenses(self, x, y, loss_spec):
        """wav -> MFCC -> [flag-1 defenses] -> CMVN -> [flag-2 defenses] -> TDNN ... -> loss, and back.

        The order is process_sequential's (:52-63).  Each stage's backward is the native one: the model's own
        loss_grad from the last defended level, the defenses' ``bwd``, CMVN / MFCC backward in between."""
        bm = self.base_model
        chainable = (self.order == sequential and hasattr(bm, 'frontend_forward')
                     and all(hasattr(d, 'fwd') and hasattr(d, 'bwd') for f in (0, 1, 2) for d in self.flag2defense.get(f, [])))
        if not chainable:
            raise NotImplementedError('gradient through this defense configuration is not built: needs a native base '
                                      'model, sequential order, and defenses exposing fwd/bwd (FeCoDefense at the feature '
                                      'levels; adaptive_attack.BPDA.BPDA(f, sub_f) around any input-level transform)')
        # input-level defenses (flag 0: wav -> wav), e.g. BPDA-wrapped quantisation (defense/time_domain.py:44)
        tape0 = []
        for d in self.flag2defense.get(0, []):
            x, sv = self._fwd(d, x)
            tape0.append((d, sv))
        if not self.flag2defense.get(1) and not self.flag2defense.get(2):
            # nothing sits between front-end and network: one native call from the (defended) waveform
            decisions, scores, loss, g = bm.loss_grad(x, y, loss_spec, flag=0, want_grad=True)
            for d, sv in reversed(tape0):
                g = d.bwd(sv, g)
            return decisions, scores, loss, g
        feats, saved_front = bm.frontend_forward(x)
        tape1, tape2 = [], []
        for d in self.flag2defense[1]:
            feats, sv = self._fwd(d, feats)
            tape1.append((d, sv))
        if self.flag2defense.get(2):  # xv_plda only (AudioNet has no level 2, audionet_csine.py:127-129)
            feats = bm.comput_feat_from_feat(feats, ori_flag=1, des_flag=2)
            for d in self.flag2defense[2]:
                feats, sv = self._fwd(d, feats)
                tape2.append((d, sv))
            decisions, scores, loss, g = bm.loss_grad(feats, y, loss_spec, flag=2, want_grad=True)
            for d, sv in reversed(tape2):
                g = d.bwd(sv, g)
            g = bm.cmvn_backward(g)
        else:
            decisions, scores, loss, g = bm.loss_grad(feats, y, loss_spec, flag=1, want_grad=True)
        for d, sv in reversed(tape1):
            g = d.bwd(sv, g)
        g = bm.frontend_backward(saved_front, g)
        for d, sv in reversed(tape0):
            g = d.bwd(sv, g)
        return decisions, scores, loss, g

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
