"""AudioNet CSI-NE on the HIP engine; host-side mirror of reference model/audionet_csine.py.

Same constructor arguments (``extractor_file``, ``num_class``, ``label_encoder``, ``device``), the
same methods (``compute_feat / embedding / forward / score / make_decision``) and attributes
(``threshold = -inf``, ``allowed_flags = [0, 1]``, ``range_type = 'scale'``, ``spk_ids``).  Inference
only: the reference's training branch (``extractor_file=None``) has no counterpart here.
"""
import ctypes as C

import numpy as np
import torch

from .. import _native as N
from ._engine_ops import EngineOps

BITS = 16
_CONVS = ("conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv8")


def _f32(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


class audionet_csine(EngineOps):
    allowed_flags = [0, 1]  # 0: wav; 1: raw (log-mel) feat -- audionet_csine.py:127-129
    range_type = "scale"

    def __init__(self, extractor_file=None, num_class=None, label_encoder=None, device="cuda:0"):
        if extractor_file is None:
            raise N.NativeError("the engine is inference-only: pass a pre-trained extractor_file (state_dict)")
        sd = torch.load(extractor_file, map_location="cpu") if isinstance(extractor_file, str) else extractor_file
        self._init(sd, label_encoder, device)
        if num_class is not None:
            assert num_class == self.num_spks

    @classmethod
    def from_weights(cls, state_dict, device="cuda:0"):
        self = cls.__new__(cls)
        self._init(state_dict, None, device)
        return self

    def _init(self, sd, label_encoder, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise N.NativeError("audionet_csine runs on the HIP engine only; device must be a GPU (got %s)" % device)
        idx = self.device.index if self.device.index is not None else 0
        self.device = torch.device("cuda", idx)
        keep = []

        def hp(a):
            a = _f32(a)
            keep.append(a)
            return a.ctypes.data_as(C.c_void_p)

        w = N.AnWeights()
        w.conv1_weight, w.conv1_bias = hp(sd["conv1.0.weight"]), hp(sd["conv1.0.bias"])
        for i, k in enumerate(("weight", "bias", "running_mean", "running_var")):
            w.bn1[i] = hp(sd["conv1.1." + k])
        for l, name in enumerate(_CONVS):
            w.conv_weight[l], w.conv_bias[l] = hp(sd[name + ".0.weight"]), hp(sd[name + ".0.bias"])
            w.bn_weight[l], w.bn_bias[l] = hp(sd[name + ".1.weight"]), hp(sd[name + ".1.bias"])
            w.bn_mean[l], w.bn_var[l] = hp(sd[name + ".1.running_mean"]), hp(sd[name + ".1.running_var"])
        w.fc_weight, w.fc_bias = hp(sd["fc.weight"]), hp(sd["fc.bias"])
        self.num_spks = int(_f32(sd["fc.bias"]).shape[0])
        w.num_class, w.bn_eps = self.num_spks, 1e-5
        self.ctx = N.Context(idx)
        self.ctx.call("sg_an_load", C.byref(w))
        self.threshold = -np.inf  # CSI-NE: never rejects (audionet_csine.py:126)
        if label_encoder is not None:  # id <-> label table, audionet_csine.py:36-46
            rows = np.loadtxt(label_encoder, dtype=str, converters={0: lambda s: s[1:-1]})
            label2id = {int(r[1]): r[0] for r in rows}
            self.spk_ids = [label2id[i] for i in range(len(label2id))]
            assert len(self.spk_ids) == self.num_spks
        else:
            self.spk_ids = [str(i) for i in range(self.num_spks)]
        self.dither = 0.0  # the AudioNet front-end has no dither

    def eval(self):
        return self

    def configure_frontend(self, fft_bits=32, spectrum_cache=None, fused_overlap_add=None):
        """How the STFT front-end runs on the engine (sg_an_configure): float32 transforms are the reference's own precision
        (_audionet/Preprocessor.py:100-105); 64 is the float64 form of rounds 1-4, kept as the counterpart.
        spectrum_cache / fused_overlap_add None: the engine picks per call from the batch's size."""
        ola = -1 if fused_overlap_add is None else int(bool(fused_overlap_add))
        cache = -1 if spectrum_cache is None else int(bool(spectrum_cache))
        self.ctx.call("sg_an_configure", int(fft_bits), cache, ola)
        return self

    def _prep(self, x, flag):
        assert flag in self.allowed_flags
        x = x.to(self.device, torch.float32).contiguous()
        if flag == 0:
            assert x.dim() == 3 and x.shape[1] == 1, "wav input must be (B, 1, T)"
            return x, x.shape[0], x.shape[2]
        assert x.dim() == 3 and x.shape[2] == 32, "feature input must be (B, F, 32)"
        return x, x.shape[0], x.shape[1]

    def compute_feat(self, x, flag=1):
        """wav (B,1,T) -> log-mel (B,F,32); audionet_csine.py:133-146."""
        assert flag == 1
        x, B, T = self._prep(x, 0)
        F = N.load().sg_an_num_frames(T)
        feats = torch.empty(B, F, 32, device=self.device, dtype=torch.float32)
        self.ctx.call("sg_an_logmel", N._ptr(x), B, T, N._ptr(feats), self._stream())
        return feats

    def raw(self, x):
        return self.compute_feat(x, 1)

    # ---- front-end with its backward (used when a feature-level defense sits behind it) -------------
    def frontend_forward(self, x):
        x, B, T = self._prep(x, 0)
        return self.compute_feat(x, 1), x

    def frontend_backward(self, saved, dfeats):
        """d loss / d log-mel (B,F,32) -> d loss / d wav (B,1,T)."""
        x = saved
        dfeats = dfeats.to(self.device, torch.float32).contiguous()
        grad = torch.empty_like(x)
        # reuse_forward = 1: `saved` is the tensor frontend_forward just transformed; if another pass has used the
        # context's mel cache in between (pointer / shape no longer match) the library recomputes the forward by itself
        self.ctx.call("sg_an_logmel_backward", N._ptr(x), x.shape[0], x.shape[2], N._ptr(dfeats), N._ptr(grad), 1,
                      self._stream())
        return grad

    def _forward(self, x, flag, want_emb=False):
        x, B, TF = self._prep(x, flag)
        dec = torch.empty(B, device=self.device, dtype=torch.int64)
        scores = torch.empty(B, self.num_spks, device=self.device, dtype=torch.float32)
        emb = torch.empty(B, 32, device=self.device, dtype=torch.float32) if want_emb else None
        self.ctx.call("sg_an_forward", N._ptr(x), B, TF, flag, N._ptr(dec), N._ptr(scores), N._ptr(emb), self._stream())
        return dec, scores, emb

    def embedding(self, x, flag=0):
        return self._forward(x, flag, want_emb=True)[2]

    def forward(self, x, flag=0, return_emb=False, enroll_embs=None):
        _, scores, emb = self._forward(x, flag, want_emb=return_emb)
        return (scores, emb) if return_emb else scores

    __call__ = forward

    def score(self, x, flag=0, enroll_embs=None):
        return self.forward(x, flag=flag)

    def make_decision(self, x, flag=0, enroll_embs=None):
        dec, scores, _ = self._forward(x, flag)
        return dec, scores

    def read_activation(self, layer, B):
        rows, ch = C.c_int32(), C.c_int32()
        self.ctx.call("sg_an_debug_activation", layer, None, 0, C.byref(rows), C.byref(ch), self._stream())
        out = torch.empty(B, rows.value, ch.value, device=self.device, dtype=torch.float32)
        self.ctx.call("sg_an_debug_activation", layer, N._ptr(out), out.numel(), None, None, self._stream())
        return out

    # ---- engine protocol used by attack.*
    def loss_grad(self, x, y, loss_spec, flag=0, want_grad=True):
        x, B, TF = self._prep(x, flag)
        y = y.to(self.device, torch.int64).contiguous()
        dec = torch.empty(B, device=self.device, dtype=torch.int64)
        scores = torch.empty(B, self.num_spks, device=self.device, dtype=torch.float32)
        loss = torch.empty(B, device=self.device, dtype=torch.float32)
        grad = torch.empty_like(x) if want_grad else None
        if hasattr(loss_spec, 'check'):
            loss_spec.check(B, self.num_spks)
        spec = loss_spec.native()
        self.ctx.call("sg_an_loss_grad", N._ptr(x), N._ptr(y), B, TF, flag, C.byref(spec), N._ptr(dec), N._ptr(scores),
                      N._ptr(loss), N._ptr(grad), self._stream())
        return dec, scores, loss, grad

    # per-pass generator keys of the fused FeCo loop (sg_an_pgd_run_feco): step `it`, EOT repeat `r`
    @staticmethod
    def fused_pass_seed(base_seed, it, r=0):
        return (int(base_seed) + it * 0x9E3779B97F4A7C15 + r * 0xC2B2AE3D27D4EB4F) & 0xFFFFFFFFFFFFFFFF

    def pgd_run_feco(self, x, y, lower, upper, loss_spec, step_size, max_iter, grad_sign, feco, eot_size=1, eot_batch_size=1,
                     trace=False):
        """attack/FGSM.py:38-70 attack_batch against defended_model(self, [(1, feco)]) (BASELINE.json configs[3]) as one
        device-resident loop: log-mel -> FeCo -> CNN forward, hand-chained backward, EOT repeats over the defense's
        random initial frames (``feco.init == 'random'``) summed on the device.  `feco`: a FeCoDefense."""
        x, B, T = self._prep(x, 0)
        x_adv = x.clone()
        y = y.to(self.device, torch.int64).contiguous()
        lower = lower.to(self.device, torch.float32).expand_as(x).contiguous()
        upper = upper.to(self.device, torch.float32).expand_as(x).contiguous()
        if hasattr(loss_spec, "check"):
            loss_spec.check(B, self.num_spks)  # ScoreVJP: one (B, S) table, shared by the EOT repeats of an utterance
        p = N.PgdParams()
        p.loss = loss_spec.native()
        p.step_size, p.max_iter, p.grad_sign = float(step_size), int(max_iter), int(grad_sign)
        p.eot_size, p.eot_batch_size = int(eot_size), int(eot_batch_size)
        f = N.FecoParams()
        f.k = int(N.load().sg_an_num_frames(T) * feco.param)  # feature_level.py:184
        f.max_iter = int(feco.max_iter)
        f.random_init = int(feco.init == 'random')
        # one key per fused call, from the model's noise bookkeeping (attack call, restart, call number); the passes
        # inside derive theirs from it; a row is keyed by its utterance's GLOBAL index (chunk base + row)
        f.seed = self.defense_seed(feco.seed)
        feco.calls += 1
        self.last_fused_seed = int(f.seed)
        f.index_base = int(feco.index_base) + self.row_keys()[0]
        success = torch.empty(B, device=self.device, dtype=torch.uint8)
        dec = torch.empty(B, device=self.device, dtype=torch.int64)
        scores = torch.empty(B, self.num_spks, device=self.device, dtype=torch.float32)
        loss = torch.empty(B, device=self.device, dtype=torch.float32)
        ltr = torch.empty(max_iter + 1, B, device=self.device, dtype=torch.float32) if trace else None
        dtr = torch.empty(max_iter + 1, B, device=self.device, dtype=torch.int64) if trace else None
        self.ctx.call("sg_an_pgd_run_feco", N._ptr(x_adv), N._ptr(y), N._ptr(lower), N._ptr(upper), B, T, C.byref(p),
                      C.byref(f), N._ptr(success), N._ptr(dec), N._ptr(scores), N._ptr(loss), N._ptr(ltr), N._ptr(dtr),
                      self._stream())
        return x_adv, success, dec, scores, loss, ltr, dtr

    def pgd_run(self, x, y, lower, upper, loss_spec, step_size, max_iter, grad_sign, eot_size=1, eot_batch_size=1,
                trace=False):
        x, B, T = self._prep(x, 0)
        x_adv = x.clone()
        y = y.to(self.device, torch.int64).contiguous()
        lower = lower.to(self.device, torch.float32).expand_as(x).contiguous()
        upper = upper.to(self.device, torch.float32).expand_as(x).contiguous()
        p = N.PgdParams()
        p.loss = loss_spec.native()
        p.step_size, p.max_iter, p.grad_sign = float(step_size), int(max_iter), int(grad_sign)
        p.eot_size, p.eot_batch_size = int(eot_size), int(eot_batch_size)
        success = torch.empty(B, device=self.device, dtype=torch.uint8)
        dec = torch.empty(B, device=self.device, dtype=torch.int64)
        scores = torch.empty(B, self.num_spks, device=self.device, dtype=torch.float32)
        loss = torch.empty(B, device=self.device, dtype=torch.float32)
        ltr = torch.empty(max_iter + 1, B, device=self.device, dtype=torch.float32) if trace else None
        dtr = torch.empty(max_iter + 1, B, device=self.device, dtype=torch.int64) if trace else None
        self.ctx.call("sg_an_pgd_run", N._ptr(x_adv), N._ptr(y), N._ptr(lower), N._ptr(upper), B, T, C.byref(p),
                      N._ptr(success), N._ptr(dec), N._ptr(scores), N._ptr(loss), N._ptr(ltr), N._ptr(dtr), self._stream())
        return x_adv, success, dec, scores, loss, ltr, dtr
