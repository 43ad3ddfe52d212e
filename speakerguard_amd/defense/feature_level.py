"""FeCo feature-level defense on the native engine; mirrors reference defense/feature_level.py:15-50,168-217
(method 'kmeans', distance 'L2').

``FeCo(feat, method, param, other_param)`` keeps the reference's function signature (forward only).
``FeCoDefense`` is the same transform as an object with ``fwd`` / ``bwd`` so that ``defended_model`` can chain
the hand-coded backward through it (adaptive attacks against a FeCo-defended model).

The cluster ids come from the library's k-means (contract in csrc/k_feco.hip); the reference's ids come from a
randomly initialised third-party k-means and are not reproducible, so parity with the reference exists only for
the step after the ids (cluster means + empty-cluster fallback, :204-216).  ``init='even'`` (default) starts from
evenly spaced frames: one input, one output.  ``init='random'`` starts every call from k distinct random frames like
the reference's k-means does -- a randomised defense, the case expectation-over-transformation attacks are for --
with draws that are a function of (key, utterance row) only (Philox4x32-10).  Called on its own, the key is (seed, call
number); inside ``defended_model`` it comes from the base model's noise bookkeeping -- (seed, attack call, restart, GLOBAL
index of the chunk's first utterance, call number inside the chunk), like the MFCC dither -- so an attack is reproducible
and independent of how the batch is cut into per-GPU shards.
'warped_kmeans' and the cosine distance are not built.
"""
import ctypes as C

import torch

from .. import _native as N
from ..metric.metric import _context


class FeCoDefense:
    # the reference's `force` flag follows the size of the MODEL CALL (feature_level.py:33: feat.shape[0] > 1): a call with
    # one utterance drops empty clusters, a larger one fills them in.  Whoever re-cuts a batch (shard.py) must not turn an
    # utterance of a multi-utterance call into a call of its own, or the other way round.
    batch_coupled = True

    def __init__(self, param=0.5, method='kmeans', other_param='L2', max_iter=10, init='even', seed=0):
        if method != 'kmeans':
            raise NotImplementedError('Currently FEATURE COMPRESSION only supports kmeans on the native engine')
        if other_param != 'L2':
            raise NotImplementedError("only the 'L2' distance is built (the reference notes 'cos' works poorly, :178)")
        if init not in ('even', 'random'):
            raise ValueError("init must be 'even' or 'random'")
        self.param, self.max_iter, self.init, self.seed = param, max_iter, init, int(seed)
        self.calls = 0       # fwd calls so far: every call of the randomised defense draws fresh initial frames
        self.index_base = 0  # global index of row 0 (set by sharded callers)

    def call_seed(self, call):
        """Generator key of fwd call number `call` (0-based)."""
        from ..model._engine_ops import mix64
        return mix64(self.seed ^ 0x4665436F, call)

    # ---- forward with saved state ------------------------------------------------------------------
    def fwd(self, feat, seed=None, ids=None, row_keys=None):
        """feat (B,F,D) -> (compressed (B,k,D) [or (1,k',D) with empty clusters dropped when B == 1], saved).
        `seed`: explicit generator key for this call of the randomised defense (tests replay the fused loop's keys).
        `row_keys` = (index_base, row_base, rep_rows) as in sg_dither (speakerguard_hip.h): which global utterance and
        which EOT repeat every row of `feat` is -- repeat r draws from key + r * 0xC2B2AE3D27D4EB4F, like repeat r of the
        device loop (sg_an_pgd_run_feco); default: row b is utterance ``self.index_base + b``.
        `ids` (B,F) int32: cluster ids from elsewhere -- only the reference's step after the clustering runs
        (feature_level.py:204-216; tests/golden/feco_ref.npz pins it against the reference's own code)."""
        feat = feat.to(torch.float32).contiguous()
        if not feat.is_cuda:
            raise N.NativeError("FeCo runs on the HIP device only")
        B, F, D = feat.shape
        k = int(F * self.param)  # :184
        ctx, s = _context(feat.device), N.current_stream_ptr(feat.device)
        out = torch.empty(B, k, D, device=feat.device, dtype=torch.float32)
        counts = torch.empty(B, k, device=feat.device, dtype=torch.int32)
        if ids is not None:
            ids = ids.to(device=feat.device, dtype=torch.int32).contiguous()
            if tuple(ids.shape) != (B, F):
                raise ValueError("ids must have shape (B, F)")
            ctx.call("sg_feco_compress", N._ptr(feat), N._ptr(ids), B, F, D, k, N._ptr(out), N._ptr(counts), s)
        else:
            ids = torch.empty(B, F, device=feat.device, dtype=torch.int32)
            key = 0
            if self.init == 'random':
                key = self.call_seed(self.calls) if seed is None else int(seed) & 0xFFFFFFFFFFFFFFFF
            self.calls += 1
            # clustering + cluster means (:204-216) in one launch per EOT repeat the rows belong to (normally one)
            index_base, row_base, rep_rows = row_keys if row_keys is not None else (self.index_base, 0, 0)
            b0 = 0
            while b0 < B:
                g = row_base + b0
                rep = g // rep_rows if rep_rows > 0 else 0
                u = g - rep * rep_rows
                nb = min(B - b0, rep_rows - u) if rep_rows > 0 else B
                sl = slice(b0, b0 + nb)
                ctx.call("sg_feco_kmeans_compress", N._ptr(feat[sl]), nb, F, D, k, self.max_iter, int(self.init == 'random'),
                         C.c_uint64((key + rep * 0xC2B2AE3D27D4EB4F) & 0xFFFFFFFFFFFFFFFF), int(index_base + u), 1,
                         N._ptr(ids[sl]), N._ptr(out[sl]), N._ptr(counts[sl]), s)
                b0 += nb
        force = B > 1  # :33 force=feat.shape[0] > 1
        keep = None
        if not force and bool((counts == 0).any()):
            keep = torch.nonzero(counts[0] > 0).flatten()  # :209-212: empty clusters are skipped
            out = out.index_select(1, keep)
        return out, (ids, counts, (B, F, D, k), force, keep)

    def bwd(self, saved, dout):
        ids, counts, (B, F, D, k), force, keep = saved
        dout = dout.to(torch.float32)
        if keep is not None:
            full = torch.zeros(B, k, D, device=dout.device, dtype=torch.float32)
            full.index_copy_(1, keep, dout)
            dout = full
        dout = dout.contiguous()
        dfeat = torch.empty(B, F, D, device=dout.device, dtype=torch.float32)
        _context(dout.device).call("sg_feco_compress_backward", N._ptr(dout), N._ptr(ids), N._ptr(counts), B, F, D, k,
                                   1 if force else 0, N._ptr(dfeat), N.current_stream_ptr(dout.device))
        return dfeat

    def __call__(self, feat):
        return self.fwd(feat)[0]


def FeCo(feat, method='kmeans', param=0.5, other_param='L2'):
    return FEATURE_COMPRESSION(feat, method, param, other_param)


def FEATURE_COMPRESSION(feat, method='kmeans', param=0.5, other_param='L2'):
    return FeCoDefense(param=param, method=method, other_param=other_param)(feat)
