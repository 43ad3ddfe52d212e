"""Native attack-state updates shared by every engine-backed model (C-ABI: sg_pgd_update, sg_cw2_step,
sg_nes_queries, sg_nes_grad, sg_fakebob_step).  The attack classes call these through the model so
that the only implementation in the product is the HIP one (tests substitute a CPU double)."""
import ctypes as C

import torch

from .. import _native as N


_MASK64 = (1 << 64) - 1


def mix64(*vals):
    """Order-dependent 64-bit hash of integers (splitmix-style); seeds of the engine's counter-based generators."""
    h = 0x9E3779B97F4A7C15
    for v in vals:
        h = ((h ^ (int(v) & _MASK64)) * 0xBF58476D1CE4E5B9) & _MASK64
        h ^= h >> 31
    return h


class EngineOps:
    """Mixin: expects ``self.ctx`` (N.Context) and ``self.device``.

    Noise bookkeeping (dither of the MFCC front-end, NES queries, FeCo's random start): the reference draws from the
    process-global torch RNG, which makes an utterance's noise depend on everything that ran before it.  Here every
    draw is keyed by (user seed, attack call, restart, pass number inside the chunk) and, inside the kernels, by the
    GLOBAL index of the utterance (chunk base + row; an EOT repeat moves the key, not the index) -- so the noise an
    utterance sees does not depend on how the batch is chunked or cut into per-GPU shards, wherever the cut falls
    (round 4; rounds 2-3 hashed the chunk base into the key, which tied the invariance to chunk-aligned cuts).
    ``attack()`` calls ``begin_attack`` once, ``_run_batches`` calls ``begin_batch`` per chunk; without them every
    forward simply advances ``_draw`` (fresh noise per call, like the reference).

    What a kernel needs to find "global utterance, repeat" from a row of the call (sg_dither, speakerguard_hip.h):
    ``row_keys()`` = (index base of row 0, offset of this call's row 0 inside the full call, rows per EOT repeat).
    """
    _noise_epoch = 0   # attack() calls so far
    _batch_salt = 0    # restart number + 1 (0: none)
    _index_base = 0    # global index of the first utterance of the chunk being attacked
    _draw = 0          # passes since begin_batch
    _nes_draw = 0      # NES.forward calls since begin_batch
    _def_draw = 0      # calls of a randomised defense (FeCoDefense(init='random')) since begin_batch
    _row_base = 0      # position of this call's row 0 inside the full model call it is a slice of (shard.QueryShardedModel)
    _row_scale = 1     # rows every utterance of the chunk contributes to the call (NES: its queries), set by the caller
    _rep_rows = 0      # > 0: the full call's rows are EOT repeats of _rep_rows rows (adaptive_attack/EOT.py sets it)

    def begin_attack(self):
        self._noise_epoch += 1

    def begin_batch(self, index_base=0, salt=0):
        self._index_base, self._batch_salt, self._draw, self._nes_draw, self._def_draw = int(index_base), int(salt), 0, 0, 0

    def noise_seed(self, user_seed, draw):
        return mix64(user_seed, self._noise_epoch, self._batch_salt, draw)

    def row_keys(self):
        """(index_base, row_base, rep_rows) of the model call being made, see sg_dither."""
        return int(self._index_base) * int(self._row_scale), int(self._row_base), int(self._rep_rows)

    def defense_seed(self, user_seed):
        """Generator key of the next call of a randomised defense sitting on this model (defended_model): keyed like the
        dither -- (seed, attack call, restart, call number inside the chunk) + global utterance index -- so that the clusterings an
        utterance sees do not depend on the shard layout either (rows: ``row_keys()``)."""
        key = self.noise_seed(int(user_seed) ^ 0x4665436F, self._def_draw)
        self._def_draw += 1
        return key

    def _stream(self):
        return N.current_stream_ptr(self.device)

    def check_health(self):
        """Raise NativeError if a kernel of an earlier launch flagged its own result as invalid (sg_health: a
        stream-K hand-off wait that timed out).  No synchronisation: call it once the results were awaited."""
        self.ctx.call("sg_health")

    def set_streamk(self, enable):
        """sg_set_streamk: False = every contraction as one block per tile (same bits; no need for all blocks of a launch to
        be resident at once -- the fallback on a GPU this process does not have to itself)."""
        self.ctx.call("sg_set_streamk", int(bool(enable)))
        self.streamk = bool(enable)

    def trace_stages(self, fn, max_records=4096):
        """Run fn() with the library's stage trace on (sg_trace_begin / sg_trace_end: a HIP-event pair around every launch
        of the pass sequences, on the launch stream) and return [(stage name, milliseconds)] in launch order.  Measurement
        aid (bench.py `roofline`); the events cost a few microseconds per launch, so trace a run of its own."""
        self.ctx.call("sg_trace_begin", int(max_records))
        try:
            fn()
        finally:
            tags = (C.c_int32 * max_records)()
            ms = (C.c_float * max_records)()
            n = C.c_int32()
            self.ctx.call("sg_trace_end", tags, ms, int(max_records), C.byref(n))
        k = min(n.value, max_records)
        return [(N.STAGE_NAMES.get(tags[i], str(tags[i])), float(ms[i])) for i in range(k)]

    def pgd_update(self, x, grad, lower, upper, step_size, grad_sign):
        """x <- min(max(x + step*sign(grad)*grad_sign, lower), upper) in place (attack/FGSM.py:65,68)."""
        self.ctx.call("sg_pgd_update", N._ptr(x), N._ptr(grad), N._ptr(lower), N._ptr(upper), x.numel(),
                      float(step_size), int(grad_sign), self._stream())
        return x

    def cw2_step(self, modifier, exp_avg, exp_avg_sq, x, input_cur, grad1, const, lr, step_t):
        """attack/CW2.py:72-82.  Updates modifier/Adam state in place when grad1 is given; returns the
        next (input_x, loss2)."""
        B, _, T = x.shape
        input_next = torch.empty_like(x)
        loss2 = torch.empty(B, device=x.device, dtype=torch.float32)
        self.ctx.call("sg_cw2_step", N._ptr(modifier), N._ptr(exp_avg), N._ptr(exp_avg_sq), N._ptr(x), N._ptr(input_cur),
                      N._ptr(grad1), N._ptr(const), B, T, float(lr), int(step_t), N._ptr(input_next), N._ptr(loss2),
                      self._stream())
        return input_next, loss2

    def nes_queries(self, x, half, with_clean, sigma, seed, pair_base, noise_in=None, want_noise=False, index_base=0):
        """adaptive_attack/NES.py:19-25 -> queries (n*(2*half+with_clean), 1, T) [, noise (n, half, 1, T)]."""
        n, _, T = x.shape
        Q = 2 * half + int(with_clean)
        queries = torch.empty(n * Q, 1, T, device=x.device, dtype=torch.float32)
        noise = torch.empty(n, half, 1, T, device=x.device, dtype=torch.float32) if want_noise else None
        self.ctx.call("sg_nes_queries", N._ptr(x), n, T, half, int(with_clean), float(sigma), C.c_uint64(seed), int(index_base),
                      int(pair_base), N._ptr(noise_in), N._ptr(queries), N._ptr(noise), self._stream())
        return queries, noise

    def nes_grad(self, loss, grad, n, T, half, with_clean, seed, pair_base, noise_in, accumulate, final_sigma, final_batches,
                 index_base=0):
        """adaptive_attack/NES.py:47-54; accumulates into `grad` (n,1,T)."""
        self.ctx.call("sg_nes_grad", N._ptr(loss), n, T, half, int(with_clean), C.c_uint64(seed), int(index_base), int(pair_base),
                      N._ptr(noise_in), int(accumulate), float(final_sigma), int(final_batches), N._ptr(grad), self._stream())
        return grad

    def fakebob_step(self, x, grad, prev_grad, lr, lower, upper, momentum, grad_sign):
        """attack/FAKEBOB.py:93-104; grad and x are updated in place."""
        n, _, T = x.shape
        self.ctx.call("sg_fakebob_step", N._ptr(x), N._ptr(grad), N._ptr(prev_grad), N._ptr(lr), N._ptr(lower), N._ptr(upper),
                      n, T, float(momentum), float(1.0 - momentum), int(grad_sign), self._stream())
        return x, grad
