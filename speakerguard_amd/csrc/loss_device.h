// Decision + per-example attack loss + d loss / d scores, shared by the x-vector tail and the AudioNet
// head: serial code for ONE thread (loss_and_dscores) and a block-cooperative wrapper for the cross-entropy over many
// classes (loss_and_dscores_block).
//   decision: argmax, rejected (-1) unless max > threshold     model/iv_plda.py:182-194, audionet_csine.py:246-257
//   losses:   SEC4SR_CrossEntropy / SEC4SR_MarginLoss           attack/utils.py:7-102
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/speakerguard_hip.h"

namespace sg {

constexpr int kLossMaxS = 1024;

// argmax in index order (the first maximum wins), eight independent loads in flight per step: written as one load +
// compare per iteration, a single thread walking the 251 classes of the AudioNet head paid an LDS round trip per class
// (6 us; together with the equally serial sum below 17 of the 23 us of an_tail_kernel).  Same comparisons, same order.
__device__ __forceinline__ void argmax_in_order(const float* sc, int S, float& mx, int& ja) {
    mx = sc[0];
    ja = 0;
    int s = 1;
    for (; s + 8 <= S; s += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = sc[s + u];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (v[u] > mx) { mx = v[u]; ja = s + u; }
    }
    for (; s < S; ++s)
        if (sc[s] > mx) { mx = sc[s]; ja = s; }
}
// sum of ex[s], s != skip, in index order; the skipped term enters as + 0.f, which leaves a sum that started at + 0 unchanged
__device__ __forceinline__ float sum_in_order_except(const float* ex, int S, int skip) {
    float so = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ex[s + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) so += (s + u != skip) ? v[u] : 0.f;
    }
    for (; s < S; ++s) so += (s != skip) ? ex[s] : 0.f;
    return so;
}

// sc[S]: scores (LDS or registers); dsc[S]: must be zero on entry, receives d loss / d scores.
// coef: this utterance's row of ls.coef_dev (SG_LOSS_LINEAR), else unused.
// (mx, ja): the maximum score and the first index that holds it
__device__ __forceinline__ float loss_and_dscores_given(const float* sc, float* dsc, int S, float threshold, int64_t yy,
                                                        bool has_y, const sg_loss_spec& ls, int64_t* dec_out, const float* coef,
                                                        float mx, int ja) {
    *dec_out = mx > threshold ? (int64_t)ja : (int64_t)-1;
    float loss = 0.f;
    if (ls.loss == SG_LOSS_LINEAR) {  // vector-Jacobian product of the scores: the label plays no role
        if (coef)
            for (int s = 0; s < S; ++s) {
                loss += coef[s] * sc[s];
                dsc[s] = coef[s];
            }
        return loss;
    }
    if (has_y) {
        if (ls.loss == SG_LOSS_ENTROPY && ls.task == SG_TASK_CSI) {
            if (yy >= 0) {
                // softmax - onehot exactly as log_softmax + nll compute it in fp32 (the reference's
                // F.cross_entropy): d/ds_y = fl(exp(s_y - lse)) - 1.  The cancellation is part of
                // the reference's behaviour: for a confidently classified utterance (sum of the
                // other probabilities < 6e-8, i.e. a margin > 16.6 -- the normal case for PLDA
                // scores) exp() rounds to 1 and d/ds_y is EXACTLY 0, so the reference ascends along
                // sum_j p_j grad(s_j) only; an "exact" -(sum of other p_j) would follow a different
                // direction (measured: 71 % of the samples differ after 5 steps).  The only liberty
                // taken: the non-max terms are summed first and 1 is added last, so `se` carries a
                // single rounding instead of up to S-1.
                float so = 0.f;
                for (int s = 0; s < S; ++s)
                    if (s != ja) so += expf(sc[s] - mx);
                const float lse = logf(1.f + so);  // log-sum-exp RELATIVE to the max: torch's
                // log_softmax is (x - max) - log(sum exp(x - max)); adding the max back first would
                // round the 4e-5 of a confident utterance to ulp(max) ~ 8e-6 (a 9 % error on d/ds_y)
                loss = lse - (sc[yy] - mx);
                for (int s = 0; s < S; ++s) dsc[s] = expf((sc[s] - mx) - lse);
                dsc[yy] -= 1.f;
            }
        } else if (ls.task == SG_TASK_SV) {
            const bool enr = yy == 0;
            if (enr == (ls.targeted != 0)) { loss = ls.threshold + ls.confidence - sc[0]; dsc[0] = -1.f; }
            else { loss = sc[0] + ls.confidence - ls.threshold; dsc[0] = 1.f; }
        } else if (yy >= 0) {
            const float real = sc[yy];
            int jo = -1;
            float other = -10000.f;  // attack/utils.py:72
            for (int s = 0; s < S; ++s)
                if (s != yy && sc[s] > other) { other = sc[s]; jo = s; }
            if (ls.targeted) {
                if (ls.task == SG_TASK_CSI) {
                    loss = other + ls.confidence - real;
                    if (jo >= 0) dsc[jo] += 1.f;
                } else {
                    loss = fmaxf(other, ls.threshold) + ls.confidence - real;
                    if (jo >= 0 && other >= ls.threshold) dsc[jo] += 1.f;
                }
                dsc[yy] -= 1.f;
            } else if (ls.task == SG_TASK_CSI) {
                loss = real + ls.confidence - other;
                dsc[yy] += 1.f;
                if (jo >= 0) dsc[jo] -= 1.f;
            } else {
                const float f_rej = mx + ls.confidence - ls.threshold;
                const float f_mis = fmaxf(real, ls.threshold) + ls.confidence - other;
                loss = fminf(f_rej, f_mis);
                const float wr = f_rej < f_mis ? 1.f : (f_rej == f_mis ? 0.5f : 0.f);
                dsc[ja] += wr;
                if (real >= ls.threshold) dsc[yy] += 1.f - wr;
                if (jo >= 0) dsc[jo] -= 1.f - wr;
            }
        } else if (ls.task == SG_TASK_OSI) {
            if (ls.targeted) { loss = mx + ls.confidence - ls.threshold; dsc[ja] = 1.f; }
            else { loss = ls.threshold + ls.confidence - mx; dsc[ja] = -1.f; }
        }
        if (ls.loss == SG_LOSS_MARGIN || ls.task != SG_TASK_CSI) {
            if (ls.clip_max) {
                const float k = loss > 0.f ? 1.f : (loss == 0.f ? 0.5f : 0.f);
                for (int s = 0; s < S; ++s) dsc[s] *= k;
                loss = fmaxf(loss, 0.f);
            }
        }
    }
    return loss;
}

__device__ __forceinline__ float loss_and_dscores(const float* sc, float* dsc, int S, float threshold, int64_t yy,
                                                  bool has_y, const sg_loss_spec& ls, int64_t* dec_out,
                                                  const float* coef = nullptr) {
    int ja;
    float mx;
    argmax_in_order(sc, S, mx, ja);
    return loss_and_dscores_given(sc, dsc, S, threshold, yy, has_y, ls, dec_out, coef, mx, ja);
}

// One-wave form for S <= 64 (the x-vector tail: a handful of enrolled speakers), called by the 64 lanes of ONE wave; the
// returned loss and *dec_out are valid on lane 0.  Bit-identical to loss_and_dscores: the arg-max is the serial scan (on lane
// 0), the 2 S expf evaluations of the cross-entropy sit side by side on the lanes, the SUM of the other classes' terms stays
// in index order on lane 0.  Every other loss is the serial function on lane 0 (no transcendental in them).
// ex: >= 64 floats of LDS scratch.  The caller separates it from other users of sc / dsc / ex with block barriers.
__device__ __forceinline__ float loss_and_dscores_wave(const float* sc, float* dsc, float* ex, int S, float threshold, int64_t yy,
                                                       bool has_y, const sg_loss_spec& ls, int64_t* dec_out, int lane,
                                                       const float* coef = nullptr) {
    const bool ce = has_y && ls.loss == SG_LOSS_ENTROPY && ls.task == SG_TASK_CSI && yy >= 0;  // wave-uniform
    if (!ce) {
        float loss = 0.f;
        if (lane == 0) loss = loss_and_dscores(sc, dsc, S, threshold, yy, has_y, ls, dec_out, coef);
        return loss;
    }
    auto fence = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    float mx = 0.f;
    int ja = 0;
    if (lane == 0) argmax_in_order(sc, S, mx, ja);
    mx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, mx)));
    ja = __builtin_amdgcn_readfirstlane(ja);
    const float mine = lane < S ? sc[lane] : 0.f;
    if (lane < S) ex[lane] = expf(mine - mx);
    fence();
    float loss = 0.f, lse = 0.f;
    if (lane == 0) {
        *dec_out = mx > threshold ? (int64_t)ja : (int64_t)-1;
        const float so = sum_in_order_except(ex, S, ja);
        lse = logf(1.f + so);
        loss = lse - (sc[yy] - mx);
    }
    lse = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lse)));
    if (lane < S) dsc[lane] = expf((mine - mx) - lse) - (lane == (int)yy ? 1.f : 0.f);
    return loss;
}

// The same (maximum, first index holding it) by the whole block: per-thread scan in ascending index order, then merges
// that prefer the larger value and, among equal values, the lower index -- order-independent, so the result is the serial
// one whatever the reduction tree (scores are finite; -inf everywhere gives index 0 like the serial scan).  NaN scores: every
// comparison with a NaN is false, so a NaN never becomes the maximum, here or in the serial scan, but WHICH finite entry
// wins can then depend on where the NaNs sit relative to the merge tree -- a pass that produced NaN scores is garbage
// anyway (sg_xv_loss_grad and the device loops do not try to define it).  scratch: 2 floats per wave.  Ends with a barrier; valid on every thread.
__device__ __forceinline__ void argmax_block(const float* sc, int S, int tid, int nt, float* scratch, float& mx, int& ja) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if (tid < S) { bv = sc[tid]; bi = tid; }
    for (int s = tid + nt; s < S; s += nt) {
        const float v = sc[s];
        if (v > bv) { bv = v; bi = s; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    __syncthreads();  // (scratch may still be read by an earlier phase)
    if ((tid & 63) == 0) {
        scratch[2 * (tid >> 6)] = bv;
        scratch[2 * (tid >> 6) + 1] = __int_as_float(bi);
    }
    __syncthreads();
    bv = scratch[0];
    bi = __float_as_int(scratch[1]);
    for (int w = 1; w < nt / 64; ++w) {
        const float ov = scratch[2 * w];
        const int oi = __float_as_int(scratch[2 * w + 1]);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    mx = bv;
    ja = bi;
    __syncthreads();
}

// Block-cooperative form, called by ALL threads of the block (it synchronises); the returned loss and *dec_out are valid
// on thread 0.  Only the cross-entropy branch differs from the serial function: its 2 S expf evaluations -- 40 us on one
// thread for the 251 classes of the AudioNet head -- are spread over the block, while the SUM stays on thread 0 in index
// order (eight loads in flight per step) and the arg-max, which does not depend on the order, is a block reduction
// (round 4: the two serial scans were 17 of the 23 us of an_tail_kernel; now 5 of 11), so the result is bit-identical to
// loss_and_dscores.  ex: max(S, 32) floats of LDS scratch, bc: 4 floats.
__device__ __forceinline__ float loss_and_dscores_block(const float* sc, float* dsc, float* ex, float* bc, int S, float threshold,
                                                        int64_t yy, bool has_y, const sg_loss_spec& ls, int64_t* dec_out, int tid,
                                                        int nt, const float* coef = nullptr) {
    const bool ce = has_y && ls.loss == SG_LOSS_ENTROPY && ls.task == SG_TASK_CSI && yy >= 0;  // block-uniform
    if (S <= 32) {
        // a handful of enrolled speakers (the x-vector tail: S = 10): the serial function on one thread is ~1 us; the
        // cooperative form below costs seven block barriers (3.9 us measured in tail_kernel with 1024 threads)
        float loss = 0.f;
        if (tid == 0) loss = loss_and_dscores(sc, dsc, S, threshold, yy, has_y, ls, dec_out, coef);
        __syncthreads();
        return loss;
    }
    float mx;
    int ja;
    argmax_block(sc, S, tid, nt, ex, mx, ja);  // (one thread scanning the 251 AudioNet classes took 6 us)
    if (!ce) {
        float loss = 0.f;
        if (tid == 0) loss = loss_and_dscores_given(sc, dsc, S, threshold, yy, has_y, ls, dec_out, coef, mx, ja);
        __syncthreads();
        return loss;
    }
    if (tid == 0) *dec_out = mx > threshold ? (int64_t)ja : (int64_t)-1;
    for (int s = tid; s < S; s += nt) ex[s] = expf(sc[s] - mx);
    __syncthreads();
    float loss = 0.f;
    if (tid == 0) {
        const float so = sum_in_order_except(ex, S, ja);
        const float lse = logf(1.f + so);
        loss = lse - (sc[yy] - mx);
        bc[2] = lse;
    }
    __syncthreads();
    const float lse = bc[2];
    for (int s = tid; s < S; s += nt) dsc[s] = expf((sc[s] - mx) - lse);
    __syncthreads();
    if (tid == 0) dsc[yy] -= 1.f;
    __syncthreads();
    return loss;
}

}  // namespace sg
