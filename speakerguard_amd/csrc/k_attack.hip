// Elementwise attack-state kernels around the model call (all HBM-bound, one pass each):
//   cw2_step      attack/CW2.py:72-82   tanh box, L2 term, chain through tanh, torch.optim.Adam update
//   nes_queries   adaptive_attack/NES.py:19-25  antithetic Gaussian queries around x
//   nes_grad      adaptive_attack/NES.py:47-54  loss-weighted noise average (noise regenerated, never stored)
//   fakebob_step  attack/FAKEBOB.py:93-104      momentum, per-example LR sign step, epsilon-ball clamp
#include "loss_device.h"
#include "sg_internal.h"

// These updates mirror torch elementwise expressions (separate multiply and add roundings); keep hipcc
// from contracting them into FMAs so the results are bit-identical to the reference formulas.
#pragma clang fp contract(off)

namespace sg {

__device__ __forceinline__ float wave_sum_a(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------- CW2
// One pass: (optional) Adam update of the modifier from d loss1/d input_x of the CURRENT input, then
// the NEXT input_x = tanh(modifier + atanh(0.999999 x)) and per-block partials of
// loss2 = sum (input_x - x)^2.  grid (nblk, B), block 256; part[b][blockIdx.x].
__global__ __launch_bounds__(256) void cw2_step_kernel(float* __restrict__ modifier, float* __restrict__ exp_avg,
                                                       float* __restrict__ exp_avg_sq, const float* __restrict__ x,
                                                       const float* __restrict__ input_cur,
                                                       const float* __restrict__ grad1,
                                                       const float* __restrict__ const_c, int T, float step_size,
                                                       float bc2_sqrt, float one_m_b1, float beta2, float one_m_b2, float eps,
                                                       float* __restrict__ input_next, float* __restrict__ part) {
    const int b = blockIdx.y;
    const size_t base = (size_t)b * T;
    const float c = grad1 ? const_c[b] : 0.f;
    float acc = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < T; i += gridDim.x * 256) {
        const size_t o = base + i;
        const float xv = x[o];
        float m = modifier[o];
        if (grad1) {
            const float in = input_cur[o];
            // d/d modifier of const*loss1 + ||input - x||^2 through input = tanh(.)
            const float g = (c * grad1[o] + 2.f * (in - xv)) * (1.f - in * in);
            // torch.optim.Adam (single-tensor path): exp_avg.lerp_(g, 1-b1); exp_avg_sq.mul_(b2)
            // .addcmul_(g, g, value=1-b2); denom = sqrt(v)/sqrt(bc2) + eps; p.addcdiv_(m, denom, -lr/bc1)
            float ea = exp_avg[o], es = exp_avg_sq[o];
            ea = ea + one_m_b1 * (g - ea);
            es = es * beta2 + (one_m_b2 * g) * g;
            const float denom = sqrtf(es) / bc2_sqrt + eps;
            m = m + (-step_size * ea) / denom;
            exp_avg[o] = ea;
            exp_avg_sq[o] = es;
            modifier[o] = m;
        }
        const float nx = tanhf(m + atanhf(xv * 0.999999f));
        input_next[o] = nx;
        const float d = nx - xv;
        acc += d * d;
    }
    __shared__ float red[4];
    acc = wave_sum_a(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)b * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(64) void row_sum_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    const int b = blockIdx.x;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) acc += part[(size_t)b * n + i];
    acc = wave_sum_a(acc);
    if (threadIdx.x == 0) out[b] = acc;
}

hipError_t launch_cw2_step(float* modifier, float* exp_avg, float* exp_avg_sq, const float* x, const float* input_cur,
                           const float* grad1, const float* const_c, int B, int T, float lr, int step_t,
                           float* input_next, float* loss2, float* scratch, hipStream_t s) {
    const double b1 = 0.9, b2 = 0.999;  // torch.optim.Adam defaults (CW2.py:57); scalars rounded like Python does
    const float beta2 = (float)b2, eps = 1e-8f;
    const double bc1 = 1.0 - pow(b1, step_t), bc2 = 1.0 - pow(b2, step_t);
    const int nblk = 32;
    hipLaunchKernelGGL(cw2_step_kernel, dim3(nblk, B), dim3(256), 0, s, modifier, exp_avg, exp_avg_sq, x, input_cur, grad1,
                       const_c, T, (float)(lr / bc1), (float)sqrt(bc2), (float)(1.0 - b1), beta2, (float)(1.0 - b2), eps, input_next,
                       scratch);
    hipLaunchKernelGGL(row_sum_kernel, dim3(B), dim3(64), 0, s, scratch, nblk, loss2);
    return hipGetLastError();
}

// ---------------------------------------------------------------- NES / FAKEBOB
__device__ __forceinline__ uint32_t philox_a(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                             uint32_t* second) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    *second = c1;
    return c0;
}

// standard normal for (example index, antithetic pair index, sample t): Box-Muller on two Philox words
__device__ __forceinline__ float nes_normal(uint64_t seed, int64_t example, int pair, int t) {
    uint32_t r1;
    const uint32_t r0 = philox_a(seed, (uint32_t)t, (uint32_t)pair, (uint32_t)example, (uint32_t)((uint64_t)example >> 32), &r1);
    const float u0 = ((float)(r0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u1 = ((float)(r1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.f * logf(u0)) * cosf(6.283185307179586f * u1);
}

// queries (n, Q, T): [clean?] then S/2 "+" then S/2 "-" (NES.py:19-23); noise_out optional (n, S/2, T)
__global__ __launch_bounds__(256) void nes_queries_kernel(const float* __restrict__ x, int T, int half, int with_clean,
                                                          float sigma, uint64_t seed, int64_t index_base, int pair_base,
                                                          const float* __restrict__ noise_in,
                                                          float* __restrict__ queries, float* __restrict__ noise_out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int e = blockIdx.y;
    if (t >= T) return;
    const int Q = 2 * half + with_clean;
    const float xv = x[(size_t)e * T + t];
    float* q = queries + (size_t)e * Q * T + t;
    if (with_clean) q[0] = xv;
    for (int p = 0; p < half; ++p) {
        const float z = noise_in ? noise_in[((size_t)e * half + p) * T + t] : nes_normal(seed, index_base + e, pair_base + p, t);
        if (noise_out) noise_out[((size_t)e * half + p) * T + t] = z;
        q[(size_t)(with_clean + p) * T] = z * sigma + xv;
        q[(size_t)(with_clean + half + p) * T] = -z * sigma + xv;
    }
}

// grad[e][t] (+)= (1/S) sum_p (loss[e][p+] - loss[e][p-]) * noise[e][p][t]; accumulate != 0 adds to grad
__global__ __launch_bounds__(256) void nes_grad_kernel(const float* __restrict__ loss, int T, int half, int with_clean,
                                                       uint64_t seed, int64_t index_base, int pair_base,
                                                       const float* __restrict__ noise_in, int accumulate,
                                                       float final_sigma, float final_batches,
                                                       float* __restrict__ grad) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int e = blockIdx.y;
    if (t >= T) return;
    const int Q = 2 * half + with_clean;
    const float* l = loss + (size_t)e * Q + with_clean;
    float acc = 0.f;
    // mean over the 2*half queries of loss * noise, in the reference's order: all "+" terms, then all "-"
    for (int p = 0; p < half; ++p) {
        const float z = noise_in ? noise_in[((size_t)e * half + p) * T + t] : nes_normal(seed, index_base + e, pair_base + p, t);
        acc += l[p] * z;
    }
    for (int p = 0; p < half; ++p) {
        const float z = noise_in ? noise_in[((size_t)e * half + p) * T + t] : nes_normal(seed, index_base + e, pair_base + p, t);
        acc += l[half + p] * -z;
    }
    acc /= (float)(2 * half);
    const size_t o = (size_t)e * T + t;
    float g = accumulate ? grad[o] + acc : acc;
    if (final_sigma > 0.f) g = (g / final_sigma) / final_batches;  // NES.py:54 divides twice
    grad[o] = g;
}

// grad = momentum*prev + (1-momentum)*grad (in place); x = clamp(x + grad_sign*lr[e]*sign(grad), lower, upper)
__global__ __launch_bounds__(256) void fakebob_step_kernel(float* __restrict__ x, float* __restrict__ grad,
                                                           const float* __restrict__ prev_grad,
                                                           const float* __restrict__ lr, const float* __restrict__ lower,
                                                           const float* __restrict__ upper, int T, float momentum,
                                                           float one_m_momentum, int grad_sign) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int e = blockIdx.y;
    if (t >= T) return;
    const size_t o = (size_t)e * T + t;
    // FAKEBOB.py:93: the two weights are Python doubles rounded to fp32 by the caller
    const float g = momentum * prev_grad[o] + one_m_momentum * grad[o];
    grad[o] = g;
    const float sg = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
    const float v = x[o] + (float)grad_sign * lr[e] * sg;
    x[o] = fminf(fmaxf(v, lower[o]), upper[o]);
}

hipError_t launch_nes_queries(const float* x, int n, int T, int half, int with_clean, float sigma, uint64_t seed,
                              int64_t index_base, int pair_base, const float* noise_in, float* queries, float* noise_out,
                              hipStream_t s) {
    hipLaunchKernelGGL(nes_queries_kernel, dim3((T + 255) / 256, n), dim3(256), 0, s, x, T, half, with_clean, sigma, seed,
                       index_base, pair_base, noise_in, queries, noise_out);
    return hipGetLastError();
}

hipError_t launch_nes_grad(const float* loss, int n, int T, int half, int with_clean, uint64_t seed, int64_t index_base,
                           int pair_base, const float* noise_in, int accumulate, float final_sigma, int final_batches,
                           float* grad, hipStream_t s) {
    hipLaunchKernelGGL(nes_grad_kernel, dim3((T + 255) / 256, n), dim3(256), 0, s, loss, T, half, with_clean, seed,
                       index_base, pair_base, noise_in, accumulate, final_sigma, (float)final_batches, grad);
    return hipGetLastError();
}

hipError_t launch_fakebob_step(float* x, float* grad, const float* prev_grad, const float* lr, const float* lower,
                               const float* upper, int n, int T, float momentum, float one_m_momentum, int grad_sign,
                               hipStream_t s) {
    hipLaunchKernelGGL(fakebob_step_kernel, dim3((T + 255) / 256, n), dim3(256), 0, s, x, grad, prev_grad, lr, lower, upper,
                       T, momentum, one_m_momentum, grad_sign);
    return hipGetLastError();
}

// ---------------------------------------------------------------- loss stage alone
// attack/utils.py:7-102 on given scores: one thread per utterance (S is 10 .. 251), global memory as its scratch.
__global__ __launch_bounds__(64) void loss_eval_kernel(const float* __restrict__ scores, const int64_t* __restrict__ y, int B,
                                                       int S, float threshold, sg_loss_spec ls, int64_t* __restrict__ dec_out,
                                                       float* __restrict__ loss_out, float* __restrict__ dsc_out) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    float* dsc = dsc_out + (size_t)b * S;
    for (int s = 0; s < S; ++s) dsc[s] = 0.f;
    int64_t dec = 0;
    const float loss = loss_and_dscores(scores + (size_t)b * S, dsc, S, threshold, y ? y[b] : 0, y != nullptr, ls, &dec,
                                        ls.coef_dev ? ls.coef_dev + (size_t)b * S : nullptr);
    if (dec_out) dec_out[b] = dec;
    if (loss_out) loss_out[b] = loss;
}

hipError_t launch_loss_eval(const float* scores, const int64_t* y, int B, int S, float threshold, const sg_loss_spec& ls,
                            int64_t* dec, float* loss, float* dscores, hipStream_t s) {
    if (S < 1 || S > kLossMaxS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(loss_eval_kernel, dim3((B + 63) / 64), dim3(64), 0, s, scores, y, B, S, threshold, ls, dec, loss, dscores);
    return hipGetLastError();
}

}  // namespace sg
