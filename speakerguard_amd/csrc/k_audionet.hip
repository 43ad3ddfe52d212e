// AudioNet CSI-NE kernels other than the 1-D convolutions (those run on conv_gemm_kernel):
//   log-mel front-end and its hand-coded backward   reference model/_audionet/Preprocessor.py:85-112
//   5x5 pre-filter (Conv2d + BatchNorm2d, folded)   model/audionet_csine.py:66-71,181-185
//   MaxPool1d(2) forward / un-pool + ReLU mask      :77,94,111
//   max over time, Linear, decision, loss, backward :205-211,246-257 + attack/utils.py losses
// Activations are channel-last (B, T, C) like everywhere else in the library.
#include "loss_device.h"
#include "sg_internal.h"

namespace sg {

constexpr int kAnWavesPerBlock = 2;

__device__ __forceinline__ void an_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float an_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct AnFrameLds {
    double2 spec[kAnFft];   // FFT work buffer (fp64: same dynamic-range argument as the MFCC front-end)
    float mel[32];
    float dmel[34];
};

// radix-2 DIT (bit-reversed input, natural output), 1024 points, one wave, private LDS buffer
__device__ __forceinline__ void an_fft_dit(double2* buf, const double2* __restrict__ tw, int lane) {
#pragma unroll 1
    for (int s = 0; s < 10; ++s) {
        const int half = 1 << s;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = lane + 64 * i;
            const int pos = j & (half - 1);
            const int i0 = ((j >> s) << (s + 1)) + pos;
            const int i1 = i0 + half;
            const double2 w = tw[pos << (9 - s)];
            const double2 a = buf[i0], b = buf[i1];
            const double tr = b.x * w.x - b.y * w.y;
            const double ti = b.x * w.y + b.y * w.x;
            buf[i0] = make_double2(a.x + tr, a.y + ti);
            buf[i1] = make_double2(a.x - tr, a.y - ti);
        }
        an_wave_sync();
    }
}
// inverse, DIF: natural input, element n of the (unnormalised) result lands at buf[bitrev(n)]
__device__ __forceinline__ void an_ifft_dif(double2* buf, const double2* __restrict__ tw, int lane) {
#pragma unroll 1
    for (int s = 9; s >= 0; --s) {
        const int half = 1 << s;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = lane + 64 * i;
            const int pos = j & (half - 1);
            const int i0 = ((j >> s) << (s + 1)) + pos;
            const int i1 = i0 + half;
            const double2 w = tw[pos << (9 - s)];
            const double2 a = buf[i0], b = buf[i1];
            const double dx = a.x - b.x, dy = a.y - b.y;
            buf[i0] = make_double2(a.x + b.x, a.y + b.y);
            buf[i1] = make_double2(dx * w.x + dy * w.y, dy * w.x - dx * w.y);
        }
        an_wave_sync();
    }
}

// pre-emphasised sample p of utterance row xr (length T), p in [0, T-2]: x[p+1] - 0.97 x[p]
__device__ __forceinline__ float an_preemph(const float* __restrict__ xr, int p, float scale) {
    return (xr[p + 1] - 0.97f * xr[p]) * scale;
}
__device__ __forceinline__ int an_reflect(int p, int L) { return p < 0 ? -p : (p >= L ? 2 * (L - 1) - p : p); }

// spectrum of frame f into L.spec, mel energies into L.mel
__device__ __forceinline__ void an_frame_forward(const AnTables& t, AnFrameLds& L, const float* __restrict__ xr, int T,
                                                 int f, float scale, int lane) {
    const int Lp = T - 1;                         // length of the pre-emphasised signal
    const int base = f * kAnHop - kAnWin / 2;     // centre=True: frame f is centred on sample f*hop
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = lane + 64 * i;              // FFT input index; the 800-tap window sits at 112..911
        const int n = q - (kAnFft - kAnWin) / 2;
        double v = 0.0;
        if (n >= 0 && n < kAnWin) v = (double)(an_preemph(xr, an_reflect(base + n, Lp), scale) * t.window[n]);
        L.spec[t.bitrev[q]] = make_double2(v, 0.0);
    }
    an_wave_sync();
    an_fft_dit(L.spec, t.twiddle, lane);
    // 32 slaney-mel filters, two lanes per filter (each takes half of the filter's bin range)
    {
        const int m = lane >> 1, h = lane & 1;
        const int lo = t.mel_lo[m], hi = t.mel_hi[m];
        const int mid = lo + (hi - lo + 1) / 2;
        float acc = 0.f;
        for (int k = h ? mid : lo; k < (h ? hi : mid); ++k) {
            const double2 c = L.spec[k];
            acc += (float)(c.x * c.x + c.y * c.y) * t.mel_w[m * kAnBins + k];
        }
        acc += __shfl_xor(acc, 1, 64);
        if (h == 0) L.mel[m] = acc;
    }
    an_wave_sync();
}

__global__ __launch_bounds__(128) void an_logmel_fwd_kernel(AnTables t, const float* __restrict__ x, int T, int F,
                                                            const float* __restrict__ scale_p, float* __restrict__ feats) {
    __shared__ AnFrameLds lds[kAnWavesPerBlock];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int f = blockIdx.x * kAnWavesPerBlock + wid;
    if (f >= F) return;
    const float scale = scale_p ? *scale_p : 1.f;
    AnFrameLds& L = lds[wid];
    an_frame_forward(t, L, x + (size_t)b * T, T, f, scale, lane);
    if (lane < kAnMel) feats[((size_t)b * F + f) * kAnMel + lane] = 10.f * log10f(fmaxf(L.mel[lane], 1e-16f));
}

// dfeats (B,F,32) -> dframes (B,F,800): gradient wrt the pre-emphasised, reflect-padded frame samples
__global__ __launch_bounds__(128) void an_logmel_bwd_kernel(AnTables t, const float* __restrict__ x, int T, int F,
                                                            const float* __restrict__ scale_p,
                                                            const float* __restrict__ dfeats, float* __restrict__ dframes) {
    __shared__ AnFrameLds lds[kAnWavesPerBlock];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int f = blockIdx.x * kAnWavesPerBlock + wid;
    if (f >= F) return;
    const float scale = scale_p ? *scale_p : 1.f;
    AnFrameLds& L = lds[wid];
    an_frame_forward(t, L, x + (size_t)b * T, T, f, scale, lane);
    if (lane < 34) {
        float dm = 0.f;
        if (lane < kAnMel) {
            const float mel = L.mel[lane];
            // d/d mel of 10 log10(max(mel, 1e-16))
            dm = mel > 1e-16f ? dfeats[((size_t)b * F + f) * kAnMel + lane] * (10.f / 2.302585092994046f) / mel : 0.f;
        }
        L.dmel[lane] = dm;
    }
    an_wave_sync();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = lane + 64 * i;
        double2 g = make_double2(0.0, 0.0);
        if (k < kAnBins) {
            const int m0 = t.bin_m0[k];
            if (m0 >= 0) {
                const double dp = 2.0 * (double)(L.dmel[m0] * t.bin_w0[k] + L.dmel[m0 + 1] * t.bin_w1[k]);
                const double2 c = L.spec[k];
                g = make_double2(c.x * dp, c.y * dp);
            }
        }
        L.spec[k] = g;
    }
    an_wave_sync();
    an_ifft_dif(L.spec, t.twiddle, lane);
    float* out = dframes + ((size_t)b * F + f) * kAnWin;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const int n = lane + 64 * i;
        if (n < kAnWin) out[n] = (float)L.spec[t.bitrev[n + (kAnFft - kAnWin) / 2]].x * t.window[n];
    }
}

// Deterministic overlap-add + pre-emphasis backward (+ optional fused PGD update).
// d pre[p] gathers every frame position that reads pre-emphasised sample p (directly or through
// the reflect padding of torch.stft); d x[t] = d pre[t-1] - 0.97 d pre[t].
__global__ __launch_bounds__(256) void an_frames_to_wave_kernel(const float* __restrict__ dframes, int T, int F,
                                                                const float* __restrict__ scale_p,
                                                                float* __restrict__ grad_out, float* __restrict__ x_io,
                                                                const float* __restrict__ lower,
                                                                const float* __restrict__ upper, float step, int grad_sign) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= T) return;
    const float scale = scale_p ? *scale_p : 1.f;
    const float* df = dframes + (size_t)b * F * kAnWin;
    const int Lp = T - 1;
    const int pmin = -kAnWin / 2, pmax = (F - 1) * kAnHop + kAnWin / 2 - 1;
    auto at_pos = [&](int p) {  // sum over the frames that cover signal position p
        float g = 0.f;
        if (p < pmin || p > pmax) return g;
        const int q = p + kAnWin / 2;  // >= 0
        int fhi = q / kAnHop;
        const int flo = q > kAnWin - 1 ? (q - (kAnWin - 1) + kAnHop - 1) / kAnHop : 0;
        if (fhi > F - 1) fhi = F - 1;
        for (int f = flo; f <= fhi; ++f) g += df[(size_t)f * kAnWin + (q - f * kAnHop)];
        return g;
    };
    auto dpre = [&](int s) {  // gradient wrt pre-emphasised sample s in [0, Lp)
        float g = at_pos(s);
        if (s >= 1) g += at_pos(-s);                    // left reflection  p = -s
        if (s <= Lp - 2) g += at_pos(2 * (Lp - 1) - s);  // right reflection p = 2(L-1) - s
        return g;
    };
    float g = 0.f;
    if (t >= 1) g += dpre(t - 1);
    if (t <= Lp - 1) g -= 0.97f * dpre(t);
    g *= scale;
    const size_t o = (size_t)b * T + t;
    if (grad_out) grad_out[o] = g;
    if (x_io) {
        const float sg = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
        x_io[o] = fminf(fmaxf(x_io[o] + step * sg * (float)grad_sign, lower[o]), upper[o]);
    }
}

// ---------------------------------------------------------------- 5x5 pre-filter over (time, mel)
// out[t][m] = b + sum_{i,j} w[i][j] in[t + j - 2][m + i - 2]   (Conv2d on the (mel, time) image, zero pad)
// transpose != 0 computes the data gradient (correlation with the flipped kernel, no bias).
__global__ __launch_bounds__(256) void an_prefilter_kernel(const float* __restrict__ in, float* __restrict__ out, int T,
                                                           const float* __restrict__ w25, float bias, int transpose) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (idx >= T * kAnMel) return;
    const int t = idx / kAnMel, m = idx - t * kAnMel;
    const float* x = in + (size_t)b * T * kAnMel;
    float acc = transpose ? 0.f : bias;
#pragma unroll
    for (int i = 0; i < 5; ++i) {      // mel offset
#pragma unroll
        for (int j = 0; j < 5; ++j) {  // time offset
            const int mm = transpose ? m - i + 2 : m + i - 2;
            const int tt = transpose ? t - j + 2 : t + j - 2;
            if (mm >= 0 && mm < kAnMel && tt >= 0 && tt < T) acc += w25[i * 5 + j] * x[(size_t)tt * kAnMel + mm];
        }
    }
    out[(size_t)b * T * kAnMel + idx] = acc;
}

// ---------------------------------------------------------------- MaxPool1d(2, stride 2) along time
__global__ __launch_bounds__(256) void an_pool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int Tin,
                                                          int C) {
    const int Tout = Tin / 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (idx >= Tout * C) return;
    const int t = idx / C, c = idx - t * C;
    const float* x = in + ((size_t)b * Tin + 2 * t) * C + c;
    out[(size_t)b * Tout * C + idx] = fmaxf(x[0], x[C]);
}
// d pre-activation of the pooled layer: route the pooled gradient to the (first) arg-max of each pair
// and apply the ReLU mask of the un-pooled activation `act`
__global__ __launch_bounds__(256) void an_pool_bwd_kernel(const float* __restrict__ act, const float* __restrict__ dpool,
                                                          float* __restrict__ dact, int Tin, int C) {
    const int Tout = Tin / 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (idx >= Tin * C) return;
    const int t = idx / C, c = idx - t * C;
    float g = 0.f;
    if (t < 2 * Tout) {
        const int tp = t >> 1;
        const float* a = act + ((size_t)b * Tin + 2 * tp) * C + c;
        const float a0 = a[0], a1 = a[C];
        const bool first = !(a1 > a0);  // torch keeps the first maximum on ties
        const float mine = (t & 1) ? a1 : a0;
        if (((t & 1) == 0) == first && mine > 0.f) g = dpool[((size_t)b * Tout + tp) * C + c];
    }
    dact[(size_t)b * Tin * C + idx] = g;
}

// ---------------------------------------------------------------- head: max over time, fc, decision, loss, backward
// one block (256 threads) per utterance; act8 (B, T8, 32) ReLU outputs of conv8
__global__ __launch_bounds__(256) void an_tail_kernel(const float* __restrict__ act8, int T8, const float* __restrict__ fc_w,
                                                      const float* __restrict__ fc_b, int S, float threshold,
                                                      const int64_t* __restrict__ y, sg_loss_spec ls, int want_grad,
                                                      float* __restrict__ emb_out, float* __restrict__ scores_out,
                                                      int64_t* __restrict__ dec_out, float* __restrict__ loss_out,
                                                      float* __restrict__ dact8, float* __restrict__ loss_trace,
                                                      int64_t* __restrict__ dec_trace, uint8_t* __restrict__ success) {
    __shared__ float emb[32];
    __shared__ int arg[32];
    __shared__ float sc[kLossMaxS], dsc[kLossMaxS];
    __shared__ float demb[32];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* a = act8 + (size_t)b * T8 * 32;
    if (tid < 32) {  // x.max(2): first maximum
        float mx = a[tid];
        int at = 0;
        for (int t = 1; t < T8; ++t) {
            const float v = a[(size_t)t * 32 + tid];
            if (v > mx) { mx = v; at = t; }
        }
        emb[tid] = mx;
        arg[tid] = at;
        if (emb_out) emb_out[(size_t)b * 32 + tid] = mx;
    }
    __syncthreads();
    for (int s = tid; s < S; s += 256) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) acc += fc_w[(size_t)s * 32 + c] * emb[c];
        acc += fc_b[s];
        sc[s] = acc;
        dsc[s] = 0.f;
        if (scores_out) scores_out[(size_t)b * S + s] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        int64_t dec;
        const float loss = loss_and_dscores(sc, dsc, S, threshold, y ? y[b] : 0, y != nullptr, ls, &dec);
        if (dec_out) dec_out[b] = dec;
        if (dec_trace) dec_trace[b] = dec;
        if (y && success) success[b] = ls.targeted ? (dec == y[b]) : (dec != y[b]);
        if (loss_out) loss_out[b] = loss;
        if (loss_trace) loss_trace[b] = loss;
    }
    __syncthreads();
    if (!want_grad || !dact8) return;
    if (tid < 32) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) {
            const float w = dsc[s];
            if (w != 0.f) acc += w * fc_w[(size_t)s * 32 + tid];
        }
        demb[tid] = acc;
    }
    __syncthreads();
    // gradient wrt the pre-activation of conv8: only the arg-max frame of each channel, if it is > 0
    for (int i = tid; i < T8 * 32; i += 256) {
        const int t = i >> 5, c = i & 31;
        dact8[(size_t)b * T8 * 32 + i] = (t == arg[c] && emb[c] > 0.f) ? demb[c] : 0.f;
    }
}

hipError_t launch_an_logmel_fwd(const AnTables& t, const float* x, int B, int T, int F, const float* scale, float* feats,
                                hipStream_t s) {
    hipLaunchKernelGGL(an_logmel_fwd_kernel, dim3((F + kAnWavesPerBlock - 1) / kAnWavesPerBlock, B), dim3(128), 0, s, t, x,
                       T, F, scale, feats);
    return hipGetLastError();
}
hipError_t launch_an_logmel_bwd(const AnTables& t, const float* x, int B, int T, int F, const float* scale,
                                const float* dfeats, float* dframes, hipStream_t s) {
    hipLaunchKernelGGL(an_logmel_bwd_kernel, dim3((F + kAnWavesPerBlock - 1) / kAnWavesPerBlock, B), dim3(128), 0, s, t, x,
                       T, F, scale, dfeats, dframes);
    return hipGetLastError();
}
hipError_t launch_an_frames_to_wave(const float* dframes, int B, int T, int F, const float* scale, float* grad_out,
                                    float* x_io, const float* lower, const float* upper, float step, int grad_sign,
                                    hipStream_t s) {
    hipLaunchKernelGGL(an_frames_to_wave_kernel, dim3((T + 255) / 256, B), dim3(256), 0, s, dframes, T, F, scale, grad_out,
                       x_io, lower, upper, step, grad_sign);
    return hipGetLastError();
}
hipError_t launch_an_prefilter(const float* in, float* out, int B, int T, const float* w25, float bias, int transpose,
                               hipStream_t s) {
    hipLaunchKernelGGL(an_prefilter_kernel, dim3((T * kAnMel + 255) / 256, B), dim3(256), 0, s, in, out, T, w25, bias,
                       transpose);
    return hipGetLastError();
}
hipError_t launch_an_pool_fwd(const float* in, float* out, int B, int Tin, int C, hipStream_t s) {
    hipLaunchKernelGGL(an_pool_fwd_kernel, dim3(((Tin / 2) * C + 255) / 256, B), dim3(256), 0, s, in, out, Tin, C);
    return hipGetLastError();
}
hipError_t launch_an_pool_bwd(const float* act, const float* dpool, float* dact, int B, int Tin, int C, hipStream_t s) {
    hipLaunchKernelGGL(an_pool_bwd_kernel, dim3((Tin * C + 255) / 256, B), dim3(256), 0, s, act, dpool, dact, Tin, C);
    return hipGetLastError();
}
hipError_t launch_an_tail(const float* act8, int B, int T8, const float* fc_w, const float* fc_b, int S, float threshold,
                          const int64_t* y, const sg_loss_spec& ls, int want_grad, float* emb, float* scores,
                          int64_t* decisions, float* loss, float* dact8, float* loss_trace, int64_t* dec_trace,
                          uint8_t* success, hipStream_t s) {
    if (S < 1 || S > kLossMaxS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(an_tail_kernel, dim3(B), dim3(256), 0, s, act8, T8, fc_w, fc_b, S, threshold, y, ls, want_grad, emb,
                       scores, decisions, loss, dact8, loss_trace, dec_trace, success);
    return hipGetLastError();
}

}  // namespace sg
