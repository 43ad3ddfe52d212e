// AudioNet CSI-NE kernels other than the 1-D convolutions (those run on conv_gemm_kernel):
//   log-mel front-end and its hand-coded backward   reference model/_audionet/Preprocessor.py:85-112
//   5x5 pre-filter (Conv2d + BatchNorm2d, folded)   model/audionet_csine.py:66-71,181-185
//   MaxPool1d(2) forward / un-pool + ReLU mask      :77,94,111
//   max over time, Linear, decision, loss, backward :205-211,246-257 + attack/utils.py losses
// Activations are channel-last (B, T, C) like everywhere else in the library.
#include <cstdio>
#include <cstdlib>

#include "loss_device.h"
#include "sg_internal.h"
#include "fft512.h"

namespace sg {

constexpr int kAnWavesPerBlock = 4;
constexpr int kAnMaxBlocks = 512;   // 2 blocks (51 KB of LDS each) per CU x 256 CUs
constexpr int kAnHalf = kAnFft / 2; // 512: a 1024-point REAL frame is one 512-point complex transform + a split step
constexpr int kAnMelLaneBins = 44;  // >= bins per half filter (exactly 44 for 32 slaney filters over 513 bins; host-checked)

__device__ __forceinline__ float an_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Log-mel front-end, one wave per frame, persistent over frames (same recipe as the MFCC kernels, k_mfcc.hip):
//   * the 1024 real samples x[q] of a frame are packed as z[n] = x[2n] + i x[2n+1] and transformed by ONE 512-point
//     complex FFT (fft512.h: in-register radix-8, padded conflict-free LDS buffer, fp64); the spectrum follows from
//         X[k] = A_k Z[k] + B_k conj(Z[512-k]),   A_k = (1 - i W^k)/2,  B_k = (1 + i W^k)/2,  W = exp(-2 pi i/1024),
//     for k = 0..512 -- half the butterflies and half the LDS of a 1024-point transform (the first version: radix-2,
//     320 LDS accesses per lane per transform with twiddles and bit-reversal tables read from global memory);
//   * the backward is the adjoint of exactly that: dZ[j] = conj(A_j) G[j] + B_{512-j} conj(G[512-j]) (+ the k = 512
//     terms folded into j = 0), one inverse 512-point transform, d x[2n] = Re dz[n], d x[2n+1] = Im dz[n];
//   * everything a lane needs for every frame is in registers for the whole kernel: its 16 window taps, the W^k of
//     its 4 spectrum pairs, the weights of its half mel filter (ascending bins, zero-padded), its bins' filter
//     membership; only the FFT twiddles live in LDS.
struct AnFrameLds {
    double2 spec[kAnHalf + kAnHalf / 8];  // element i at SP(i)
    float power[kAnBins + 3];
    float mel[32];
    float dmel[34];
};

// pre-emphasised sample p of utterance row xr (length T), p in [0, T-2]: x[p+1] - 0.97 x[p]
__device__ __forceinline__ float an_preemph(const float* __restrict__ xr, int p, float scale) {
    return (xr[p + 1] - 0.97f * xr[p]) * scale;
}
__device__ __forceinline__ int an_reflect(int p, int L) { return p < 0 ? -p : (p >= L ? 2 * (L - 1) - p : p); }

struct AnLaneConst {
    float win[16];            // window tap of FFT input q = 2 (lane + 64 i) + {0, 1} (0 outside the 800-tap window)
    double2 wk[4];            // W^j for this lane's pairs j = lane + 64 i
    int mel_k0;
    float mel_w[kAnMelLaneBins];
};

__device__ __forceinline__ void an_lane_init(const AnTables& t, int lane, AnLaneConst& lc) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = 2 * (lane + 64 * i) + h - (kAnFft - kAnWin) / 2;
            lc.win[2 * i + h] = (n >= 0 && n < kAnWin) ? t.window[n] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) lc.wk[i] = t.twiddle[lane + 64 * i];
    const int m = lane >> 1, h = lane & 1;
    const int lo = t.mel_lo[m], hi = t.mel_hi[m];
    const int mid = lo + (hi - lo + 1) / 2;
    const int k0 = h ? mid : lo, cnt = (h ? hi : mid) - k0;
    lc.mel_k0 = k0;
#pragma unroll
    for (int j = 0; j < kAnMelLaneBins; ++j) lc.mel_w[j] = j < cnt ? t.mel_w[m * kAnBins + min(k0 + j, kAnBins - 1)] : 0.f;
}

// raw samples of frame f: x[p], x[p + 1] for the 16 FFT inputs of this lane (loaded one frame ahead)
struct AnRaw {
    float a[16], b[16];
};
__device__ __forceinline__ void an_load_frame(const float* __restrict__ x, int T, int F, int gf, int total, int lane, AnRaw& r) {
    const int g = gf < total ? gf : total - 1;
    const int bb = g / F, f = g - bb * F;
    const float* xr = x + (size_t)bb * T;
    const int Lp = T - 1;                         // length of the pre-emphasised signal
    const int base = f * kAnHop - kAnWin / 2;     // centre=True: frame f is centred on sample f*hop
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = 2 * (lane + 64 * (i >> 1)) + (i & 1);
        int n = q - (kAnFft - kAnWin) / 2;
        n = n < 0 ? 0 : (n >= kAnWin ? kAnWin - 1 : n);  // outside the window the tap is 0: any in-range sample will do
        const int p = an_reflect(base + n, Lp);
        r.a[i] = xr[p];
        r.b[i] = xr[p + 1];
    }
}

__device__ __forceinline__ double2 an_cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 an_conj(double2 a) { return make_double2(a.x, -a.y); }

// X[k] and X[512 - k] from the pair (Z[k], Z[512 - k]); w = W^k.  A_k = (1 - i w)/2, B_k = (1 + i w)/2,
// and W^(512-k) = -conj(W^k).
__device__ __forceinline__ void an_split(double2 zk, double2 zr, double2 w, double2& xk, double2& xr) {
    const double2 iw = make_double2(-w.y, w.x);                       // i w
    const double2 ak = make_double2(0.5 * (1.0 - iw.x), -0.5 * iw.y);
    const double2 bk = make_double2(0.5 * (1.0 + iw.x), 0.5 * iw.y);
    const double2 t0 = an_cmul(ak, zk), t1 = an_cmul(bk, an_conj(zr));
    xk = make_double2(t0.x + t1.x, t0.y + t1.y);
    const double2 wr = make_double2(-w.x, w.y);                       // W^(512-k)
    const double2 iwr = make_double2(-wr.y, wr.x);
    const double2 ar = make_double2(0.5 * (1.0 - iwr.x), -0.5 * iwr.y);
    const double2 br = make_double2(0.5 * (1.0 + iwr.x), 0.5 * iwr.y);
    const double2 u0 = an_cmul(ar, zr), u1 = an_cmul(br, an_conj(zk));
    xr = make_double2(u0.x + u1.x, u0.y + u1.y);
}

// packed spectrum Z of the frame into L.spec (element i at SP(i)), power of bins 0..512 into L.power, mel into L.mel
// WITH_MEL = false: only the packed spectrum (the cached backward takes the mel energies from the forward pass)
template <bool WITH_MEL>
__device__ __forceinline__ void an_frame_forward(const double2* tw1, const double2* tw2, AnFrameLds& L, const AnLaneConst& lc,
                                                 const AnRaw& r, float scale, int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v0 = (r.b[2 * i] - 0.97f * r.a[2 * i]) * scale * lc.win[2 * i];
        const float v1 = (r.b[2 * i + 1] - 0.97f * r.a[2 * i + 1]) * scale * lc.win[2 * i + 1];
        L.spec[SP(lane + 64 * i)] = make_double2((double)v0, (double)v1);
    }
    wave_sync();
    fft512_r8_t(L.spec, tw1, tw2, lane, -1.0);
    if (!WITH_MEL) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane + 64 * i;  // pairs (k, 512 - k), k = 0..255
        const double2 zk = L.spec[SP(k)], zr = L.spec[SP((kAnHalf - k) & (kAnHalf - 1))];
        double2 xk, xr;
        an_split(zk, zr, lc.wk[i], xk, xr);
        L.power[k] = (float)(xk.x * xk.x + xk.y * xk.y);
        L.power[kAnHalf - k] = (float)(xr.x * xr.x + xr.y * xr.y);  // k = 0: X[512] = Re Z[0] - Im Z[0] (split with w = 1)
    }
    if (lane == 0) {  // the self-paired bin 256: w = W^256 = -i
        const double2 z = L.spec[SP(256)];
        double2 xk, xr;
        an_split(z, z, make_double2(0.0, -1.0), xk, xr);
        L.power[256] = (float)(xk.x * xk.x + xk.y * xk.y);
    }
    wave_sync();
    // 32 slaney-mel filters, two lanes per filter (each takes half of the filter's bin range, ascending)
    {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < kAnMelLaneBins; ++j) acc += L.power[min(lc.mel_k0 + j, kAnBins - 1)] * lc.mel_w[j];
        acc += __shfl_xor(acc, 1, 64);
        if ((lane & 1) == 0) L.mel[lane >> 1] = acc;
    }
    wave_sync();
}

// the 512-point transform's twiddles as its lanes read them (fft512.h: conflict-free tables), from the half circle of
// W512^i = W1024^(2 i) staged in tw512
__device__ __forceinline__ void an_stage_tw(const AnTables& t, double2* tw512, double2* tw1, double2* tw2) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) tw512[i] = t.twiddle[2 * i];
    __syncthreads();
    fft512_fill_tables(tw512, tw1, tw2);
    __syncthreads();
}

__global__ __launch_bounds__(256, 2) void an_logmel_fwd_kernel(AnTables t, const float* __restrict__ x, int B, int T, int F,
                                                            const float* __restrict__ scale_p, float* __restrict__ feats) {
    __shared__ AnFrameLds lds[kAnWavesPerBlock];
    __shared__ double2 tw512[256], tw1[kFftTw1], tw2[kFftTw2];
    an_stage_tw(t, tw512, tw1, tw2);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float scale = scale_p ? *scale_p : 1.f;
    AnFrameLds& L = lds[wid];
    AnLaneConst lc;
    an_lane_init(t, lane, lc);
    const int total = B * F, stride = gridDim.x * kAnWavesPerBlock;
    AnRaw cur;  // (a one-frame-ahead prefetch of these 32 registers cost the second wave per SIMD)
    for (int gf = blockIdx.x * kAnWavesPerBlock + wid; gf < total; gf += stride) {
        an_load_frame(x, T, F, gf, total, lane, cur);
        an_frame_forward<true>(tw1, tw2, L, lc, cur, scale, lane);
        if (lane < kAnMel) {
            feats[(size_t)gf * kAnMel + lane] = 10.f * log10f(fmaxf(L.mel[lane], 1e-16f));
            if (t.mel_cache) t.mel_cache[(size_t)gf * kAnMel + lane] = L.mel[lane];
        }
        wave_sync();
    }
}

// dfeats (B,F,32) -> dframes (B,F,800): gradient wrt the pre-emphasised, reflect-padded frame samples
// CACHED: the forward kernel of the same pass left the mel energies in t.mel_cache (the attack loop); otherwise they
// are recomputed (standalone sg_an_logmel_backward).  Two instantiations: the cached one does not carry the 44
// mel-weight registers.
template <bool CACHED>
__global__ __launch_bounds__(256, 2) void an_logmel_bwd_kernel(AnTables t, const float* __restrict__ x, int B, int T, int F,
                                                            const float* __restrict__ scale_p,
                                                            const float* __restrict__ dfeats, float* __restrict__ dframes) {
    __shared__ AnFrameLds lds[kAnWavesPerBlock];
    __shared__ double2 tw512[256], tw1[kFftTw1], tw2[kFftTw2];
    an_stage_tw(t, tw512, tw1, tw2);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float scale = scale_p ? *scale_p : 1.f;
    AnFrameLds& L = lds[wid];
    AnLaneConst lc;
    an_lane_init(t, lane, lc);
    // filter membership of every bin (which mel filter pair it feeds, with which weights): LDS, shared by the block
    __shared__ int bin_m0[kAnBins + 3];
    __shared__ float bin_w0[kAnBins + 3], bin_w1[kAnBins + 3];
    for (int i = threadIdx.x; i < kAnBins; i += blockDim.x) {
        bin_m0[i] = t.bin_m0[i];
        bin_w0[i] = t.bin_w0[i];
        bin_w1[i] = t.bin_w1[i];
    }
    __syncthreads();
    const int total = B * F, stride = gridDim.x * kAnWavesPerBlock;
    AnRaw cur;  // (a one-frame-ahead prefetch of these 32 registers cost the second wave per SIMD)
    for (int gf = blockIdx.x * kAnWavesPerBlock + wid; gf < total; gf += stride) {
        an_load_frame(x, T, F, gf, total, lane, cur);
        an_frame_forward<!CACHED>(tw1, tw2, L, lc, cur, scale, lane);
        if (lane < 34) {
            float dm = 0.f;
            if (lane < kAnMel) {
                const float mel = CACHED ? t.mel_cache[(size_t)gf * kAnMel + lane] : L.mel[lane];
                // d/d mel of 10 log10(max(mel, 1e-16))
                dm = mel > 1e-16f ? dfeats[(size_t)gf * kAnMel + lane] * (10.f / 2.302585092994046f) / mel : 0.f;
            }
            L.dmel[lane] = dm;
        }
        wave_sync();
        // G[k] = 2 X[k] dP[k] for both bins of every pair, folded straight into dZ (in place: a lane owns its pair)
        auto dpow = [&](int k) {
            const int m0 = bin_m0[k];
            return m0 >= 0 ? 2.0 * (double)(L.dmel[m0] * bin_w0[k] + L.dmel[m0 + 1] * bin_w1[k]) : 0.0;
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane + 64 * i;
            const int kr = (kAnHalf - k) & (kAnHalf - 1);
            const double2 zk = L.spec[SP(k)], zr = L.spec[SP(kr)];
            const double2 w = lc.wk[i];
            double2 xk, xr;
            an_split(zk, zr, w, xk, xr);
            const double dpk = dpow(k), dpr = dpow(kAnHalf - k);
            const double2 gk = make_double2(xk.x * dpk, xk.y * dpk), gr = make_double2(xr.x * dpr, xr.y * dpr);
            const double2 iw = make_double2(-w.y, w.x);
            const double2 ak = make_double2(0.5 * (1.0 - iw.x), -0.5 * iw.y), bk = make_double2(0.5 * (1.0 + iw.x), 0.5 * iw.y);
            const double2 wr = make_double2(-w.x, w.y);
            const double2 iwr = make_double2(-wr.y, wr.x);
            const double2 ar = make_double2(0.5 * (1.0 - iwr.x), -0.5 * iwr.y), br = make_double2(0.5 * (1.0 + iwr.x), 0.5 * iwr.y);
            // dZ[k] = conj(A_k) G[k] + B_{512-k} conj(G[512-k]);  dZ[512-k] = conj(A_{512-k}) G[512-k] + B_k conj(G[k])
            const double2 d0 = an_cmul(an_conj(ak), gk), d1 = an_cmul(br, an_conj(gr));
            const double2 e0 = an_cmul(an_conj(ar), gr), e1 = an_cmul(bk, an_conj(gk));
            if (k == 0) {
                // bins 0 and 512 both fold onto dZ[0]
                L.spec[SP(0)] = make_double2(d0.x + d1.x + e0.x + e1.x, d0.y + d1.y + e0.y + e1.y);
            } else {
                L.spec[SP(k)] = make_double2(d0.x + d1.x, d0.y + d1.y);
                L.spec[SP(kr)] = make_double2(e0.x + e1.x, e0.y + e1.y);
            }
        }
        wave_sync();
        if (lane == 0) {  // self-paired bin 256
            const double2 z = L.spec[SP(256)];
            const double2 w = make_double2(0.0, -1.0);
            double2 xk, xr;
            an_split(z, z, w, xk, xr);
            const double dp = dpow(256);
            const double2 g = make_double2(xk.x * dp, xk.y * dp);
            const double2 iw = make_double2(-w.y, w.x);
            const double2 ak = make_double2(0.5 * (1.0 - iw.x), -0.5 * iw.y), bk = make_double2(0.5 * (1.0 + iw.x), 0.5 * iw.y);
            const double2 d0 = an_cmul(an_conj(ak), g), d1 = an_cmul(bk, an_conj(g));
            L.spec[SP(256)] = make_double2(d0.x + d1.x, d0.y + d1.y);
        }
        wave_sync();
        fft512_r8_t(L.spec, tw1, tw2, lane, 1.0);
        float* out = dframes + (size_t)gf * kAnWin;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = lane + 64 * i;
            const int q = 2 * n - (kAnFft - kAnWin) / 2;  // window index of FFT input 2n (even, since (1024-800)/2 = 112)
            const double2 dz = L.spec[SP(n)];
            if (q >= 0 && q < kAnWin)
                *reinterpret_cast<float2*>(out + q) = make_float2((float)dz.x * lc.win[2 * i], (float)dz.y * lc.win[2 * i + 1]);
        }
        wave_sync();
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
    // every d pre value is gathered once per block (up to 5 frame reads) and shared through LDS: d x[t] needs
    // d pre[t-1] and d pre[t] -- the first version gathered both per thread and read the 491 MB of per-frame
    // gradients of a batch-512 step twice
    __shared__ float dp[257];  // dp[i] = d pre[t0 - 1 + i]
    const int t0 = blockIdx.x * 256;
    {
        const int sidx = t0 + (int)threadIdx.x;  // d pre[s], stored at dp[threadIdx.x + 1]
        dp[threadIdx.x + 1] = sidx <= Lp - 1 ? dpre(sidx) : 0.f;
        if (threadIdx.x == 0) dp[0] = t0 >= 1 ? dpre(t0 - 1) : 0.f;
    }
    __syncthreads();
    if (t >= T) return;
    float g = 0.f;
    if (t >= 1) g += dp[threadIdx.x];
    if (t <= Lp - 1) g -= 0.97f * dp[threadIdx.x + 1];
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
            // branch-free: clamped address + select.  A guarded load makes hipcc wait vmcnt(0) at every join, i.e.
            // 25 serial L2 round trips per output (56 us per launch at batch 512 for 20 MB of data).
            const bool ok = mm >= 0 && mm < kAnMel && tt >= 0 && tt < T;
            const float v = x[(size_t)min(max(tt, 0), T - 1) * kAnMel + min(max(mm, 0), kAnMel - 1)];
            // an explicit fused multiply-add: left to -ffp-contract the compiler fused the main body of a loop and not its
            // remainder (k_audionet_fused.hip must reproduce these bits whatever shape its loops take)
            acc = fmaf(w25[i * 5 + j], ok ? v : 0.f, acc);
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
                                                      int64_t* __restrict__ dec_trace, uint8_t* __restrict__ success, int coef_rows,
                                                      unsigned long long* __restrict__ trace) {
#define ATSTAMP(i) if (trace && threadIdx.x == 0 && blockIdx.x == 0) trace[i] = __builtin_amdgcn_s_memrealtime();
    ATSTAMP(0)
    __shared__ float emb[32];
    __shared__ int arg[32];
    __shared__ float sc[kLossMaxS], dsc[kLossMaxS];
    __shared__ float demb[32];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* a = act8 + (size_t)b * T8 * 32;
    {   // x.max(2): first maximum.  Eight time slices per channel (thread = (slice, channel): 32 consecutive floats
        // per load), each a short chain of loads instead of one chain of T8; slices merged in time order with a strict
        // >, so the earliest frame still wins ties.
        __shared__ float pmx[8][32];
        __shared__ int pat[8][32];
        const int c = tid & 31, sl = tid >> 5;
        const int t0 = (int)((long long)T8 * sl / 8), t1 = (int)((long long)T8 * (sl + 1) / 8);
        float mx = -INFINITY;
        int at = t0;
        for (int t = t0; t < t1; ++t) {
            const float v = a[(size_t)t * 32 + c];
            if (v > mx) { mx = v; at = t; }
        }
        pmx[sl][c] = mx;
        pat[sl][c] = at;
        __syncthreads();
        if (tid < 32) {
            mx = pmx[0][tid];
            at = pat[0][tid];
#pragma unroll
            for (int i = 1; i < 8; ++i)
                if (pmx[i][tid] > mx) { mx = pmx[i][tid]; at = pat[i][tid]; }
            emb[tid] = mx;
            arg[tid] = at;
            if (emb_out) emb_out[(size_t)b * 32 + tid] = mx;
        }
    }
    __syncthreads();
    ATSTAMP(1)
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
    ATSTAMP(2)
    {
        __shared__ float ex[kLossMaxS];
        __shared__ float bc[4];
        int64_t dec = 0;
        const float loss = loss_and_dscores_block(sc, dsc, ex, bc, S, threshold, y ? y[b] : 0, y != nullptr, ls, &dec, tid, 256,
                                                  ls.coef_dev ? ls.coef_dev + (size_t)(coef_rows > 0 ? b % coef_rows : b) * S : nullptr);
        if (tid == 0) {
            if (dec_out) dec_out[b] = dec;
            if (dec_trace) dec_trace[b] = dec;
            if (y && success) success[b] = ls.targeted ? (dec == y[b]) : (dec != y[b]);
            if (loss_out) loss_out[b] = loss;
            if (loss_trace) loss_trace[b] = loss;
        }
    }
    __syncthreads();
    ATSTAMP(3)
    if (!want_grad || !dact8) return;
    {   // d emb[c] = sum_s dsc[s] fc_w[s][c]: 8 class-strided partial sums per channel, combined in a fixed order,
        // loads unguarded (a guarded load in this loop serialised 251 L2 round trips per utterance)
        __shared__ float part[8][32];
        const int c = tid & 31, q = tid >> 5;
        float acc = 0.f;
        for (int s = q; s < S; s += 8) acc += dsc[s] * fc_w[(size_t)s * 32 + c];
        part[q][c] = acc;
        __syncthreads();
        if (tid < 32) {
            float r = part[0][tid];
#pragma unroll
            for (int i = 1; i < 8; ++i) r += part[i][tid];
            demb[tid] = r;
        }
    }
    __syncthreads();
    ATSTAMP(4)
    // gradient wrt the pre-activation of conv8: only the arg-max frame of each channel, if it is > 0
    for (int i = tid; i < T8 * 32; i += 256) {
        const int t = i >> 5, c = i & 31;
        dact8[(size_t)b * T8 * 32 + i] = (t == arg[c] && emb[c] > 0.f) ? demb[c] : 0.f;
    }
    ATSTAMP(5)
#undef ATSTAMP
}

hipError_t launch_an_logmel_fwd(const AnTables& t, const float* x, int B, int T, int F, const float* scale, float* feats,
                                hipStream_t s) {
    const int want = (B * F + kAnWavesPerBlock - 1) / kAnWavesPerBlock;
    hipLaunchKernelGGL(an_logmel_fwd_kernel, dim3(want < kAnMaxBlocks ? want : kAnMaxBlocks), dim3(256), 0, s, t, x, B, T, F,
                       scale, feats);
    return hipGetLastError();
}
hipError_t launch_an_logmel_bwd(const AnTables& t, const float* x, int B, int T, int F, const float* scale,
                                const float* dfeats, float* dframes, hipStream_t s) {
    const int want = (B * F + kAnWavesPerBlock - 1) / kAnWavesPerBlock;
    const dim3 grid(want < kAnMaxBlocks ? want : kAnMaxBlocks);
    if (t.mel_cache) hipLaunchKernelGGL(an_logmel_bwd_kernel<true>, grid, dim3(256), 0, s, t, x, B, T, F, scale, dfeats, dframes);
    else hipLaunchKernelGGL(an_logmel_bwd_kernel<false>, grid, dim3(256), 0, s, t, x, B, T, F, scale, dfeats, dframes);
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
                          uint8_t* success, hipStream_t s, int coef_rows) {
    if (S < 1 || S > kLossMaxS) return hipErrorInvalidValue;
    // tuning aid: SG_AN_TAIL_TRACE=1 prints the phase timestamps (100 MHz) of block 0 after every launch (synchronises)
    static const bool tr_on = getenv("SG_AN_TAIL_TRACE") != nullptr;
    static unsigned long long* tr_dev = nullptr;
    if (tr_on && !tr_dev) (void)hipMalloc(reinterpret_cast<void**>(&tr_dev), 8 * 8);
    hipLaunchKernelGGL(an_tail_kernel, dim3(B), dim3(256), 0, s, act8, T8, fc_w, fc_b, S, threshold, y, ls, want_grad, emb,
                       scores, decisions, loss, dact8, loss_trace, dec_trace, success, coef_rows, tr_on ? tr_dev : nullptr);
    if (tr_on && tr_dev && hipStreamSynchronize(s) == hipSuccess) {
        unsigned long long h[8];
        if (hipMemcpy(h, tr_dev, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "an_tail phases (us): max-over-time %.2f  fc %.2f  loss %.2f  d emb %.2f  d act8 %.2f  total %.2f\n", (h[1] - h[0]) * 0.01,
                    (h[2] - h[1]) * 0.01, (h[3] - h[2]) * 0.01, (h[4] - h[3]) * 0.01, (h[5] - h[4]) * 0.01, (h[5] - h[0]) * 0.01);
        }
    }
    return hipGetLastError();
}

}  // namespace sg
