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
#include "fft512t.h"

namespace sg {

constexpr int kAnWavesPerBlock = 4;
constexpr int kAnCus = 256;
constexpr int kAnHalf = kAnFft / 2; // 512: a 1024-point REAL frame is one 512-point complex transform + a split step
constexpr int kAnMelLaneBins = 20;  // bins per lane of the mel stage: a filter's bins in runs of <= 20 (an_build_tables)

__device__ __forceinline__ float an_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#pragma clang fp contract(off)  // see fft512t.h: every fused multiply-add of the log-mel front-end is written out

// Log-mel front-end, one wave per frame (same recipe as the MFCC kernels, k_mfcc.hip):
//   * the 1024 real samples x[q] of a frame are packed as z[n] = x[2n] + i x[2n+1] and transformed by ONE 512-point
//     complex FFT (fft512t.h: in-register radix-8, padded conflict-free LDS buffer); the spectrum follows from
//         X[k] = A_k Z[k] + B_k conj(Z[512-k]),   A_k = (1 - i W^k)/2,  B_k = (1 + i W^k)/2,  W = exp(-2 pi i/1024),
//     for k = 0..512 -- half the butterflies and half the LDS of a 1024-point transform;
//   * the backward is the adjoint of exactly that: dZ[j] = conj(A_j) G[j] + B_{512-j} conj(G[512-j]) (+ the k = 512
//     terms folded into j = 0), one inverse 512-point transform, d x[2n] = Re dz[n], d x[2n+1] = Im dz[n];
//   * everything a lane needs for every frame is in registers for the whole kernel: its 16 window taps, the W^k of
//     its 4 spectrum pairs, the weights of its half mel filter (ascending bins, zero-padded), its bins' filter
//     membership; only the FFT twiddles live in LDS.
// Round 5: the scalar type R of the transforms is a template parameter.  float: what the reference computes in
// (Preprocessor.py:100-105: torch.stft on a float32 signal) -- half the LDS bytes of the exchange passes these kernels are
// bound by; double: the form of rounds 1-4 (sg_an_configure).  Twiddles are float64 values rounded once in both.
template <typename R, bool POWER>
struct AnFrameLdsT {
    union {
        cx<R> spec[kAnHalf + kAnHalf / 8];  // the transform's exchange buffer, element i at SP(i)
        // the power spectrum shares it (the transform is done with the buffer when the powers are written: 4 blocks per CU
        // instead of 3); bins past 512 are zeroed every frame: a lane's 20 mel taps need no clamp
        float power[POWER ? kAnBins + 3 + kAnMelLaneBins : 4];
    };
    float mel[POWER ? 32 : 1];
    float dmel[36];
    float part[POWER ? 64 : 1];  // the mel stage's per-lane sums
};

__device__ __forceinline__ int an_reflect(int p, int L) { return p < 0 ? -p : (p >= L ? 2 * (L - 1) - p : p); }

// Per-lane constants of a block's waves.  The window taps and the half mel filter of a lane are the same in every wave:
// they sit in block-shared LDS tables read lane-linearly (round 5: 60 registers less per thread -- three blocks per CU
// instead of two for the float32 kernels, whose stalls are LDS and L2 round trips, not arithmetic).
template <bool MEL>
struct AnLaneTab {
    float win[16][64];                        // [tap of FFT input q = 2 (lane + 64 i) + {0, 1}][lane] (0 outside the 800-tap window)
    float mel_w[MEL ? kAnMelLaneBins : 1][64];  // [tap][lane]: weights of the lane's run of mel-filter bins, ascending, zero-padded
};
template <typename R>
struct AnLaneConstT {
    const float* win;         // &tab.win[0][lane]: tap i at win[64 i]
    const float* mel_w;       // &tab.mel_w[0][lane]
    cx<R> c2[4];              // -i W^e / 2 for the lane's pairs in transform order, e = (lane >> 3) + 8 (lane & 7) + 64 d, d < 4
    int m[8];                 // an_lane_offsets
    int mel_k0;
    int mel_seg;              // lane m < 32: first lane | runs << 8 of filter m
};
// Pairs in transform order (round 5).  After fft512T_regs lane l = (k1, c) holds Z[e_d], e_d = k1 + 8 c + 64 d.  The bin that
// pairs with e_d, 512 - e_d, is element 7 - d of ONE other lane: (8 - k1, 7 - c) = lane 71 - l for l >= 8, lane 8 - l for
// 1 <= l <= 7 (lane 4 pairs with itself).  Lane 0 holds the multiples of 64 and pairs with itself at 8 - d (d = 0: bins 0 and
// 512, d = 4: the self-paired bin 256).  So a lane takes the pairs of its elements d = 0..3 (every pair exactly once over the
// wave) and the partner values cross by ds_bpermute: no LDS buffer, no bank conflicts, half the bytes of a write + read.
__device__ __forceinline__ int an_partner_lane(int lane) { return lane == 0 ? 0 : (lane < 8 ? 8 - lane : 71 - lane); }
__device__ __forceinline__ float an_from_lane(int lane4, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(lane4, __float_as_int(v)));
}
__device__ __forceinline__ double an_from_lane(int lane4, double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(lane4, (int)(b & 0xffffffffll)), hi = __builtin_amdgcn_ds_bpermute(lane4, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <typename R>
__device__ __forceinline__ cx<R> an_from_lane(int lane4, cx<R> v) {
    return cmk<R>(an_from_lane(lane4, v.x), an_from_lane(lane4, v.y));
}

__device__ __forceinline__ void an_lane_offsets(int lane, int (&m)[8]);
// fills the block's table (all threads; the caller synchronises before the first frame) and the lane's registers
template <typename R, bool MEL>
__device__ __forceinline__ void an_lane_init(const AnTables& t, int lane, AnLaneTab<MEL>& tab, AnLaneConstT<R>& lc) {
    // the tables arrive laid out per lane (sg_api_audionet.hip an_build_tables): two linear copies
    for (int e = threadIdx.x; e < 16 * 64; e += blockDim.x) (&tab.win[0][0])[e] = t.lane_win[e];
    if constexpr (MEL)
        for (int e = threadIdx.x; e < kAnMelLaneBins * 64; e += blockDim.x) (&tab.mel_w[0][0])[e] = t.lane_melw[e];
    lc.win = &tab.win[0][lane];
    lc.mel_w = &tab.mel_w[0][lane];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const double2 w = t.twiddle[(lane >> 3) + 8 * (lane & 7) + 64 * d];
        lc.c2[d] = cmk<R>((R)(0.5 * w.y), (R)(-0.5 * w.x));
    }
    an_lane_offsets(lane, lc.m);
    lc.mel_k0 = MEL ? t.lane_k0[lane] : 0;
    lc.mel_seg = MEL && lane < kAnMel ? t.mel_seg[lane] : 0;
}

// raw samples of frame f: x[p], x[p + 1] for the 16 FFT inputs of this lane
struct AnRaw {
    float a[16], b[16];
};
// Round 5 (the float32 kernels are bound by VALU issue, and two thirds of what they issued was index arithmetic): a frame
// that does not touch the reflected ends reads x[base + m_i + {0, 1, 2}] for the lane's 8 pairs of FFT inputs, m_i a lane
// constant (kept by the caller) -- the frame's base address is wave-uniform, so the loads need no per-lane address
// arithmetic at all.  A tap outside the 800-sample window has weight 0: any finite sample will do there (m_i is clamped).
__device__ __forceinline__ void an_lane_offsets(int lane, int (&m)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = 2 * (lane + 64 * i) - (kAnFft - kAnWin) / 2;
        m[i] = n < 0 ? 0 : (n > kAnWin - 2 ? kAnWin - 2 : n);  // taps n, n + 1 and the sample after them
    }
}
__device__ __forceinline__ void an_load_frame(const float* __restrict__ xr, int T, int f, int lane, const int (&m)[8], AnRaw& r) {
    const int Lp = T - 1;                         // length of the pre-emphasised signal
    const int base = f * kAnHop - kAnWin / 2;     // centre=True: frame f is centred on sample f*hop
    if (base >= 0 && base + kAnWin <= Lp) {       // x[base .. base + 800] exist and no position is reflected (wave-uniform)
        const float* xb = xr + base;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v0 = xb[m[i]], v1 = xb[m[i] + 1], v2 = xb[m[i] + 2];
            r.a[2 * i] = v0;
            r.b[2 * i] = v1;
            r.a[2 * i + 1] = v1;
            r.b[2 * i + 1] = v2;
        }
        return;
    }
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

// X[k] and X[512 - k] from the pair (Z[k], Z[512 - k]): X[k] = A_k Z[k] + B_k conj(Z[512-k]), A_k = (1 - i W^k)/2,
// B_k = (1 + i W^k)/2, and W^(512-k) = -conj(W^k) -- with the common factors taken out (round 5: 12 operations, not ~30):
//   e = (Z[k] + conj Z[512-k]) / 2,  t = c2 (Z[k] - conj Z[512-k]),  c2 = -i W^k / 2:   X[k] = e + t,  X[512-k] = conj(e - t)
template <typename R>
__device__ __forceinline__ void an_split2(cx<R> zk, cx<R> zr, cx<R> c2, cx<R>& xk, cx<R>& xr) {
    const R half = (R)0.5;
    const cx<R> s = cmk<R>(zk.x + zr.x, zk.y - zr.y), d = cmk<R>(zk.x - zr.x, zk.y + zr.y);
    const cx<R> t = cmulT<R>(c2, d);
    xk = cmk<R>(fmaT(half, s.x, t.x), fmaT(half, s.y, t.y));
    xr = cmk<R>(fmaT(half, s.x, -t.x), fmaT(-half, s.y, t.y));
}
// ... and its adjoint: from gk = d L / d X[k], gr = d L / d X[512-k] to d L / d Z[k], d L / d Z[512-k]
//   d e = gk + conj gr,  d t = gk - conj gr:   d Z[k] = d e / 2 + conj(c2) d t,   d Z[512-k] = conj(d e / 2 - conj(c2) d t)
template <typename R>
__device__ __forceinline__ void an_split2_adjoint(cx<R> gk, cx<R> gr, cx<R> c2, cx<R>& dzk, cx<R>& dzr) {
    const R half = (R)0.5;
    const cx<R> de = cmk<R>(gk.x + gr.x, gk.y - gr.y), dt = cmk<R>(gk.x - gr.x, gk.y + gr.y);
    const cx<R> gd = cmulT<R>(cconjT<R>(c2), dt);
    dzk = cmk<R>(fmaT(half, de.x, gd.x), fmaT(half, de.y, gd.y));
    dzr = cmk<R>(fmaT(half, de.x, -gd.x), fmaT(-half, de.y, gd.y));
}

// One frame forward: the packed spectrum Z in TRANSFORM order in z (lane (k1, c): z[d] = Z[k1 + 8 c + 64 d]); WITH_MEL: power
// of bins 0..512 into L.power, mel into L.mel (false: only the packed spectrum -- the cached backward takes the mel energies
// from the forward pass)
template <typename R, bool WITH_MEL, bool POWER>
__device__ __forceinline__ void an_frame_forward(const cx<R>* tw1, const cx<R>* tw2, AnFrameLdsT<R, POWER>& L, const AnLaneConstT<R>& lc,
                                                 const AnRaw& r, float scale, int lane, cx<R> (&z)[8]) {
    static_assert(POWER || !WITH_MEL, "the mel energies need the power buffer");
    cx<R> in[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v0 = fmaT(-0.97f, r.a[2 * i], r.b[2 * i]) * scale * lc.win[64 * (2 * i)];
        const float v1 = fmaT(-0.97f, r.a[2 * i + 1], r.b[2 * i + 1]) * scale * lc.win[64 * (2 * i + 1)];
        in[i] = cmk<R>((R)v0, (R)v1);
    }
    fft512T_regs<R>(L.spec, tw1, tw2, lane, (R)-1, in, z);
    if constexpr (WITH_MEL) {
        const int e0 = (lane >> 3) + 8 * (lane & 7), p4 = 4 * an_partner_lane(lane);
        const bool l0 = lane == 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            // what this lane shows its partner for the partner's pair d: element 7 - d (lane 0, to itself: 8 - d, Z[0] for d = 0)
            const cx<R> show = l0 ? z[(8 - d) & 7] : z[7 - d];
            const cx<R> zr = an_from_lane<R>(p4, show);
            cx<R> xk, xr;
            an_split2<R>(z[d], zr, lc.c2[d], xk, xr);
            L.power[e0 + 64 * d] = (float)fmaT(xk.x, xk.x, xk.y * xk.y);
            L.power[kAnHalf - e0 - 64 * d] = (float)fmaT(xr.x, xr.x, xr.y * xr.y);  // lane 0, d = 0: X[512] = Re Z[0] - Im Z[0]
        }
        if (l0) L.power[256] = (float)fmaT(z[4].x, z[4].x, z[4].y * z[4].y);  // the self-paired bin: X[256] = conj Z[256]
        if (lane < 3 + kAnMelLaneBins) L.power[kAnBins + lane] = 0.f;  // the pad the mel taps may reach into (weight 0 there)
        wave_sync();
        // 32 slaney-mel filters: a lane sums one run of <= 20 consecutive bins of one filter (ascending), lane m < 32 the runs
        // of filter m (ascending)
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < kAnMelLaneBins; ++j) acc = fmaT(L.power[lc.mel_k0 + j], lc.mel_w[64 * j], acc);  // weights past the run are 0
        L.part[lane] = acc;
        wave_sync();
        if (lane < kAnMel) {
            const int first = lc.mel_seg & 255, runs = lc.mel_seg >> 8;
            float m = L.part[first];
#pragma unroll
            for (int i = 1; i < 5; ++i)
                if (i < runs) m += L.part[first + i];
            L.mel[lane] = m;
        }
        wave_sync();
    }
}

// the 512-point transform's twiddles as its lanes read them (fft512.h: conflict-free tables): laid out and rounded by the
// host (an_build_tables), copied to LDS
template <typename R>
__device__ __forceinline__ void an_stage_tw(const AnTables& t, cx<R>* tw1, cx<R>* tw2) {
    const cx<R>* s1 = sizeof(R) == 4 ? reinterpret_cast<const cx<R>*>(t.tw1f) : reinterpret_cast<const cx<R>*>(t.tw1d);
    const cx<R>* s2 = sizeof(R) == 4 ? reinterpret_cast<const cx<R>*>(t.tw2f) : reinterpret_cast<const cx<R>*>(t.tw2d);
    for (int i = threadIdx.x; i < kFftTw1; i += blockDim.x) tw1[i] = s1[i];
    for (int i = threadIdx.x; i < kFftTw2; i += blockDim.x) tw2[i] = s2[i];
}

// Which frames a wave of the persistent front-end kernels takes.  Blocks are dispatched round robin over the 8 XCDs, each
// with its own L2: dealing frames out by block index makes every XCD touch every utterance (8 x 98 MB of waveform through the
// fabric per pass at 512 utterances).  Instead XCD x = block % 8 takes the x-th eighth of the frames, its blocks striding
// through that range -- neighbouring frames (which share 4/5 of their samples) meet in one L2.
struct AnFrameRange {
    int first, end, stride;
};
__device__ __forceinline__ AnFrameRange an_frame_range(int total, int wid) {
    const int nb = gridDim.x, b = blockIdx.x;
    if (nb < 16 || (nb & 7)) return {b * kAnWavesPerBlock + wid, total, nb * kAnWavesPerBlock};
    const int xcd = b & 7, local = b >> 3, nloc = nb >> 3;
    const int lo = (int)((long long)total * xcd / 8), hi = (int)((long long)total * (xcd + 1) / 8);
    return {lo + local * kAnWavesPerBlock + wid, hi, nloc * kAnWavesPerBlock};
}

template <typename R>
__global__ __launch_bounds__(256, sizeof(R) == 4 ? 4 : 2) void an_logmel_fwd_kernel(AnTables t, const float* __restrict__ x, int B, int T, int F,
                                                            const float* __restrict__ scale_p, float* __restrict__ feats) {
    __shared__ AnFrameLdsT<R, true> lds[kAnWavesPerBlock];
    __shared__ cx<R> tw1[kFftTw1], tw2[kFftTw2];
    an_stage_tw<R>(t, tw1, tw2);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float scale = scale_p ? *scale_p : 1.f;
    AnFrameLdsT<R, true>& L = lds[wid];
    __shared__ AnLaneTab<true> ltab;
    AnLaneConstT<R> lc;
    an_lane_init<R, true>(t, lane, ltab, lc);
    __syncthreads();
    const AnFrameRange fr = an_frame_range(B * F, wid);
    AnRaw cur;  // (a one-frame-ahead prefetch of these 32 registers cost the second wave per SIMD)
    for (int gf = fr.first; gf < fr.end; gf += fr.stride) {
        const int bb = gf / F;
        an_load_frame(x + (size_t)bb * T, T, gf - bb * F, lane, lc.m, cur);
        cx<R> z[8];
        an_frame_forward<R, true, true>(tw1, tw2, L, lc, cur, scale, lane, z);
        if (t.spec_cache) {  // packed spectrum for the backward of the same pass (float32 whatever R is), in TRANSFORM order
            float2* sc = t.spec_cache + (size_t)gf * kAnHalf;  // (fft512_transform_pos): straight from the registers
#pragma unroll
            for (int d = 0; d < 8; ++d) sc[lane + 64 * d] = make_float2((float)z[d].x, (float)z[d].y);
        }
        if (lane < kAnMel) {
            feats[(size_t)gf * kAnMel + lane] = 10.f * log10f(fmaxf(L.mel[lane], 1e-16f));
            if (t.mel_cache) t.mel_cache[(size_t)gf * kAnMel + lane] = L.mel[lane];
        }
        wave_sync();
    }
}

// filter membership of every bin (which mel filter pair it feeds, with which weights): LDS, shared by the block
struct AnBinLds {
    int m0[kAnBins + 3];
    float w0[kAnBins + 3], w1[kAnBins + 3];
};
__device__ __forceinline__ void an_stage_bins(const AnTables& t, AnBinLds& bl) {
    for (int i = threadIdx.x; i < kAnBins; i += blockDim.x) {
        bl.m0[i] = t.bin_m0[i];
        bl.w0[i] = t.bin_w0[i];
        bl.w1[i] = t.bin_w1[i];
    }
}

// One frame of the adjoint: from d loss / d log-mel (32 values of frame gf) to the 512 complex dz[n] = (d x[2n], d x[2n+1])
// of the frame's 1024 FFT inputs, n = lane + 64 j in dz[j]; the caller multiplies by the window.
// Entirely in transform order (round 5): a lane takes its four pairs Z[e_d], Z[512 - e_d] (an_partner_lane) -- SPEC: straight
// from the spectrum cache the forward kernel of the same pass left (any lane can read any address); else from a second
// forward transform of the waveform row xr, the partner's half by ds_bpermute --, forms d Z for both bins of every pair,
// hands the partner its half by ds_bpermute and runs the TRANSPOSED inverse transform (fft512T_transposed), which takes
// transform order in and leaves natural order in registers.  LDS exchanges per frame: the transform's 2; the natural-order
// form of rounds 3-4 (spectrum to LDS, pair loop in LDS, five exchanges of the inverse, LDS to the window multiply) made 8.
// CACHED: the mel energies come from t.mel_cache (else from the same second forward pass).
// what a frame's adjoint reads from the forward pass's caches (SPEC): the lane's pairs in transform order, lane 0's Z[256],
// and for lanes < 32 the mel energy and its gradient.  A separate step so that a kernel can issue the next frame's loads
// before it transforms the current one (an_logmel_bwd_kernel).
struct AnSpecIn {
    float2 zk[4], zr[4], z4;
    float mel, dfeat;
};
__device__ __forceinline__ void an_spec_load(const AnTables& t, const float* __restrict__ dfeats, size_t gf, int lane, AnSpecIn& in) {
    const float2* sc = t.spec_cache + gf * kAnHalf;
    const int pl = an_partner_lane(lane), z64 = lane == 0 ? 64 : 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        in.zk[d] = sc[lane + 64 * d];
        in.zr[d] = sc[(pl + 64 * (7 - d) + z64) & (kAnHalf - 1)];  // lane 0: its own element 8 - d (d = 0: Z[0] again)
    }
    in.z4 = sc[lane + 64 * 4];
    in.mel = in.dfeat = 0.f;
    if (lane < kAnMel) {
        in.mel = t.mel_cache[gf * kAnMel + lane];
        in.dfeat = dfeats[gf * kAnMel + lane];
    }
}

template <typename R, bool CACHED, bool SPEC, bool POWER>
__device__ __forceinline__ void an_frame_backward(const AnTables& t, const cx<R>* tw1, const cx<R>* tw2, const AnBinLds& bl,
                                                  AnFrameLdsT<R, POWER>& L, const AnLaneConstT<R>& lc, const float* __restrict__ xr, int T,
                                                  int f, size_t gf, float scale, const float* __restrict__ dfeats, int lane,
                                                  const AnSpecIn& pre, cx<R> (&dz)[8]) {
    static_assert(CACHED || !SPEC, "the spectrum cache comes with the mel cache");
    const int e0 = (lane >> 3) + 8 * (lane & 7), pl = an_partner_lane(lane), p4 = 4 * pl;
    const bool l0 = lane == 0;
    cx<R> zk[4], zr[4], z4;  // z4: lane 0's Z[256]
    if constexpr (SPEC) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            zk[d] = cmk<R>((R)pre.zk[d].x, (R)pre.zk[d].y);
            zr[d] = cmk<R>((R)pre.zr[d].x, (R)pre.zr[d].y);
        }
        z4 = cmk<R>((R)pre.z4.x, (R)pre.z4.y);
    } else {
        AnRaw cur;
        an_load_frame(xr, T, f, lane, lc.m, cur);
        cx<R> z[8];
        an_frame_forward<R, !CACHED, POWER>(tw1, tw2, L, lc, cur, scale, lane, z);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            zk[d] = z[d];
            zr[d] = an_from_lane<R>(p4, l0 ? z[(8 - d) & 7] : z[7 - d]);
        }
        z4 = z[4];
    }
    if (lane < 34) {
        float dm = 0.f;
        if (lane < kAnMel) {
            const float mel = SPEC ? pre.mel : (CACHED ? t.mel_cache[gf * kAnMel + lane] : L.mel[lane]);
            const float df = SPEC ? pre.dfeat : dfeats[gf * kAnMel + lane];
            // d/d mel of 10 log10(max(mel, 1e-16))
            dm = mel > 1e-16f ? df * (10.f / 2.302585092994046f) / mel : 0.f;
        }
        L.dmel[lane] = dm;
    }
    wave_sync();
    auto dpow = [&](int k) {  // 2 dP[k]: G[k] = 2 X[k] dP[k]
        const int m0 = bl.m0[k];
        return m0 >= 0 ? (R)2 * (R)fmaT(L.dmel[m0], bl.w0[k], L.dmel[m0 + 1] * bl.w1[k]) : (R)0;
    };
    cx<R> theirs[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int k = e0 + 64 * d;
        cx<R> xk, xr2;
        an_split2<R>(zk[d], zr[d], lc.c2[d], xk, xr2);
        const R dpk = dpow(k), dpr = dpow(kAnHalf - k);
        an_split2_adjoint<R>(cmk<R>(xk.x * dpk, xk.y * dpk), cmk<R>(xr2.x * dpr, xr2.y * dpr), lc.c2[d], dz[d], theirs[d]);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) dz[7 - d] = an_from_lane<R>(p4, theirs[d]);  // the partner's pair d is this lane's element 7 - d
    if (l0) {  // lane 0 is its own partner at 8 - d: its pair d belongs at element 8 - d; pair 0 folds bins 0 and 512 onto dz[0]
        const R dp = dpow(256);
        dz[0] = cmk<R>(dz[0].x + theirs[0].x, dz[0].y + theirs[0].y);
        dz[7] = theirs[1];
        dz[6] = theirs[2];
        dz[5] = theirs[3];
        dz[4] = cmk<R>(z4.x * dp, z4.y * dp);  // the self-paired bin: X[256] = conj Z[256], so d Z[256] = conj(G[256]) = 2 dP Z[256]
    }
    fft512T_transposed<R>(L.spec, tw1, tw2, lane, (R)1, dz);
}

// dfeats (B,F,32) -> dframes (B,F,800): gradient wrt the pre-emphasised, reflect-padded frame samples (the overlap-add is
// an_frames_to_wave_kernel's).  Instantiations by (R, CACHED, SPEC): the cached ones do not carry the mel-weight
// registers.
template <typename R, bool CACHED, bool SPEC>
__global__ __launch_bounds__(256, sizeof(R) == 4 && SPEC ? 4 : (sizeof(R) == 4 && CACHED ? 3 : 2)) void an_logmel_bwd_kernel(AnTables t, const float* __restrict__ x, int B, int T, int F,
                                                            const float* __restrict__ scale_p,
                                                            const float* __restrict__ dfeats, float* __restrict__ dframes) {
    constexpr bool POWER = !CACHED && !SPEC;
    __shared__ AnFrameLdsT<R, POWER> lds[kAnWavesPerBlock];
    __shared__ cx<R> tw1[kFftTw1], tw2[kFftTw2];
    __shared__ AnBinLds bl;
    an_stage_bins(t, bl);
    an_stage_tw<R>(t, tw1, tw2);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float scale = scale_p ? *scale_p : 1.f;
    AnFrameLdsT<R, POWER>& L = lds[wid];
    __shared__ AnLaneTab<POWER> ltab;
    AnLaneConstT<R> lc;
    an_lane_init<R, POWER>(t, lane, ltab, lc);
    __syncthreads();
    const AnFrameRange fr = an_frame_range(B * F, wid);
    AnSpecIn cur, nxt;  // SPEC: the next frame's cache lines are on their way while this one is transformed
    if (SPEC && fr.first < fr.end) an_spec_load(t, dfeats, (size_t)fr.first, lane, cur);
    for (int gf = fr.first; gf < fr.end; gf += fr.stride) {
        const int bb = gf / F;
        if (SPEC && gf + fr.stride < fr.end) an_spec_load(t, dfeats, (size_t)(gf + fr.stride), lane, nxt);
        cx<R> dzv[8];  // dz[n], n = lane + 64 i
        an_frame_backward<R, CACHED || SPEC, SPEC, POWER>(t, tw1, tw2, bl, L, lc, x + (size_t)bb * T, T, gf - bb * F, (size_t)gf, scale, dfeats,
                                                          lane, cur, dzv);
        if (SPEC) cur = nxt;
        float* out = dframes + (size_t)gf * kAnWin;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = lane + 64 * i;
            const int q = 2 * n - (kAnFft - kAnWin) / 2;  // window index of FFT input 2n (even, since (1024-800)/2 = 112)
            if (q >= 0 && q < kAnWin)
                *reinterpret_cast<float2*>(out + q) =
                    make_float2((float)dzv[i].x * lc.win[64 * (2 * i)], (float)dzv[i].y * lc.win[64 * (2 * i + 1)]);
        }
        wave_sync();
    }
}

// ---- overlap-add pieces shared by the three kernels that produce d loss / d waveform (same expressions, same bits)
// d x[t] from d pre[t - 1], d pre[t] (pre-emphasis adjoint: pre[p] = x[p + 1] - 0.97 x[p]) and the input scale
__device__ __forceinline__ float an_dx(float dpm1, float dp0, int t, int Lp, float scale) {
    float g = 0.f;
    if (t >= 1) g += dpm1;
    if (t <= Lp - 1) g = fmaT(-0.97f, dp0, g);
    return g * scale;
}
__device__ __forceinline__ void an_emit(float g, size_t o, float* __restrict__ grad_out, const float* __restrict__ x_in,
                                        float* __restrict__ x_out, const float* __restrict__ lower, const float* __restrict__ upper,
                                        float step, int grad_sign) {
    if (grad_out) grad_out[o] = g;
    if (x_out) {
        const float sg = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
        x_out[o] = fminf(fmaxf(x_in[o] + step * sg * (float)grad_sign, lower[o]), upper[o]);
    }
}
// sum over the frames that cover signal position p (ascending frames), frames in global memory
__device__ __forceinline__ float an_at_pos(const float* __restrict__ df, int F, int p) {
    const int pmin = -kAnWin / 2, pmax = (F - 1) * kAnHop + kAnWin / 2 - 1;
    float g = 0.f;
    if (p < pmin || p > pmax) return g;
    const int q = p + kAnWin / 2;  // >= 0
    int fhi = q / kAnHop;
    const int flo = q > kAnWin - 1 ? (q - (kAnWin - 1) + kAnHop - 1) / kAnHop : 0;
    if (fhi > F - 1) fhi = F - 1;
    for (int f = flo; f <= fhi; ++f) g += df[(size_t)f * kAnWin + (q - f * kAnHop)];
    return g;
}
// gradient wrt pre-emphasised sample s in [0, Lp): the position itself and what torch.stft's reflect padding maps onto it
__device__ __forceinline__ float an_dpre(const float* __restrict__ df, int F, int Lp, int s) {
    float g = an_at_pos(df, F, s);
    if (s >= 1) g += an_at_pos(df, F, -s);                    // left reflection  p = -s
    if (s <= Lp - 2) g += an_at_pos(df, F, 2 * (Lp - 1) - s);  // right reflection p = 2(L-1) - s
    return g;
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
    // every d pre value is gathered once per block (up to 5 frame reads) and shared through LDS: d x[t] needs
    // d pre[t-1] and d pre[t] -- the first version gathered both per thread and read the 491 MB of per-frame
    // gradients of a batch-512 step twice
    __shared__ float dp[257];  // dp[i] = d pre[t0 - 1 + i]
    const int t0 = blockIdx.x * 256;
    {
        const int sidx = t0 + (int)threadIdx.x;  // d pre[s], stored at dp[threadIdx.x + 1]
        dp[threadIdx.x + 1] = sidx <= Lp - 1 ? an_dpre(df, F, Lp, sidx) : 0.f;
        if (threadIdx.x == 0) dp[0] = t0 >= 1 ? an_dpre(df, F, Lp, t0 - 1) : 0.f;
    }
    __syncthreads();
    if (t >= T) return;
    const float g = an_dx(dp[threadIdx.x], dp[threadIdx.x + 1], t, Lp, scale);
    an_emit(g, (size_t)b * T + t, grad_out, x_io, x_io, lower, upper, step, grad_sign);
}

// Four signal positions per thread (T % 4 == 0).  The one-position kernel above ran at 2.1 TB/s on the 491 MB of a
// batch-512 step (236 us; 86 % of its wave cycles waiting, r05 PMC): a data-dependent loop of up to 5 dependent scalar
// loads per thread.  Positions q .. q + 3 with q % 4 == 0 never straddle a frame start (160 f), so all four are covered by
// the SAME frames q / 160 - 4 .. q / 160: five independent float4 loads, added in ascending frame order like an_at_pos
// (same bits).  A block whose positions touch what the reflect padding folds back (the first 401 and last ~400 samples)
// takes the one-position expressions instead.
constexpr int kAnF2wPerBlock = 1024;
__global__ __launch_bounds__(256) void an_frames_to_wave4_kernel(const float* __restrict__ dframes, int T, int F,
                                                                 const float* __restrict__ scale_p,
                                                                 float* __restrict__ grad_out, float* __restrict__ x_io,
                                                                 const float* __restrict__ lower,
                                                                 const float* __restrict__ upper, float step, int grad_sign) {
    const int b = blockIdx.y, tid = threadIdx.x;
    const float scale = scale_p ? *scale_p : 1.f;
    const float* df = dframes + (size_t)b * F * kAnWin;
    const int Lp = T - 1;
    __shared__ __attribute__((aligned(16))) float dp[4 + kAnF2wPerBlock];  // dp[3 + i] = d pre[t0 - 1 + i]
    const int t0 = blockIdx.x * kAnF2wPerBlock, s0 = t0 + 4 * tid;
    const int pmax = (F - 1) * kAnHop + kAnWin / 2 - 1;
    const int s_last = t0 + kAnF2wPerBlock - 1;
    const bool interior = t0 - 1 >= kAnWin / 2 + 1 && s_last <= Lp - 1 && 2 * (Lp - 1) - s_last > pmax;
    // the update's operands do not depend on the sums: requested first, one memory round trip instead of two
    const size_t o = (size_t)b * T + s0;
    float4 xi = make_float4(0.f, 0.f, 0.f, 0.f), lo = xi, up = xi;
    if (x_io && s0 < T) {
        xi = *reinterpret_cast<const float4*>(x_io + o);
        lo = *reinterpret_cast<const float4*>(lower + o);
        up = *reinterpret_cast<const float4*>(upper + o);
    }
    if (interior) {
        const int q = s0 + kAnWin / 2, k = q / kAnHop, r = q - k * kAnHop;
        float4 v[5];
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            const int f = k - 4 + d;  // >= 0 here: q >= 802
            v[d] = f <= F - 1 ? *reinterpret_cast<const float4*>(df + (size_t)f * kAnWin + (r + kAnHop * (4 - d)))
                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float pre = 0.f;
        if (tid == 0) pre = an_at_pos(df, F, t0 - 1) + 0.f + 0.f;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            if (k - 4 + d <= F - 1) {
                g.x += v[d].x;
                g.y += v[d].y;
                g.z += v[d].z;
                g.w += v[d].w;
            }
        }
        // an_dpre's two reflection terms are 0.f here; adding them keeps a -0.f sum what the one-position form makes of it
        g.x = g.x + 0.f + 0.f;
        g.y = g.y + 0.f + 0.f;
        g.z = g.z + 0.f + 0.f;
        g.w = g.w + 0.f + 0.f;
        *reinterpret_cast<float4*>(&dp[4 + 4 * tid]) = g;
        if (tid == 0) dp[3] = pre;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sidx = t0 + tid + 256 * j;
            dp[4 + tid + 256 * j] = sidx <= Lp - 1 ? an_dpre(df, F, Lp, sidx) : 0.f;
        }
        if (tid == 0) dp[3] = t0 >= 1 ? an_dpre(df, F, Lp, t0 - 1) : 0.f;
    }
    __syncthreads();
    if (s0 >= T) return;
    const float4 c = *reinterpret_cast<const float4*>(&dp[4 + 4 * tid]);
    const float pm1 = dp[3 + 4 * tid];
    float4 gx;
    gx.x = an_dx(pm1, c.x, s0, Lp, scale);
    gx.y = an_dx(c.x, c.y, s0 + 1, Lp, scale);
    gx.z = an_dx(c.y, c.z, s0 + 2, Lp, scale);
    gx.w = an_dx(c.z, c.w, s0 + 3, Lp, scale);
    if (grad_out) *reinterpret_cast<float4*>(grad_out + o) = gx;
    if (x_io) {
        const float fs = (float)grad_sign;
        auto upd = [&](float g, float x, float l, float u) {
            const float sg = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
            return fminf(fmaxf(x + step * sg * fs, l), u);
        };
        *reinterpret_cast<float4*>(x_io + o) =
            make_float4(upd(gx.x, xi.x, lo.x, up.x), upd(gx.y, xi.y, lo.y, up.y), upd(gx.z, xi.z, lo.z, up.z), upd(gx.w, xi.w, lo.w, up.w));
    }
}

// ---- round 5: the overlap-add INSIDE the adjoint.  The separate pair writes every frame's 800 gradient samples to HBM and
// reads them back (491 MB each way at 512 utterances; both kernels of the pair are HBM-bound there).  Here a WAVE owns a run
// of consecutive frames [fa, fb) of ONE utterance and walks it in ascending order, adding every frame's windowed gradient
// into a circular 800-position accumulator in LDS (position q = p + 400 at q mod 800; q is covered by frames
// (q - 799) / 160 .. q / 160, so the first frame to touch a position is the one that has it in its last hop: it stores,
// the next four add -- ascending frames, like an_at_pos: same sums in the same order, same bits as the separate pair,
// tests/test_gpu_audionet.py).  After frame f the 160 positions of its FIRST hop are complete: the wave turns them into
// d x (+ the fused sign / project / clamp update).  A run starts 5 frames before fa (its first d x needs d pre one position
// to the left of its first own position, which reaches 5 frames back); those halo frames are transformed twice -- the price
// of cutting an utterance: 512 utterances x 6 runs of 50 frames: +10 %; 64 utterances are too few frames for this form
// (runs of 6 frames: +80 %), sg_an_configure's automatic choice keeps the separate pair there.
// (The first version of this round kept a ring of whole frames per BLOCK, 38 KB: two blocks per CU, slower than the pair.)
// What reaches across the utterance's ends -- the reflect padding of torch.stft, the first 402 and the last ~400 samples --
// is left to an_edge_to_wave_kernel, for which the first 6 and last 5 frames also go to HBM.
// The update reads x_in and writes x_out, two DIFFERENT buffers: without the spectrum cache a neighbour run still reads the
// waveform around the cut (its halo frames) while this one writes its positions.
// NW waves per block.  float: 4 (48 KB: three blocks = 12 waves per CU, 143 registers; blocks of 8 waves -- 78 KB, two per CU,
// 16 waves at 128 registers with 8 spilled -- measured 1.333 against 1.312 ms per step at 512 utterances); double: 4 (69 KB, two).
template <typename R, bool SPEC, int NW>
__global__ __launch_bounds__(NW * 64, sizeof(R) == 4 ? 3 : 2) void an_logmel_bwd_ola_kernel(AnTables t, AnOlaArgs a) {
    __shared__ AnFrameLdsT<R, false> lds[NW];
    __shared__ cx<R> tw1[kFftTw1], tw2[kFftTw2];
    __shared__ AnBinLds bl;
    __shared__ AnLaneTab<false> ltab;
    __shared__ __attribute__((aligned(16))) float acc_all[NW][kAnWin];
    an_stage_bins(t, bl);
    an_stage_tw<R>(t, tw1, tw2);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float scale = a.scale_p ? *a.scale_p : 1.f;
    AnFrameLdsT<R, false>& L = lds[wid];
    AnLaneConstT<R> lc;
    an_lane_init<R, false>(t, lane, ltab, lc);
    __syncthreads();
    const int run = blockIdx.x * NW + wid;
    if (run >= a.B * a.S) return;  // (no block barrier below)
    const int b = run / a.S, r = run - b * a.S, F = a.F, T = a.T, Lp = T - 1;
    const int fa = (int)((long long)F * r / a.S), fb = (int)((long long)F * (r + 1) / a.S);
    const int fs = fa > 5 ? fa - 5 : 0;
    const float* xr = a.x + (size_t)b * T;
    float* acc = acc_all[wid];
    float carry = 0.f;  // d pre of the position before the hop being finished (wave-uniform)
    AnSpecIn cur, nxt;
    if (SPEC && fs < fb) an_spec_load(t, a.dfeats, (size_t)b * F + fs, lane, cur);
    int ph = kAnHop * (fs % 5);  // 160 f mod 800
    for (int f = fs; f < fb; ++f) {
        const size_t gf = (size_t)b * F + f;
        if (SPEC && f + 1 < fb) an_spec_load(t, a.dfeats, gf + 1, lane, nxt);
        // what the update of this frame's finished hop reads, requested before the transform (an HBM round trip is a fifth
        // of a frame's time here)
        float xi[3], lo[3], up[3];
        if (a.x_out && f >= fa) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int i = lane + 64 * j, tt = kAnHop * f + i - kAnWin / 2;
                xi[j] = lo[j] = up[j] = 0.f;
                if (i < kAnHop && tt >= a.t_lo && tt <= a.t_hi) {
                    const size_t o = (size_t)b * T + tt;
                    xi[j] = a.x_in[o];
                    lo[j] = a.lower[o];
                    up[j] = a.upper[o];
                }
            }
        }
        cx<R> dzv[8];  // dz[n], n = lane + 64 i (the same expressions as an_logmel_bwd_kernel: same bits)
        an_frame_backward<R, true, SPEC, false>(t, tw1, tw2, bl, L, lc, xr, T, f, gf, scale, a.dfeats, lane, cur, dzv);
        if (SPEC) cur = nxt;
        float* edge = (f >= fa && (f < a.edge_lo || f >= a.edge_hi)) ? a.dframes + gf * kAnWin : nullptr;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = lane + 64 * i;
            const int q = 2 * n - (kAnFft - kAnWin) / 2;  // window index (even): positions 160 f + q, + 1
            if (q >= 0 && q < kAnWin) {
                const float2 v = make_float2((float)dzv[i].x * lc.win[64 * (2 * i)], (float)dzv[i].y * lc.win[64 * (2 * i + 1)]);
                int p = ph + q;
                p = p >= kAnWin ? p - kAnWin : p;
                float2* slot = reinterpret_cast<float2*>(acc + p);
                float2 sum;
                if (f == fs || q >= kAnWin - kAnHop) {  // the first frame to reach the position: 0.f + v, like an_at_pos
                    sum = make_float2(0.f + v.x, 0.f + v.y);
                } else {
                    const float2 o = *slot;
                    sum = make_float2(o.x + v.x, o.y + v.y);
                }
                *slot = sum;
                if (edge) *reinterpret_cast<float2*>(edge + q) = v;
            }
        }
        wave_sync();
        if (f >= fa) {
            float last = 0.f;  // d pre of the previous 64 positions' last one
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int i = lane + 64 * j;  // position 160 f + i of the finished hop (i < 160)
                const float dp0 = i < kAnHop ? acc[ph + i] : 0.f;  // (ph + i < 800: ph <= 640)
                float dpm1 = __shfl_up(dp0, 1);
                if (lane == 0) dpm1 = j == 0 ? carry : last;
                last = __shfl(dp0, 63);
                const int tt = kAnHop * f + i - kAnWin / 2;
                if (i < kAnHop && tt >= a.t_lo && tt <= a.t_hi) {
                    const float g = an_dx(dpm1, dp0, tt, Lp, scale);
                    const size_t o = (size_t)b * T + tt;
                    if (a.grad_out) a.grad_out[o] = g;
                    if (a.x_out) {  // an_emit's expressions on the values requested above
                        const float sg = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
                        a.x_out[o] = fminf(fmaxf(xi[j] + a.step * sg * (float)a.grad_sign, lo[j]), up[j]);
                    }
                }
                if (j == 2) carry = __shfl(dp0, kAnHop - 1 - 128);
            }
        } else if (f == fa - 1) {
            carry = acc[ph + kAnHop - 1];  // d pre of position 160 fa - 1, complete after frame fa - 1
        }
        wave_sync();  // the next frame stores into the hop just read
        ph = ph + kAnHop >= kAnWin ? 0 : ph + kAnHop;
    }
}

// the utterance's ends: d x[t] for t < t_lo and t > t_hi from the edge frames the fused kernel left in HBM
__global__ __launch_bounds__(256) void an_edge_to_wave_kernel(AnOlaArgs a) {
    const int n_left = a.t_lo, n_right = a.T - 1 - a.t_hi;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_left + n_right) return;
    const int t = idx < n_left ? idx : a.t_hi + 1 + (idx - n_left);
    const int b = blockIdx.y, Lp = a.T - 1;
    const float scale = a.scale_p ? *a.scale_p : 1.f;
    const float* df = a.dframes + (size_t)b * a.F * kAnWin;
    const float dpm1 = t >= 1 ? an_dpre(df, a.F, Lp, t - 1) : 0.f;
    const float dp0 = t <= Lp - 1 ? an_dpre(df, a.F, Lp, t) : 0.f;
    const float g = an_dx(dpm1, dp0, t, Lp, scale);
    an_emit(g, (size_t)b * a.T + t, a.grad_out, a.x_in, a.x_out, a.lower, a.upper, a.step, a.grad_sign);
}
#pragma clang fp contract(fast)  // end of the log-mel front-end

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

// persistent blocks of the front-end kernels: a multiple of 8 (one share per XCD) once there is work for 16 blocks
static int an_front_blocks(int frames, int per_cu) {
    int want = (frames + kAnWavesPerBlock - 1) / kAnWavesPerBlock;
    if (want > per_cu * kAnCus) want = per_cu * kAnCus;
    if (want >= 16) want &= ~7;
    return want;
}
hipError_t launch_an_logmel_fwd(const AnTables& t, const float* x, int B, int T, int F, const float* scale, float* feats,
                                int fft32, hipStream_t s) {
    const dim3 grid(an_front_blocks(B * F, fft32 ? 4 : 2));  // = the kernels' blocks per CU (__launch_bounds__)
    if (fft32) hipLaunchKernelGGL(an_logmel_fwd_kernel<float>, grid, dim3(256), 0, s, t, x, B, T, F, scale, feats);
    else hipLaunchKernelGGL(an_logmel_fwd_kernel<double>, grid, dim3(256), 0, s, t, x, B, T, F, scale, feats);
    return hipGetLastError();
}
hipError_t launch_an_logmel_bwd(const AnTables& t, const float* x, int B, int T, int F, const float* scale,
                                const float* dfeats, float* dframes, int fft32, hipStream_t s) {
    const int kind = (t.mel_cache ? 1 : 0) + (t.mel_cache && t.spec_cache ? 2 : 0) + (fft32 ? 4 : 0);
    const dim3 grid(an_front_blocks(B * F, kind == 7 ? 4 : (kind == 5 ? 3 : 2)));
#define AN_BWD(R, C, SP) hipLaunchKernelGGL((an_logmel_bwd_kernel<R, C, SP>), grid, dim3(256), 0, s, t, x, B, T, F, scale, dfeats, dframes)
    switch (kind) {
        case 0: AN_BWD(double, false, false); break;
        case 1: AN_BWD(double, true, false); break;
        case 3: AN_BWD(double, true, true); break;
        case 4: AN_BWD(float, false, false); break;
        case 5: AN_BWD(float, true, false); break;
        default: AN_BWD(float, true, true); break;
    }
#undef AN_BWD
    return hipGetLastError();
}

// where the fused overlap-add hands over to the edge kernel (see an_logmel_bwd_ola_kernel)
static void an_ola_ranges(int T, int F, AnOlaArgs& a) {
    const int Lp = T - 1, pmax = (F - 1) * kAnHop + kAnWin / 2 - 1;
    a.t_lo = kAnWin / 2 + 2;  // d pre[t - 1] with t - 1 > 400: no left reflection lands on it
    int s_r0 = 2 * (Lp - 1) - pmax;  // first s with a right reflection term
    if (s_r0 > F * kAnHop - kAnWin / 2) s_r0 = F * kAnHop - kAnWin / 2;  // positions q = s + 400 < 160 F: inside the block loop
    a.t_hi = s_r0 - 1;
    if (a.t_hi > Lp - 1) a.t_hi = Lp - 1;
    if (a.t_hi < a.t_lo) {  // short utterance: everything is edge
        a.t_lo = T;
        a.t_hi = T - 1;
        a.edge_lo = F;
        a.edge_hi = F;
        return;
    }
    a.edge_lo = (a.t_lo - 1 + kAnWin / 2) / kAnHop + 1;  // frames covering q <= t_lo - 1 + 400
    const int q = a.t_hi + kAnWin / 2;
    a.edge_hi = q > kAnWin - 1 ? (q - (kAnWin - 1) + kAnHop - 1) / kAnHop : 0;
}

// runs per utterance of the fused overlap-add: rounds of resident waves x frames per run (5 halo frames each)
static int an_ola_slices(int B, int F, int slots, long* cost_out = nullptr) {
    int best_s = 1;
    long best_cost = -1;
    const int smax = F / 8 > 0 ? F / 8 : 1;
    for (int sl = 1; sl <= smax; ++sl) {
        const long rounds = ((long)B * sl + slots - 1) / slots;
        const long cost = rounds * ((F + sl - 1) / sl + (sl > 1 ? 5 : 0));
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best_s = sl;
        }
    }
    if (cost_out) *cost_out = best_cost;
    return best_s;
}
static int an_ola_slots(int fft32, int num_cus) { return (num_cus > 0 ? num_cus : kAnCus) * (fft32 ? 12 : 8); }  // resident waves
// Is the fused form the faster one?  Its halo frames are transformed twice; the separate pair moves 2 x 3.2 KB per frame
// through HBM instead.  Measured (profiles/r05_an_frontend_ab.txt): the fused form wins once the halo costs < ~25 %.
bool an_ola_pays(int B, int F, int fft32, int num_cus) {
    const int slots = an_ola_slots(fft32, num_cus);
    long cost = 0;
    an_ola_slices(B, F, slots, &cost);
    return 4 * cost * slots <= 5 * (long)B * F;
}
hipError_t launch_an_logmel_bwd_ola(const AnTables& t, AnOlaArgs a, int fft32, int num_cus, hipStream_t s) {
    if (!t.mel_cache) return hipErrorInvalidValue;  // the fused form is for the pass whose forward has just run
    an_ola_ranges(a.T, a.F, a);
    const int slots = an_ola_slots(fft32, num_cus);  // resident waves
    a.S = an_ola_slices(a.B, a.F, slots);
    const bool spec = t.spec_cache != nullptr;
    hipError_t e = hipSuccess;
#define AN_OLA(R, SP, NW) hipLaunchKernelGGL((an_logmel_bwd_ola_kernel<R, SP, NW>), dim3((a.B * a.S + NW - 1) / NW), dim3(NW * 64), 0, s, t, a)
    if (fft32) {
        if (spec) AN_OLA(float, true, 4);
        else AN_OLA(float, false, 4);
    } else {
        if (spec) AN_OLA(double, true, 4);
        else AN_OLA(double, false, 4);
    }
#undef AN_OLA
    if (e != hipSuccess) return e;
    if ((e = hipGetLastError()) != hipSuccess) return e;
    const int n_edge = a.t_lo + (a.T - 1 - a.t_hi);
    if (n_edge > 0) hipLaunchKernelGGL(an_edge_to_wave_kernel, dim3((n_edge + 255) / 256, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_an_frames_to_wave(const float* dframes, int B, int T, int F, const float* scale, float* grad_out,
                                    float* x_io, const float* lower, const float* upper, float step, int grad_sign,
                                    hipStream_t s) {
    const auto al16 = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (T % 4 == 0 && al16(dframes) && al16(grad_out) && al16(x_io) && al16(lower) && al16(upper))
        hipLaunchKernelGGL(an_frames_to_wave4_kernel, dim3((T + kAnF2wPerBlock - 1) / kAnF2wPerBlock, B), dim3(256), 0, s, dframes, T,
                           F, scale, grad_out, x_io, lower, upper, step, grad_sign);
    else
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
    static const bool tr_on = sg_tune_env("SG_AN_TAIL_TRACE") != nullptr;
    static PerDeviceScratch tr_buf;
    unsigned long long* tr_dev = tr_on ? static_cast<unsigned long long*>(tr_buf.get(8 * 8)) : nullptr;
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
