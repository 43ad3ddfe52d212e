// Kaldi MFCC forward and hand-coded backward (waveform <-> 30 cepstra per 10 ms frame).
//
// Replaces torchaudio.compliance.kaldi.mfcc as called at reference model/xv_plda.py:114-148 and
// the autograd graph behind it.  Stages (torchaudio kaldi.py v0.6.0 _get_window / fbank / mfcc):
//   strided frames with reflected edges (snip_edges=False) -> optional dither -> DC removal ->
//   raw log-energy -> pre-emphasis 0.97 -> povey window -> zero-pad to 512 -> |rFFT|^2 ->
//   30 triangular mel bins (20..7600 Hz) -> log -> DCT-II (ortho, 30 ceps) -> lifter 22 ->
//   c0 <- log-energy.
//
// One wave (64 lanes) owns one frame: the 400 samples are read coalesced from the waveform, the
// 512-point FFT runs in LDS + registers (three radix-8 passes, table twiddles), reductions stay inside the wave (DPP).
// Transform precision (round 6): float32 by default -- the reference's own: torchaudio 0.6's kaldi.mfcc is float32 end to
// end, torch.rfft included -- with float64 transforms as the counterpart (sg_xv_configure).  Rounds 1-5 ran float64 only,
// after a first float32 transform (radix-2, run-time twiddles) had shown 0.8 % gradient sign mismatches against the oracle.
// With the radix-8 network and twiddles rounded once from float64 the float32 transform is as close to the oracle as the
// float64 one (profiles/r06_mfcc_precision.txt: d loss / d wav sign mismatches 1.5e-4 vs 1.3e-4; the float32 oracle itself
// differs from a float64 evaluation of the same model in 3.3e-4 of the samples), and 8-byte LDS elements + 78-100 VGPRs put
// four waves on a SIMD where the float64 form has two.
// The backward kernel reads the forward's spectrum and mel energies back (or recomputes the forward of its frame), then
// walks the stages in reverse; the gradient spectrum is formed in registers and transformed back by the TRANSPOSED network.
// It writes per-frame sample gradients (B,F,400); frames_to_wave_kernel does the deterministic overlap-add.
#include "sg_internal.h"
#include "fft512t.h"

namespace sg {

constexpr int kMfccMaxBlocks = 512;  // 2 blocks (~60 KB LDS each) per CU x 256 CUs

// Round 6: the per-frame functions are templates over the transform's scalar type R.
//   R = float  (default): the reference's own precision -- torchaudio 0.6's kaldi.mfcc is float32 end to end, torch.rfft
//              included (model/xv_plda.py:114-148) -- on the transform-order / transposed-network pair of fft512t.h: the
//              forward transform leaves bin k1 + 8 c + 64 d in registers of lane (k1, c), power, spectrum cache and the
//              gradient spectrum are formed right there, and the TRANSPOSED inverse takes that layout and returns
//              samples lane + 64 j in registers, where window, pre-emphasis adjoint (a whole-wave DPP shift), energy and DC
//              terms are applied: two LDS exchanges per transform and none around it (rounds 1-5: five, plus the spectrum
//              and sample round trips).  8-byte LDS elements, 8 waves per block, 4 waves per SIMD.
//   R = double (sg_xv_configure(ctx, 64)): the same code on float64 transforms, the form of rounds 1-5 kept as the
//              counterpart (a windowed speech frame has > 80 dB between its strongest harmonic and the weak bins; a float32
//              transform leaves a round-off floor of ~1e-7 of the strongest bin on every bin, float64 does not).
template <typename R> struct MfccCfg;
#ifndef SG_MFCC_F32_WAVES  // (tools/r06_mfcc_variants.sh builds the alternatives)
#define SG_MFCC_F32_WAVES 8
#define SG_MFCC_F32_OCC 4
#endif
template <> struct MfccCfg<float> { static constexpr int kWaves = SG_MFCC_F32_WAVES, kMinBlocks = SG_MFCC_F32_OCC; };  // (HIP: min waves per SIMD) 16 waves per CU: <= 128 VGPRs
template <> struct MfccCfg<double> { static constexpr int kWaves = 4, kMinBlocks = 2; };

template <typename R>
struct FrameLdsT {
    cx<R> spec[kFft + kFft / 8];  // the transform's exchange buffer (element i at SP(i))
    float power[256];
    float mel[32];
    float lmel[32];
    float tmp[32];
};

// Every LDS buffer in this file is private to one wave, and a wave's DS instructions execute in
// program order, so cross-lane hand-offs through LDS need no s_barrier -- only a fence that stops
// the compiler from moving LDS accesses across it (a block-wide __syncthreads() here coupled the
// independent waves at ~25 points per frame).
// Constant tables staged once per block into LDS: every per-frame table access was a dependent global
// load (L1/L2 hit, but ~0.3 us of latency each with few waves per SIMD to hide it).
// Round 6: EVERY per-lane constant lives in the block's LDS table image (MfccLdsImage, sg_internal.h: the mel weights of a
// lane's half filter, the DCT matrix in both orientations, window, twiddles): in registers they cost 55-60 VGPRs per lane and
// held the kernels at 2 waves per SIMD -- latency-bound at ~40 % VALU and ~37 % LDS utilisation.  All of these reads are
// lane-linear (conflict-free) and independent of the frame's data.
static_assert(kMfccTw1 == kFftTw1 && kMfccTw2 == kFftTw2, "table sizes of fft512.h");
template <typename R>
struct TabLdsT : MfccLdsImage<R> {
    __device__ __forceinline__ const cx<R>* tw1c() const { return reinterpret_cast<const cx<R>*>(this->tw1); }
    __device__ __forceinline__ const cx<R>* tw2c() const { return reinterpret_cast<const cx<R>*>(this->tw2); }
};
template <typename R> __device__ __forceinline__ const MfccLdsImage<R>* lds_image(const MfccTables& t);
template <> __device__ __forceinline__ const MfccLdsImage<float>* lds_image<float>(const MfccTables& t) { return t.lds_f32; }
template <> __device__ __forceinline__ const MfccLdsImage<double>* lds_image<double>(const MfccTables& t) { return t.lds_f64; }

template <typename R>
__device__ __forceinline__ void stage_tables(const MfccTables& t, TabLdsT<R>& tb) {
    const uint4* src = reinterpret_cast<const uint4*>(lds_image<R>(t));
    uint4* dst = reinterpret_cast<uint4*>(&tb);
    for (int i = threadIdx.x; i < (int)(sizeof(MfccLdsImage<R>) / 16); i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}


// Sum over the 64 lanes, the same value in every lane: a butterfly inside each row of 16 lanes on DPP (quad permutes, then
// the half-row and row mirrors pair the quads / halves), then the four row sums through scalar registers.  ~12 instructions
// and no LDS traffic; the ds_bpermute butterfly of rounds 1-5 cost six dependent LDS round trips per sum.
__device__ __forceinline__ float wave_sum(float v) {
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));  // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

// Philox4x32-10 counter-based generator (Salmon et al. 2011); one 32-bit draw per (key, counter).
__device__ __forceinline__ uint32_t philox_u32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
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
    return c0;
}

__device__ __forceinline__ float dither_draw(uint64_t seed, int64_t utt, int frame, int n, float dither) {
    const uint32_t r = philox_u32(seed, (uint32_t)n, (uint32_t)frame, (uint32_t)utt, (uint32_t)((uint64_t)utt >> 32));
    float u = ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f);
    u = fmaxf(u, kEps);
    // torchaudio 0.6.0 _get_window feeds the SAME uniform draw to both Box-Muller factors
    return sqrtf(-2.f * logf(u)) * cosf(6.283185307179586f * u) * dither;
}

struct FrameState {
    float s[7];      // DC-removed samples n = lane + 64 i
    float energy;    // sum of squares (before log)
};

// Raw samples of global frame gf (reflect-padded, snip_edges=False), sample n = lane + 64 i -> raw[i].
// Issued one frame ahead of use so the ~1 us global latency overlaps the previous frame's FFT.
// rep_utts > 0: the rows are EOT repeats of rep_utts utterances (row = repeat * rep_utts + utterance): every repeat reads
// the same waveform row.
__device__ __forceinline__ void load_frame(const float* __restrict__ x, int T, int F, int gf, int total, int lane,
                                           float (&raw)[7], int rep_utts) {
    const int g = gf < total ? gf : total - 1;
    const int row = g / F, f = g - row * F;
    const int b = rep_utts > 0 ? row % rep_utts : row;
    const int base = f * kShift - (kWin / 2 - kShift / 2);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int n = lane + 64 * i;
        const int p = base + (n < kWin ? n : kWin - 1);
        const int idx = p < 0 ? -p - 1 : (p >= T ? 2 * T - 1 - p : p);
        raw[i] = x[(size_t)b * T + idx];
    }
}

// An offset the compiler cannot see through (always 0): table reads addressed with it are re-issued per frame instead of
// being hoisted out of the frame loop into 30-60 registers per lane.
__device__ __forceinline__ int opaque_zero() {
    int z = 0;
    asm volatile("" : "+v"(z));
    return z;
}

// No implicit contraction in the per-frame arithmetic (as in fft512t.h): the same functions are instantiated in the forward
// kernel and in both backward kernels and must give the same bits there; every fused multiply-add is written out.
#pragma clang fp contract(off)

// MODE 0: full forward (writes the spectrum / mel cache when t.spec_cache is set); MODE 1: backward with a cache:
// only the sample statistics are recomputed, spectrum and mel energies are read back (no FFT, no mel/DCT loops).
// Xk[d] = the spectrum at bin (lane >> 3) + 8 (lane & 7) + 64 d ("transform order", fft512t.h), d < 4: the 256 bins that
// exist, in registers.  The transform's input stays out of LDS as well: the windowed frame is real and zero beyond sample
// 399, so pass 1 takes it as 7 values per lane.  The spectrum cache keeps a frame's bins in transform order (element
// 64 d + lane): written and read back as whole 512-byte rows.
// DZ: 0 = no dither, 1 = the kernel's own Philox draws, 2 = an explicit noise tensor (the hot loop carries only its own case)
template <typename R, int MODE, int DZ>
__device__ __forceinline__ void frame_forward(const MfccTables& t, const TabLdsT<R>& tb, int tz, FrameLdsT<R>& L,
                                              const float (&raw)[7], int F, int b, int f, float scale,
                                              const sg_dither& dz, int lane, FrameState& st, float& cep_out, cx<R> (&Xk)[4]) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int n = lane + 64 * i;
        float v = 0.f;
        if (n < kWin) {
            v = raw[i] * scale;
            if (DZ == 2) v += dz.noise_dev[((size_t)b * F + f) * kWin + n];
            else if (DZ == 1) {
                // repeat r of the batched EOT passes draws from key seed + r * 0xC2B2AE3D27D4EB4F, like the r-th of the
                // sequential passes it replaces (sg_xv_pgd_run); a caller that materialised the repeats itself
                // (EOT.py:29) names their length in dz.rep_rows, and a row slice of a larger call its offset in
                // dz.row_base (speakerguard_hip.h, sg_dither): same draws however the work was cut
                const int64_t g = dz.row_base + b;
                const int64_t rlen = t.rep_utts > 0 ? t.rep_utts : dz.rep_rows;
                const int64_t rep = rlen > 0 ? g / rlen : 0, utt = g - rep * rlen;
                v += dither_draw(dz.seed + (uint64_t)rep * 0xC2B2AE3D27D4EB4Full, dz.index_base + utt, f, n, dz.dither);
            }
        }
        st.s[i] = v;
        sum += v;
    }
    const float mean = wave_sum(sum) / (float)kWin;
    float e = 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int n = lane + 64 * i;
        if (n < kWin) {
            st.s[i] -= mean;
            e = __builtin_fmaf(st.s[i], st.s[i], e);
        }
    }
    st.energy = wave_sum(e);
    const size_t gfi = (size_t)b * F + f;
    if (MODE == 1) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const float2 c = t.spec_cache[gfi * 256 + 64 * d + lane];
            Xk[d] = cmk<R>((R)c.x, (R)c.y);
        }
        if (lane < 32) L.mel[lane] = t.mel_cache[gfi * 32 + lane];
        wave_sync();
        cep_out = 0.f;
        return;
    }
    // pre-emphasis (replicate pad on the left), povey window -- in registers: sample n - 1 of n = lane + 64 i is the left
    // neighbour lane's s[i] (a whole-wave DPP shift), for lane 0 lane 63's s[i - 1] (sample 0: itself); and the windowed
    // values ARE pass 1's inputs x[lane + 64 j] of this lane: no LDS round trip between the samples and the transform
    cx<R> in[8], out[8];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const float edge = i > 0 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, st.s[i - 1]), 63)) : st.s[0];
        const float prev = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, st.s[i]),
                                                                                  0x138 /* wave_shr:1 */, 0xf, 0xf, false));
        in[i] = cmk<R>((R)((st.s[i] - 0.97f * prev) * tb.window[tz + lane + 64 * i]), (R)0);  // (window = 0 beyond sample 399)
    }
    in[7] = cmk<R>((R)0, (R)0);
    fft512T_regs<R>(L.spec, tb.tw1c() + tz, tb.tw2c() + tz, lane, (R)-1, in, out);
    {
        const int kb = (lane >> 3) + 8 * (lane & 7);  // this lane's bins: kb + 64 d
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const cx<R> c = out[d];
            Xk[d] = c;
            L.power[kb + 64 * d] = (float)fmaT(c.x, c.x, c.y * c.y);
            if (t.spec_cache) t.spec_cache[gfi * 256 + 64 * d + lane] = make_float2((float)c.x, (float)c.y);
        }
    }
    wave_sync();
    // 30 triangular mel filters, two lanes per filter (each sums half of the filter's bins); the weight
    // of bin k in filter m is bin_w0[k] if m is the lower of the two filters covering k, else bin_w1[k]
    {
        const int m = lane >> 1, h = lane & 1;
        const int k0 = tb.mel_k0[tz + lane];
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < kMelLaneBins; ++j) acc = __builtin_fmaf(L.power[min(k0 + j, 255)], tb.melw_lane[tz + j * 64 + lane], acc);
        acc += __shfl_xor(acc, 1, 64);
        if (m < kMel && h == 0) {
            L.mel[m] = acc;
            L.lmel[m] = logf(fmaxf(acc, kEps));
            if (t.mel_cache) t.mel_cache[gfi * 32 + m] = acc;
        }
    }
    wave_sync();
    cep_out = 0.f;
    if (lane < kCep) {
        float v = 0.f;
#pragma unroll
        for (int m = 0; m < kMel; ++m) v = __builtin_fmaf(L.lmel[m], tb.dct[tz + m * kCep + lane], v);
        v *= tb.lifter[lane];
        if (lane == 0) v = logf(fmaxf(st.energy, kEps));
        cep_out = v;
    }
}

template <typename R, int DZ>
__global__ __launch_bounds__(MfccCfg<R>::kWaves * 64, MfccCfg<R>::kMinBlocks) void mfcc_fwd_kernel(MfccTables t, const float* __restrict__ x, int B, int T, int F,
                                                                          const float* __restrict__ scale_p, sg_dither dz,
                                                                          float* __restrict__ feats) {
    constexpr int kWaves = MfccCfg<R>::kWaves;
    __shared__ FrameLdsT<R> lds[kWaves];
    __shared__ TabLdsT<R> tb;
    stage_tables<R>(t, tb);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float scale = scale_p ? *scale_p : 1.f;
    FrameLdsT<R>& L = lds[wid];
    // frames are dealt round-robin to the resident waves (grid sized to the chip, tables staged once)
    const int total = B * F, stride = gridDim.x * kWaves;
    float raw[7], nxt[7];
    load_frame(x, T, F, blockIdx.x * kWaves + wid, total, lane, nxt, t.rep_utts);
    for (int gf = blockIdx.x * kWaves + wid; gf < total; gf += stride) {
        const int b = gf / F, f = gf - b * F;
        const int tz = opaque_zero();
#pragma unroll
        for (int i = 0; i < 7; ++i) raw[i] = nxt[i];
        load_frame(x, T, F, gf + stride, total, lane, nxt, t.rep_utts);
        FrameState st;
        float cep;
        cx<R> Xk[4];
        frame_forward<R, 0, DZ>(t, tb, tz, L, raw, F, b, f, scale, dz, lane, st, cep, Xk);
        if (lane < kCep) feats[((size_t)b * F + f) * kCep + lane] = cep;
        wave_sync();
    }
}

// dfeats: (B,F,ld) gradient wrt the 30 cepstra; dframes: (B,F,400) gradient wrt the strided frames,
// already multiplied by `scale` (the int16 rescale of check_input_range, model/utils.py:14).
// CACHED: the forward kernel of the same pass left spectrum + mel energies in t.spec_cache / t.mel_cache (the attack
// loop); otherwise the forward is recomputed here (standalone sg_xv_mfcc_backward).  Two instantiations so that the
// cached one does not carry the forward's lane constants.
template <typename R, bool CACHED, int DZ>
__global__ __launch_bounds__(MfccCfg<R>::kWaves * 64, CACHED ? MfccCfg<R>::kMinBlocks : 1) void mfcc_bwd_kernel(MfccTables t, const float* __restrict__ x, int B, int T, int F,
                                                                          const float* __restrict__ scale_p, sg_dither dz,
                                                                          const float* __restrict__ dfeats, int ld,
                                                                          float* __restrict__ dframes) {
    constexpr int kWaves = MfccCfg<R>::kWaves;
    __shared__ FrameLdsT<R> lds[kWaves];
    __shared__ TabLdsT<R> tb;
    stage_tables<R>(t, tb);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float scale = scale_p ? *scale_p : 1.f;
    FrameLdsT<R>& L = lds[wid];
    // backward-only lane constants: the mel membership of this lane's four bins (transform order: bin kb + 64 d)
    const int kb = (lane >> 3) + 8 * (lane & 7);
    int bin_m0[4];
    float bin_w0[4], bin_w1[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        bin_m0[d] = t.bin_m0[kb + 64 * d];
        bin_w0[d] = t.bin_w0[kb + 64 * d];
        bin_w1[d] = t.bin_w1[kb + 64 * d];
    }
    // frames are dealt round-robin to the resident waves (grid sized to the chip, tables staged once)
    const int total = B * F, stride = gridDim.x * kWaves;
    float raw[7], nxt[7];
    load_frame(x, T, F, blockIdx.x * kWaves + wid, total, lane, nxt, t.rep_utts);
    for (int gf = blockIdx.x * kWaves + wid; gf < total; gf += stride) {
        const int b = gf / F, f = gf - b * F;
        const int tz = opaque_zero();
#pragma unroll
        for (int i = 0; i < 7; ++i) raw[i] = nxt[i];
        load_frame(x, T, F, gf + stride, total, lane, nxt, t.rep_utts);
        FrameState st;
        float cep;
        cx<R> Xk[4];
        frame_forward<R, CACHED ? 1 : 0, DZ>(t, tb, tz, L, raw, F, b, f, scale, dz, lane, st, cep, Xk);
        // ---- cepstra -> log-mel
        float dc = 0.f;
        if (lane < kCep) dc = dfeats[((size_t)b * F + f) * ld + lane];
        const float denergy = __shfl(dc, 0, 64);
        if (lane < 32) L.tmp[lane] = (lane == 0 || lane >= kCep) ? 0.f : dc * tb.lifter[lane];
        wave_sync();
        if (lane < 32) {
            float dm = 0.f;
            if (lane < kMel) {
                float dl = 0.f;
#pragma unroll
                for (int c = 0; c < kCep; ++c) dl = __builtin_fmaf(L.tmp[c], tb.dct_t[tz + c * 32 + lane], dl);
                const float mel = L.mel[lane];
                dm = mel > kEps ? dl / mel : 0.f;
            }
            L.lmel[lane] = dm;  // d loss / d mel energy (entries 30,31 = 0)
        }
        wave_sync();
        // ---- mel -> power -> spectrum gradient G[k] = 2 X[k] dP[k] at this lane's bins, in transform order -- the layout the
        //      TRANSPOSED inverse network takes (d >= 4: bins 256..511, zero); it returns samples lane + 64 j in registers.
        cx<R> v[8];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            v[d] = cmk<R>((R)0, (R)0);
            const int m0 = bin_m0[d];
            if (m0 >= 0) {
                const R dp = (R)(2.f * __builtin_fmaf(L.lmel[m0], bin_w0[d], L.lmel[m0 + 1] * bin_w1[d]));
                v[d] = cmk<R>(Xk[d].x * dp, Xk[d].y * dp);
            }
        }
#pragma unroll
        for (int d = 4; d < 8; ++d) v[d] = cmk<R>((R)0, (R)0);
        fft512T_transposed<R>(L.spec, tb.tw1c() + tz, tb.tw2c() + tz, lane, (R)1, v);
        // ---- window, pre-emphasis (sample n + 1 is the right neighbour lane's, lane 63: lane 0's next register), energy,
        //      DC removal
        float sw[8];
#pragma unroll
        for (int i = 0; i < 7; ++i) sw[i] = (float)v[i].x * tb.window[tz + lane + 64 * i];
        sw[7] = 0.f;
        float ds[7];
        float sum = 0.f;
        const float einv = st.energy > kEps ? 2.f * denergy / st.energy : 0.f;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int n = lane + 64 * i;
            const float edge = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sw[i + 1]), 0));
            const float next = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, sw[i]),
                                                                                      0x130 /* wave_shl:1 */, 0xf, 0xf, false));
            float g = 0.f;
            if (n < kWin) {
                g = sw[i] - 0.97f * next;  // (sample 400 does not exist: its sw is 0)
                if (n == 0) g -= 0.97f * sw[0];
                g = __builtin_fmaf(einv, st.s[i], g);
            }
            ds[i] = g;
            sum += g;
        }
        const float mean = wave_sum(sum) / (float)kWin;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int n = lane + 64 * i;
            if (n < kWin) dframes[((size_t)b * F + f) * kWin + n] = (ds[i] - mean) * scale;
        }
        wave_sync();
    }
}
#pragma clang fp contract(fast)

// Deterministic overlap-add: sample n of utterance b gathers, in a fixed order, every frame
// position that maps to it -- directly, or through the reflected edges of _get_strided.
// Optionally fuses the PGD step (attack/FGSM.py:65,68) so the gradient never round-trips HBM.
template <int V>
__global__ __launch_bounds__(256) void frames_to_wave_kernel(const float* __restrict__ dframes, int B, int T, int F, int R,
                                                             const float* acc_in, float* grad_out, float* __restrict__ x_io,
                                                             const float* __restrict__ lower,
                                                             const float* __restrict__ upper, float step, int grad_sign) {
    // acc_in (may alias grad_out): gradient accumulated over the earlier EOT repeats of this step (EOT.py:41-47)
    // V = 4: a thread owns four consecutive samples n0 .. n0 + 3 (n0 a multiple of 4; host: T % 4 == 0).  Away from the
    // reflected edges the four share their frames (frame shift and window are multiples of 4), so every gather is one aligned
    // 16-byte load and the per-sample sums keep their order; a thread whose samples touch an edge takes them one by one.
    const int n0 = (blockIdx.x * 256 + threadIdx.x) * V;
    const int b = blockIdx.y;
    if (n0 >= T) return;
    constexpr int kPad = kWin / 2 - kShift / 2;
    const int last_q = (F - 1) * kShift - kPad + kWin - 1;  // the largest original position a frame covers
    const size_t o = (size_t)b * T + n0;
    float total[V];
    const bool interior = V == 4 && n0 >= kPad && 2 * T - 1 - (n0 + 3) > last_q && n0 + 3 < T;
    // dframes (R, B, F, 400): R batched EOT repeats of the B utterances; their gradients are summed in repeat order,
    // exactly as R sequential passes handing the sum on through acc_in did (EOT.py:41-47)
    if (interior) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int q = n0 + kPad;
        int fhi = q / kShift;
        const int flo = q > kWin - 1 ? (q - (kWin - 1) + kShift - 1) / kShift : 0;
        if (fhi > F - 1) fhi = F - 1;
        f4 tot = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r) {
            const float* df = dframes + ((size_t)r * B + b) * F * kWin;
            f4 g = {0.f, 0.f, 0.f, 0.f};
            for (int f = flo; f <= fhi; ++f) g += *reinterpret_cast<const f4*>(df + (size_t)f * kWin + (q - f * kShift));
            if (r == 0) tot = acc_in ? *reinterpret_cast<const f4*>(acc_in + o) + g : g;
            else tot = tot + g;
        }
        total[0] = tot.x;
        if (V == 4) { total[1] = tot.y; total[2] = tot.z; total[3] = tot.w; }
    } else {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int n = n0 + v;
            total[v] = 0.f;
            if (n >= T) continue;
            const int pr = 2 * T - 1 - n;
            for (int r = 0; r < R; ++r) {
                const float* df = dframes + ((size_t)r * B + b) * F * kWin;
                float g = 0.f;
                auto add_pos = [&](int p) {
                    const int q = p + kPad;  // position in the padded signal, >= 0
                    int fhi = q / kShift;
                    int flo = q > kWin - 1 ? (q - (kWin - 1) + kShift - 1) / kShift : 0;
                    if (fhi > F - 1) fhi = F - 1;
                    for (int f = flo; f <= fhi; ++f) g += df[(size_t)f * kWin + (q - f * kShift)];
                };
                add_pos(n);
                if (n < kPad) add_pos(-n - 1);
                if (pr <= last_q) add_pos(pr);
                if (r == 0) total[v] = acc_in ? acc_in[o + v] + g : g;
                else total[v] = total[v] + g;
            }
        }
    }
    if (V == 4 && n0 + 3 < T) {  // (T % 4 == 0: always, for V = 4) the update as 16-byte accesses
        typedef float f4 __attribute__((ext_vector_type(4)));
        if (grad_out) *reinterpret_cast<f4*>(grad_out + o) = f4{total[0], total[V > 1 ? 1 : 0], total[V > 2 ? 2 : 0], total[V > 3 ? 3 : 0]};
        if (x_io) {
            const f4 xv = *reinterpret_cast<const f4*>(x_io + o), lo = *reinterpret_cast<const f4*>(lower + o),
                     hi = *reinterpret_cast<const f4*>(upper + o);
            f4 res;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float g = total[v < V ? v : 0];
                const float sg = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
                res[v] = fminf(fmaxf(xv[v] + step * sg * (float)grad_sign, lo[v]), hi[v]);
            }
            *reinterpret_cast<f4*>(x_io + o) = res;
        }
        return;
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        if (n0 + v >= T) break;
        const float g = total[v];
        if (grad_out) grad_out[o + v] = g;
        if (x_io) {
            const float sg = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
            float val = x_io[o + v] + step * sg * (float)grad_sign;
            val = fminf(fmaxf(val, lower[o + v]), upper[o + v]);
            x_io[o + v] = val;
        }
    }
}

template <typename R>
static void mfcc_fwd_launch(const MfccTables& t, const float* x, int B, int T, int F, const float* scale, const sg_dither& d,
                            float* feats, hipStream_t s) {
    constexpr int kWaves = MfccCfg<R>::kWaves;
    const int want = (B * F + kWaves - 1) / kWaves;
    dim3 grid(want < kMfccMaxBlocks ? want : kMfccMaxBlocks);
    if (d.noise_dev) hipLaunchKernelGGL((mfcc_fwd_kernel<R, 2>), grid, dim3(kWaves * 64), 0, s, t, x, B, T, F, scale, d, feats);
    else if (d.dither != 0.f) hipLaunchKernelGGL((mfcc_fwd_kernel<R, 1>), grid, dim3(kWaves * 64), 0, s, t, x, B, T, F, scale, d, feats);
    else hipLaunchKernelGGL((mfcc_fwd_kernel<R, 0>), grid, dim3(kWaves * 64), 0, s, t, x, B, T, F, scale, d, feats);
}

template <typename R>
static void mfcc_bwd_launch(const MfccTables& t, const float* x, int B, int T, int F, const float* scale, const sg_dither& d,
                            const float* dfeats, float* dframes, hipStream_t s) {
    constexpr int kWaves = MfccCfg<R>::kWaves;
    const int want = (B * F + kWaves - 1) / kWaves;
    dim3 grid(want < kMfccMaxBlocks ? want : kMfccMaxBlocks);
#define SG_MFCC_BWD(CACHED, DZ) \
    hipLaunchKernelGGL((mfcc_bwd_kernel<R, CACHED, DZ>), grid, dim3(kWaves * 64), 0, s, t, x, B, T, F, scale, d, dfeats, kCep, dframes)
    const int dzk = d.noise_dev ? 2 : (d.dither != 0.f ? 1 : 0);
    if (t.spec_cache) {
        if (dzk == 2) SG_MFCC_BWD(true, 2); else if (dzk == 1) SG_MFCC_BWD(true, 1); else SG_MFCC_BWD(true, 0);
    } else {
        if (dzk == 2) SG_MFCC_BWD(false, 2); else if (dzk == 1) SG_MFCC_BWD(false, 1); else SG_MFCC_BWD(false, 0);
    }
#undef SG_MFCC_BWD
}

hipError_t launch_mfcc_fwd(const MfccTables& t, const float* x, int B, int T, int F, const float* scale,
                           const sg_dither* dz, float* feats, hipStream_t s) {
    sg_dither d = dz ? *dz : sg_dither{0.f, 0, 0, nullptr};
    if (t.fft64) mfcc_fwd_launch<double>(t, x, B, T, F, scale, d, feats, s);
    else mfcc_fwd_launch<float>(t, x, B, T, F, scale, d, feats, s);
    return hipGetLastError();
}

hipError_t launch_mfcc_bwd(const MfccTables& t, const float* x, int B, int T, int F, const float* scale,
                           const sg_dither* dz, const float* dfeats, float* dframes, hipStream_t s) {
    sg_dither d = dz ? *dz : sg_dither{0.f, 0, 0, nullptr};
    if (t.fft64) mfcc_bwd_launch<double>(t, x, B, T, F, scale, d, dfeats, dframes, s);
    else mfcc_bwd_launch<float>(t, x, B, T, F, scale, d, dfeats, dframes, s);
    return hipGetLastError();
}

hipError_t launch_frames_to_wave(const float* dframes, int B, int T, int F, int R, const float* acc_in, float* grad_out,
                                 float* x_io, const float* lower, const float* upper, float step, int grad_sign,
                                 hipStream_t s) {
    // four samples per thread once the launch is bandwidth-bound (19.4 -> ~12 us at 64 utterances); small batches are
    // latency-bound and keep one sample per thread (5.1 us at 8 utterances against 6.5)
    // (T % 4 and 16-byte aligned bases: every row of the caller-owned buffers is read / written as 16-byte vectors)
    const uintptr_t bases = reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(x_io) |
                            reinterpret_cast<uintptr_t>(lower) | reinterpret_cast<uintptr_t>(upper) |
                            reinterpret_cast<uintptr_t>(acc_in) | reinterpret_cast<uintptr_t>(dframes);
    if (T % 4 == 0 && bases % 16 == 0 && (long)B * T >= 16L * 48000) {
        dim3 grid((T / 4 + 255) / 256, B);
        hipLaunchKernelGGL(frames_to_wave_kernel<4>, grid, dim3(256), 0, s, dframes, B, T, F, R, acc_in, grad_out, x_io, lower, upper,
                           step, grad_sign);
    } else {
        dim3 grid((T + 255) / 256, B);
        hipLaunchKernelGGL(frames_to_wave_kernel<1>, grid, dim3(256), 0, s, dframes, B, T, F, R, acc_in, grad_out, x_io, lower, upper,
                           step, grad_sign);
    }
    return hipGetLastError();
}

}  // namespace sg
