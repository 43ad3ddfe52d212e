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
// 512-point FFT runs in LDS (radix-2 DIT, 9 stages, table twiddles), reductions use wavefront
// shuffles.  The FORWARD transform runs in fp64: a windowed speech frame has > 80 dB between its
// strongest harmonic and the weak bins, and an fp32 FFT leaves a round-off floor of ~2e-7 of the
// strongest bin on every bin -- a 1e-3 relative error on exactly the weak mel bands whose
// 1/energy weights dominate d(log-mel)/dx; the inverse transform of the spectrum gradient has the
// same problem in the other direction (an absolute error floor on every sample, which flips the
// sign of small gradient entries).  Measured with fp32 transforms: 0.8 % gradient sign mismatches
// against the oracle, whose own fp32-vs-fp64 noise is 0.03 %.
// The backward kernel recomputes the forward of its frame (cheaper than storing 514 floats per
// frame), then walks the stages in reverse; the spectrum gradient overwrites the spectrum in
// place and is transformed back by a decimation-in-frequency pass with conjugate twiddles
// (natural-order input, bit-reversed output -- no second buffer).  It writes per-frame sample
// gradients (B,F,400); frames_to_wave_kernel does the deterministic overlap-add.
#include "sg_internal.h"

namespace sg {

constexpr int kWavesPerBlock = 4;
constexpr int kFramesPerWave = 4;
constexpr int kFramesPerBlock = kWavesPerBlock * kFramesPerWave;

struct FrameLds {
    double2 spec[kFft];  // FFT work buffer: spectrum, then (backward) its gradient; fp64, see header
    float samp[kFft];    // DC-removed samples, later windowed-gradient
    float power[256];
    float mel[32];
    float lmel[32];
    float tmp[32];
};

// Every LDS buffer in this file is private to one wave, and a wave's DS instructions execute in
// program order, so cross-lane hand-offs through LDS need no s_barrier -- only a fence that stops
// the compiler from moving LDS accesses across it (a block-wide __syncthreads() here coupled the
// four independent waves at ~25 points per frame).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
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

// In-place radix-2 FFTs of 512 complex fp64 points held in LDS.  All four waves of the block run
// them on its own buffer (wave-level sync per stage).
// Forward: decimation in time, input scattered in bit-reversed order, output in natural order.
__device__ __forceinline__ void fft512_dit(double2* buf, const double2* __restrict__ tw, int lane) {
#pragma unroll 1
    for (int s = 0; s < 9; ++s) {
        const int half = 1 << s;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = lane + 64 * i;
            const int pos = j & (half - 1);
            const int i0 = ((j >> s) << (s + 1)) + pos;
            const int i1 = i0 + half;
            const double2 w = tw[pos << (8 - s)];
            const double2 a = buf[i0], b = buf[i1];
            const double tr = b.x * w.x - b.y * w.y;
            const double ti = b.x * w.y + b.y * w.x;
            buf[i0] = make_double2(a.x + tr, a.y + ti);
            buf[i1] = make_double2(a.x - tr, a.y - ti);
        }
        wave_sync();
    }
}

// Inverse (conjugate twiddles, unnormalised): decimation in frequency, natural-order input,
// output element n lands at buf[bitrev(n)].
__device__ __forceinline__ void ifft512_dif(double2* buf, const double2* __restrict__ tw, int lane) {
#pragma unroll 1
    for (int s = 8; s >= 0; --s) {
        const int half = 1 << s;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = lane + 64 * i;
            const int pos = j & (half - 1);
            const int i0 = ((j >> s) << (s + 1)) + pos;
            const int i1 = i0 + half;
            const double2 w = tw[pos << (8 - s)];  // conj(w) = (w.x, -w.y)
            const double2 a = buf[i0], b = buf[i1];
            const double dx = a.x - b.x, dy = a.y - b.y;
            buf[i0] = make_double2(a.x + b.x, a.y + b.y);
            buf[i1] = make_double2(dx * w.x + dy * w.y, dy * w.x - dx * w.y);
        }
        wave_sync();
    }
}

struct FrameState {
    float s[7];      // DC-removed samples n = lane + 64 i
    float energy;    // sum of squares (before log)
};

// Forward of one frame up to the cepstra; leaves spectrum in L.spec, mel in L.mel, samples in L.samp.
__device__ __forceinline__ void frame_forward(const MfccTables& t, FrameLds& L, const float* __restrict__ x, int T,
                                              int F, int b, int f, bool active, float scale, const sg_dither& dz,
                                              int lane, FrameState& st, float& cep_out) {
    const int base = f * kShift - (kWin / 2 - kShift / 2);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int n = lane + 64 * i;
        float v = 0.f;
        if (active && n < kWin) {
            const int p = base + n;
            const int idx = p < 0 ? -p - 1 : (p >= T ? 2 * T - 1 - p : p);
            v = x[(size_t)b * T + idx] * scale;
            if (dz.noise_dev) v += dz.noise_dev[((size_t)b * F + f) * kWin + n];
            else if (dz.dither != 0.f) v += dither_draw(dz.seed, dz.index_base + b, f, n, dz.dither);
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
            e += st.s[i] * st.s[i];
            L.samp[n] = st.s[i];
        }
    }
    st.energy = wave_sum(e);
    wave_sync();
    // pre-emphasis (replicate pad on the left), povey window, bit-reversed scatter into the FFT buffer
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = lane + 64 * i;
        float w = 0.f;
        if (n < kWin) {
            const float prev = L.samp[n > 0 ? n - 1 : 0];
            w = (st.s[i < 7 ? i : 6] - 0.97f * prev) * t.window[n];
        }
        L.spec[t.bitrev[n]] = make_double2((double)w, 0.0);
    }
    wave_sync();
    fft512_dit(L.spec, t.twiddle, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane + 64 * i;
        const double2 c = L.spec[k];
        L.power[k] = (float)(c.x * c.x + c.y * c.y);
    }
    wave_sync();
    if (lane < kMel) {
        float acc = 0.f;
        const int lo = t.mel_lo[lane], hi = t.mel_hi[lane];
        for (int k = lo; k < hi; ++k) acc += L.power[k] * t.mel_w[lane * 256 + k];
        L.mel[lane] = acc;
        L.lmel[lane] = logf(fmaxf(acc, kEps));
    }
    wave_sync();
    cep_out = 0.f;
    if (lane < kCep) {
        float v = 0.f;
#pragma unroll 6
        for (int m = 0; m < kMel; ++m) v += L.lmel[m] * t.dct[m * kCep + lane];
        v *= t.lifter[lane];
        if (lane == 0) v = logf(fmaxf(st.energy, kEps));
        cep_out = v;
    }
}

__global__ __launch_bounds__(256) void mfcc_fwd_kernel(MfccTables t, const float* __restrict__ x, int B, int T, int F,
                                                       const float* __restrict__ scale_p, sg_dither dz,
                                                       float* __restrict__ feats) {
    __shared__ FrameLds lds[kWavesPerBlock];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const float scale = scale_p ? *scale_p : 1.f;
    FrameLds& L = lds[wid];
    for (int it = 0; it < kFramesPerWave; ++it) {
        const int f = blockIdx.x * kFramesPerBlock + it * kWavesPerBlock + wid;
        const bool active = f < F;
        FrameState st;
        float cep;
        frame_forward(t, L, x, T, F, b, active ? f : 0, active, scale, dz, lane, st, cep);
        if (active && lane < kCep) feats[((size_t)b * F + f) * kCep + lane] = cep;
        wave_sync();
    }
}

// dfeats: (B,F,ld) gradient wrt the 30 cepstra; dframes: (B,F,400) gradient wrt the strided frames,
// already multiplied by `scale` (the int16 rescale of check_input_range, model/utils.py:14).
__global__ __launch_bounds__(256) void mfcc_bwd_kernel(MfccTables t, const float* __restrict__ x, int B, int T, int F,
                                                       const float* __restrict__ scale_p, sg_dither dz,
                                                       const float* __restrict__ dfeats, int ld,
                                                       float* __restrict__ dframes) {
    __shared__ FrameLds lds[kWavesPerBlock];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const float scale = scale_p ? *scale_p : 1.f;
    FrameLds& L = lds[wid];
    for (int it = 0; it < kFramesPerWave; ++it) {
        const int f = blockIdx.x * kFramesPerBlock + it * kWavesPerBlock + wid;
        const bool active = f < F;
        const int fa = active ? f : 0;
        FrameState st;
        float cep;
        frame_forward(t, L, x, T, F, b, fa, active, scale, dz, lane, st, cep);
        // ---- cepstra -> log-mel
        float dc = 0.f;
        if (active && lane < kCep) dc = dfeats[((size_t)b * F + fa) * ld + lane];
        const float denergy = __shfl(dc, 0, 64);
        if (lane < 32) L.tmp[lane] = (lane == 0 || lane >= kCep) ? 0.f : dc * t.lifter[lane];
        wave_sync();
        if (lane < 32) {
            float dm = 0.f;
            if (lane < kMel) {
                float dl = 0.f;
#pragma unroll 6
                for (int c = 0; c < kCep; ++c) dl += L.tmp[c] * t.dct[lane * kCep + c];
                const float mel = L.mel[lane];
                dm = mel > kEps ? dl / mel : 0.f;
            }
            L.lmel[lane] = dm;  // d loss / d mel energy (entries 30,31 = 0)
        }
        wave_sync();
        // ---- mel -> power -> spectrum gradient G[k] = 2 X[k] dP[k], in place, bins 256..511 zero
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = lane + 64 * i;
            double2 g = make_double2(0.0, 0.0);
            if (k < 256) {
                const int m0 = t.bin_m0[k];
                if (m0 >= 0) {
                    const double dp = 2.0 * (double)(L.lmel[m0] * t.bin_w0[k] + L.lmel[m0 + 1] * t.bin_w1[k]);
                    const double2 c = L.spec[k];
                    g = make_double2(c.x * dp, c.y * dp);
                }
            }
            L.spec[k] = g;
        }
        wave_sync();
        ifft512_dif(L.spec, t.twiddle, lane);
        // ---- window, pre-emphasis, energy, DC removal
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int n = lane + 64 * i;
            if (n < kWin) L.samp[n] = (float)L.spec[t.bitrev[n]].x * t.window[n];
        }
        wave_sync();
        float ds[7];
        float sum = 0.f;
        const float einv = st.energy > kEps ? 2.f * denergy / st.energy : 0.f;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int n = lane + 64 * i;
            float v = 0.f;
            if (n < kWin) {
                v = L.samp[n] - (n + 1 < kWin ? 0.97f * L.samp[n + 1] : 0.f);
                if (n == 0) v -= 0.97f * L.samp[0];
                v += einv * st.s[i];
            }
            ds[i] = v;
            sum += v;
        }
        const float mean = wave_sum(sum) / (float)kWin;
        if (active) {
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int n = lane + 64 * i;
                if (n < kWin) dframes[((size_t)b * F + f) * kWin + n] = (ds[i] - mean) * scale;
            }
        }
        wave_sync();
    }
}

// Deterministic overlap-add: sample n of utterance b gathers, in a fixed order, every frame
// position that maps to it -- directly, or through the reflected edges of _get_strided.
// Optionally fuses the PGD step (attack/FGSM.py:65,68) so the gradient never round-trips HBM.
__global__ __launch_bounds__(256) void frames_to_wave_kernel(const float* __restrict__ dframes, int B, int T, int F,
                                                             float* __restrict__ grad_out, float* __restrict__ x_io,
                                                             const float* __restrict__ lower,
                                                             const float* __restrict__ upper, float step, int grad_sign) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= T) return;
    const float* df = dframes + (size_t)b * F * kWin;
    constexpr int kPad = kWin / 2 - kShift / 2;
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
    const int pr = 2 * T - 1 - n;
    if (pr <= (F - 1) * kShift - kPad + kWin - 1) add_pos(pr);
    const size_t o = (size_t)b * T + n;
    if (grad_out) grad_out[o] = g;
    if (x_io) {
        const float sg = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
        float v = x_io[o] + step * sg * (float)grad_sign;
        v = fminf(fmaxf(v, lower[o]), upper[o]);
        x_io[o] = v;
    }
}

hipError_t launch_mfcc_fwd(const MfccTables& t, const float* x, int B, int T, int F, const float* scale,
                           const sg_dither* dz, float* feats, hipStream_t s) {
    sg_dither d = dz ? *dz : sg_dither{0.f, 0, 0, nullptr};
    dim3 grid((F + kFramesPerBlock - 1) / kFramesPerBlock, B);
    hipLaunchKernelGGL(mfcc_fwd_kernel, grid, dim3(256), 0, s, t, x, B, T, F, scale, d, feats);
    return hipGetLastError();
}

hipError_t launch_mfcc_bwd(const MfccTables& t, const float* x, int B, int T, int F, const float* scale,
                           const sg_dither* dz, const float* dfeats, float* dframes, hipStream_t s) {
    sg_dither d = dz ? *dz : sg_dither{0.f, 0, 0, nullptr};
    dim3 grid((F + kFramesPerBlock - 1) / kFramesPerBlock, B);
    hipLaunchKernelGGL(mfcc_bwd_kernel, grid, dim3(256), 0, s, t, x, B, T, F, scale, d, dfeats, kCep, dframes);
    return hipGetLastError();
}

hipError_t launch_frames_to_wave(const float* dframes, int B, int T, int F, float* grad_out, float* x_io,
                                 const float* lower, const float* upper, float step, int grad_sign,
                                 hipStream_t s) {
    dim3 grid((T + 255) / 256, B);
    hipLaunchKernelGGL(frames_to_wave_kernel, grid, dim3(256), 0, s, dframes, B, T, F, grad_out, x_io, lower, upper,
                       step, grad_sign);
    return hipGetLastError();
}

}  // namespace sg
