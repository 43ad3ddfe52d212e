// One-wave fp64 FFT-512 on a private, padded LDS buffer -- shared by the MFCC front-end (k_mfcc.hip, 512-point
// frames) and the AudioNet log-mel front-end (k_audionet.hip, 1024-point REAL frames done as one 512-point complex
// transform plus a split step).
#pragma once
#include <hip/hip_runtime.h>

namespace sg {

// Every LDS buffer here is private to one wave, and a wave's DS instructions execute in program order, so cross-lane
// hand-offs through LDS need no s_barrier -- only a fence that stops the compiler from moving LDS accesses across it.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 512-point complex FFT of one wave's private LDS buffer, fp64, natural order in and out.
// 512 = 8 x 8 x 8: three passes of in-register radix-8 butterflies, one butterfly per lane and pass,
// with an LDS exchange between passes (Cooley-Tukey, decimation in frequency):
//   n = 64 n1 + n2,  k = k1 + 8 c + 64 d
//   pass 1  lane n2       : DFT8 over n1, times W512^(n2 k1)          -> y[k1][n2]
//   pass 2  lane (k1, b)  : DFT8 over a of y[k1][8a + b], times W64^(b c) -> z[k1][b][c]
//   pass 3  lane (k1, c)  : DFT8 over b of z[k1][b][c]                 -> X[k1 + 8c + 64d]
// 48 LDS accesses per lane instead of the 144 of a radix-2 network, and 6 wave-level fences instead
// of 9 (measured on the MFCC forward kernel: FFT share 55 us -> see profiles/).
// sgn = -1: forward transform; +1: unnormalised inverse (conjugate twiddles).
// The work buffer keeps element i at SP(i) = i + i/8 (one 16-byte pad per 8 elements). ds_write_b128
// is served in groups of 8 consecutive lanes over 32 banks (128 B): unpadded, the pass-2 and pass-3
// scatters put all 8 lanes of a group on one bank slot (stride 128 B) -- 8x the LDS cycles, which made
// the FFT ~1500 LDS cycles per frame and the whole kernel LDS-issue bound. With the pad every access
// pattern below is conflict-free within its lane group.
#define SP(i) ((i) + ((i) >> 3))
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
// multiply by sgn * i
__device__ __forceinline__ double2 cmuli(double2 a, double sgn) { return make_double2(-sgn * a.y, sgn * a.x); }

#define SG_DFT8(a0, a1, a2, a3, a4, a5, a6, a7, sgn)                                               \
    {                                                                                              \
        const double h = 0.70710678118654752440;                                                   \
        double2 b0 = cadd(a0, a4), b4 = csub(a0, a4), b1 = cadd(a1, a5), b5 = csub(a1, a5);        \
        double2 b2 = cadd(a2, a6), b6 = csub(a2, a6), b3 = cadd(a3, a7), b7 = csub(a3, a7);        \
        b5 = cmul(b5, make_double2(h, (sgn) * h));                                                 \
        b6 = cmuli(b6, (sgn));                                                                     \
        b7 = cmul(b7, make_double2(-h, (sgn) * h));                                                \
        const double2 c0 = cadd(b0, b2), c2 = csub(b0, b2), c1 = cadd(b1, b3), c3 = cmuli(csub(b1, b3), (sgn)); \
        const double2 c4 = cadd(b4, b6), c6 = csub(b4, b6), c5 = cadd(b5, b7), c7 = cmuli(csub(b5, b7), (sgn)); \
        a0 = cadd(c0, c1); a1 = cadd(c4, c5); a2 = cadd(c2, c3); a3 = cadd(c6, c7);                \
        a4 = csub(c0, c1); a5 = csub(c4, c5); a6 = csub(c2, c3); a7 = csub(c6, c7);                \
    }

// W512^m, m in [0, 512), from the half-circle table tw[256] (W^(m+256) = -W^m); sgn = +1 conjugates
__device__ __forceinline__ double2 tw512(const double2* __restrict__ tw, int m, double sgn) {
    const double2 w = tw[m & 255];
    const double f = (m & 256) ? -1.0 : 1.0;
    return make_double2(f * w.x, -sgn * f * w.y);
}

// The three passes, separately, with the transform's input and output in REGISTERS: a caller that knows its input (real,
// zero-padded: the MFCC forward; one-sided: the gradient spectrum) or needs only part of the output (bins 0..255; the real
// part of samples 0..399) skips the LDS traffic of the parts it does not need -- 16-byte LDS accesses are what these
// kernels are short of (PMC: the LDS pipe is busy 54 % of the MFCC forward kernel, 31 % of that in bank conflicts).
// pass 1: lane n2 holds v_j = x[n2 + 64 j]; leaves y[k1][n2] at SP(n2 + 64 k1).
__device__ __forceinline__ void fft512_pass1(double2* buf, const double2* __restrict__ tw, int lane, double sgn, double2 v0,
                                             double2 v1, double2 v2, double2 v3, double2 v4, double2 v5, double2 v6, double2 v7) {
    double2* p1 = buf + SP(lane);  // SP(lane + 64 j) = SP(lane) + 72 j
    SG_DFT8(v0, v1, v2, v3, v4, v5, v6, v7, sgn)
    p1[0] = v0;
    p1[72] = cmul(v1, tw512(tw, lane, sgn));
    p1[144] = cmul(v2, tw512(tw, 2 * lane, sgn));
    p1[216] = cmul(v3, tw512(tw, 3 * lane, sgn));
    p1[288] = cmul(v4, tw512(tw, 4 * lane, sgn));
    p1[360] = cmul(v5, tw512(tw, 5 * lane, sgn));
    p1[432] = cmul(v6, tw512(tw, 6 * lane, sgn));
    p1[504] = cmul(v7, tw512(tw, 7 * lane, sgn));
    wave_sync();
}
// pass 2: lane = (k1, b), in place
__device__ __forceinline__ void fft512_pass2(double2* buf, const double2* __restrict__ tw, int lane, double sgn) {
    double2 v0, v1, v2, v3, v4, v5, v6, v7;
    const int k1 = lane >> 3, b = lane & 7;
    const double2* r = buf + k1 * 72 + b;  // SP(64 k1 + 8 a + b) = 72 k1 + 9 a + b
    v0 = r[0]; v1 = r[9]; v2 = r[18]; v3 = r[27]; v4 = r[36]; v5 = r[45]; v6 = r[54]; v7 = r[63];
    wave_sync();
    SG_DFT8(v0, v1, v2, v3, v4, v5, v6, v7, sgn)
    double2* w = buf + k1 * 72 + 9 * b;  // z[k1][b][c] at SP(64 k1 + 8 b + c)
    w[0] = v0;
    w[1] = cmul(v1, tw512(tw, 8 * b, sgn));
    w[2] = cmul(v2, tw512(tw, 16 * b, sgn));
    w[3] = cmul(v3, tw512(tw, 24 * b, sgn));
    w[4] = cmul(v4, tw512(tw, 32 * b, sgn));
    w[5] = cmul(v5, tw512(tw, 40 * b, sgn));
    w[6] = cmul(v6, tw512(tw, 48 * b, sgn));
    w[7] = cmul(v7, tw512(tw, 56 * b, sgn));
    wave_sync();
}
// pass 3: lane = (k1, c) = (lane >> 3, lane & 7); out[d] = X[k1 + 8 c + 64 d].  Ends with the fence that lets the caller
// overwrite the buffer.
__device__ __forceinline__ void fft512_pass3(const double2* buf, int lane, double sgn, double2 (&out)[8]) {
    const int k1 = lane >> 3, c = lane & 7;
    const double2* r = buf + k1 * 72 + c;
    double2 v0 = r[0], v1 = r[9], v2 = r[18], v3 = r[27], v4 = r[36], v5 = r[45], v6 = r[54], v7 = r[63];
    wave_sync();
    SG_DFT8(v0, v1, v2, v3, v4, v5, v6, v7, sgn)
    out[0] = v0; out[1] = v1; out[2] = v2; out[3] = v3; out[4] = v4; out[5] = v5; out[6] = v6; out[7] = v7;
}

// The same passes 1 and 2 with their twiddles in dedicated tables (PMC, MFCC forward: of 240 bank-conflict cycles per
// frame 152 came from the transform, and the butterfly exchanges are conflict-free -- it was the GATHER of twiddles from the
// half-circle table: pass 2 reads tw[8 b c], eight distinct addresses per 16-lane group all on one or two 16-byte slots of
// the 256-byte bank row, 4- to 8-way, ~130 extra LDS cycles per transform):
//   tw1[(j - 1) * 64 + lane] = W512^(j lane), j = 1..7   (lane-linear reads: conflict-free)
//   tw2[9 b + c]             = W64^(b c),     b, c < 8   (row pad 1: the eight b of a lane group fall on eight slots)
// both for the forward sign; sgn = +1 conjugates.
constexpr int kFftTw1 = 7 * 64, kFftTw2 = 72;
__device__ __forceinline__ double2 tw_sgn(double2 w, double sgn) { return make_double2(w.x, -sgn * w.y); }
__device__ __forceinline__ void fft512_fill_tables(const double2* __restrict__ half_circle, double2* tw1, double2* tw2) {
    auto w512 = [&](int m) {
        const double2 w = half_circle[m & 255];
        return (m & 256) ? make_double2(-w.x, -w.y) : w;
    };
    for (int i = threadIdx.x; i < kFftTw1; i += blockDim.x) tw1[i] = w512((i / 64 + 1) * (i % 64));
    for (int i = threadIdx.x; i < kFftTw2; i += blockDim.x) {
        const int b = i / 9, c = i - 9 * b;
        tw2[i] = c < 8 ? w512(8 * b * c) : make_double2(0.0, 0.0);
    }
}
__device__ __forceinline__ void fft512_pass1_t(double2* buf, const double2* __restrict__ tw1, int lane, double sgn, double2 v0,
                                               double2 v1, double2 v2, double2 v3, double2 v4, double2 v5, double2 v6, double2 v7) {
    double2* p1 = buf + SP(lane);
    SG_DFT8(v0, v1, v2, v3, v4, v5, v6, v7, sgn)
    p1[0] = v0;
    p1[72] = cmul(v1, tw_sgn(tw1[lane], sgn));
    p1[144] = cmul(v2, tw_sgn(tw1[64 + lane], sgn));
    p1[216] = cmul(v3, tw_sgn(tw1[128 + lane], sgn));
    p1[288] = cmul(v4, tw_sgn(tw1[192 + lane], sgn));
    p1[360] = cmul(v5, tw_sgn(tw1[256 + lane], sgn));
    p1[432] = cmul(v6, tw_sgn(tw1[320 + lane], sgn));
    p1[504] = cmul(v7, tw_sgn(tw1[384 + lane], sgn));
    wave_sync();
}
__device__ __forceinline__ void fft512_pass2_t(double2* buf, const double2* __restrict__ tw2, int lane, double sgn) {
    double2 v0, v1, v2, v3, v4, v5, v6, v7;
    const int k1 = lane >> 3, b = lane & 7;
    const double2* r = buf + k1 * 72 + b;
    v0 = r[0]; v1 = r[9]; v2 = r[18]; v3 = r[27]; v4 = r[36]; v5 = r[45]; v6 = r[54]; v7 = r[63];
    wave_sync();
    SG_DFT8(v0, v1, v2, v3, v4, v5, v6, v7, sgn)
    double2* w = buf + k1 * 72 + 9 * b;
    const double2* t2 = tw2 + 9 * b;
    w[0] = v0;
    w[1] = cmul(v1, tw_sgn(t2[1], sgn));
    w[2] = cmul(v2, tw_sgn(t2[2], sgn));
    w[3] = cmul(v3, tw_sgn(t2[3], sgn));
    w[4] = cmul(v4, tw_sgn(t2[4], sgn));
    w[5] = cmul(v5, tw_sgn(t2[5], sgn));
    w[6] = cmul(v6, tw_sgn(t2[6], sgn));
    w[7] = cmul(v7, tw_sgn(t2[7], sgn));
    wave_sync();
}

// the full in-place transform with the twiddle tables
__device__ __forceinline__ void fft512_r8_t(double2* buf, const double2* __restrict__ tw1, const double2* __restrict__ tw2, int lane,
                                            double sgn) {
    const double2* p1 = buf + SP(lane);
    fft512_pass1_t(buf, tw1, lane, sgn, p1[0], p1[72], p1[144], p1[216], p1[288], p1[360], p1[432], p1[504]);
    fft512_pass2_t(buf, tw2, lane, sgn);
    double2 v[8];
    fft512_pass3(buf, lane, sgn, v);
    double2* w = buf + (lane >> 3) + 9 * (lane & 7);  // SP(k1 + 8 c + 64 d) = k1 + 9 c + 72 d
    w[0] = v[0]; w[72] = v[1]; w[144] = v[2]; w[216] = v[3]; w[288] = v[4]; w[360] = v[5]; w[432] = v[6]; w[504] = v[7];
    wave_sync();
}

__device__ __forceinline__ void fft512_r8(double2* buf, const double2* __restrict__ tw, int lane, double sgn) {
    const double2* p1 = buf + SP(lane);
    fft512_pass1(buf, tw, lane, sgn, p1[0], p1[72], p1[144], p1[216], p1[288], p1[360], p1[432], p1[504]);
    fft512_pass2(buf, tw, lane, sgn);
    double2 v[8];
    fft512_pass3(buf, lane, sgn, v);
    double2* w = buf + (lane >> 3) + 9 * (lane & 7);  // SP(k1 + 8 c + 64 d) = k1 + 9 c + 72 d
    w[0] = v[0]; w[72] = v[1]; w[144] = v[2]; w[216] = v[3]; w[288] = v[4]; w[360] = v[5]; w[432] = v[6]; w[504] = v[7];
    wave_sync();
}


}  // namespace sg
