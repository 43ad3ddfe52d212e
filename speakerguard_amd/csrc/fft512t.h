// The one-wave FFT-512 of fft512.h with the scalar type as a template parameter (round 5): the AudioNet log-mel front-end
// runs it in float32 -- the reference's own STFT is float32 (model/_audionet/Preprocessor.py:100-105 -> torch.stft) -- or in
// float64 (the form of rounds 1-4, kept as the counterpart).  Twiddles are always computed in float64 and rounded once.
// Same three radix-8 passes, same padded buffer (element i at SP(i) = i + i / 8), same lane <-> element maps; with 8-byte
// elements the pass-2 / pass-3 gathers (lane (k1, b): 72 k1 + b + 9 a) stay conflict-free for ds_read_b64 (32 lanes over
// 64 banks: 8 k1 + b covers 32 distinct 8-byte slots) and the scatters for ds_write_b64 (16 lanes: 9 b + 8 k1 mod 16
// distinct).
#pragma once
#include <hip/hip_runtime.h>

#include "fft512.h"

namespace sg {

template <typename T> struct Cx;
template <> struct Cx<float> { using type = float2; };
template <> struct Cx<double> { using type = double2; };
template <typename T> using cx = typename Cx<T>::type;

// No implicit contraction in the front-end's arithmetic (round 5): the same per-frame functions are instantiated in five
// kernels (forward, adjoint with / without the caches, with / without the fused overlap-add) that must give the SAME bits,
// and what -ffp-contract fuses depends on the code around an expression.  Every fused multiply-add is written out.
#pragma clang fp contract(off)
__device__ __forceinline__ float fmaT(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fmaT(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T> __device__ __forceinline__ cx<T> cmk(T x, T y) { cx<T> r; r.x = x; r.y = y; return r; }
template <typename T> __device__ __forceinline__ cx<T> cmulT(cx<T> a, cx<T> b) {
    return cmk<T>(fmaT(a.x, b.x, -(a.y * b.y)), fmaT(a.x, b.y, a.y * b.x));
}
template <typename T> __device__ __forceinline__ cx<T> caddT(cx<T> a, cx<T> b) { return cmk<T>(a.x + b.x, a.y + b.y); }
template <typename T> __device__ __forceinline__ cx<T> csubT(cx<T> a, cx<T> b) { return cmk<T>(a.x - b.x, a.y - b.y); }
template <typename T> __device__ __forceinline__ cx<T> cconjT(cx<T> a) { return cmk<T>(a.x, -a.y); }
// multiply by sgn * i
template <typename T> __device__ __forceinline__ cx<T> cmuliT(cx<T> a, T sgn) { return cmk<T>(-sgn * a.y, sgn * a.x); }
template <typename T> __device__ __forceinline__ cx<T> tw_sgnT(cx<T> w, T sgn) { return cmk<T>(w.x, -sgn * w.y); }

template <typename T>
__device__ __forceinline__ void dft8T(cx<T>& a0, cx<T>& a1, cx<T>& a2, cx<T>& a3, cx<T>& a4, cx<T>& a5, cx<T>& a6, cx<T>& a7, T sgn) {
    const T h = (T)0.70710678118654752440;
    cx<T> b0 = caddT<T>(a0, a4), b4 = csubT<T>(a0, a4), b1 = caddT<T>(a1, a5), b5 = csubT<T>(a1, a5);
    cx<T> b2 = caddT<T>(a2, a6), b6 = csubT<T>(a2, a6), b3 = caddT<T>(a3, a7), b7 = csubT<T>(a3, a7);
    b5 = cmulT<T>(b5, cmk<T>(h, sgn * h));
    b6 = cmuliT<T>(b6, sgn);
    b7 = cmulT<T>(b7, cmk<T>(-h, sgn * h));
    const cx<T> c0 = caddT<T>(b0, b2), c2 = csubT<T>(b0, b2), c1 = caddT<T>(b1, b3), c3 = cmuliT<T>(csubT<T>(b1, b3), sgn);
    const cx<T> c4 = caddT<T>(b4, b6), c6 = csubT<T>(b4, b6), c5 = caddT<T>(b5, b7), c7 = cmuliT<T>(csubT<T>(b5, b7), sgn);
    a0 = caddT<T>(c0, c1); a1 = caddT<T>(c4, c5); a2 = caddT<T>(c2, c3); a3 = caddT<T>(c6, c7);
    a4 = csubT<T>(c0, c1); a5 = csubT<T>(c4, c5); a6 = csubT<T>(c2, c3); a7 = csubT<T>(c6, c7);
}

// tw1[(j - 1) * 64 + lane] = W512^(j lane), tw2[9 b + c] = W64^(b c) (fft512.h), from the float64 half circle of
// W512^i (i < 256), rounded to T
template <typename T>
__device__ __forceinline__ void fft512_fill_tablesT(const double2* __restrict__ half_circle, cx<T>* tw1, cx<T>* tw2) {
    auto w512 = [&](int m) {
        const double2 w = half_circle[m & 255];
        return (m & 256) ? cmk<T>((T)-w.x, (T)-w.y) : cmk<T>((T)w.x, (T)w.y);
    };
    for (int i = threadIdx.x; i < kFftTw1; i += blockDim.x) tw1[i] = w512((i / 64 + 1) * (i % 64));
    for (int i = threadIdx.x; i < kFftTw2; i += blockDim.x) {
        const int b = i / 9, c = i - 9 * b;
        tw2[i] = c < 8 ? w512(8 * b * c) : cmk<T>((T)0, (T)0);
    }
}

// pass 1 with the input in registers: lane n2 holds v_j = x[n2 + 64 j]; leaves y[k1][n2] at SP(n2 + 64 k1)
template <typename T>
__device__ __forceinline__ void fft512_pass1T(cx<T>* buf, const cx<T>* __restrict__ tw1, int lane, T sgn, cx<T> v0, cx<T> v1, cx<T> v2,
                                              cx<T> v3, cx<T> v4, cx<T> v5, cx<T> v6, cx<T> v7) {
    cx<T>* p1 = buf + SP(lane);
    dft8T<T>(v0, v1, v2, v3, v4, v5, v6, v7, sgn);
    p1[0] = v0;
    p1[72] = cmulT<T>(v1, tw_sgnT<T>(tw1[lane], sgn));
    p1[144] = cmulT<T>(v2, tw_sgnT<T>(tw1[64 + lane], sgn));
    p1[216] = cmulT<T>(v3, tw_sgnT<T>(tw1[128 + lane], sgn));
    p1[288] = cmulT<T>(v4, tw_sgnT<T>(tw1[192 + lane], sgn));
    p1[360] = cmulT<T>(v5, tw_sgnT<T>(tw1[256 + lane], sgn));
    p1[432] = cmulT<T>(v6, tw_sgnT<T>(tw1[320 + lane], sgn));
    p1[504] = cmulT<T>(v7, tw_sgnT<T>(tw1[384 + lane], sgn));
    wave_sync();
}
template <typename T>
__device__ __forceinline__ void fft512_pass2T(cx<T>* buf, const cx<T>* __restrict__ tw2, int lane, T sgn) {
    const int k1 = lane >> 3, b = lane & 7;
    const cx<T>* r = buf + k1 * 72 + b;
    cx<T> v0 = r[0], v1 = r[9], v2 = r[18], v3 = r[27], v4 = r[36], v5 = r[45], v6 = r[54], v7 = r[63];
    wave_sync();
    dft8T<T>(v0, v1, v2, v3, v4, v5, v6, v7, sgn);
    cx<T>* w = buf + k1 * 72 + 9 * b;
    const cx<T>* t2 = tw2 + 9 * b;
    w[0] = v0;
    w[1] = cmulT<T>(v1, tw_sgnT<T>(t2[1], sgn));
    w[2] = cmulT<T>(v2, tw_sgnT<T>(t2[2], sgn));
    w[3] = cmulT<T>(v3, tw_sgnT<T>(t2[3], sgn));
    w[4] = cmulT<T>(v4, tw_sgnT<T>(t2[4], sgn));
    w[5] = cmulT<T>(v5, tw_sgnT<T>(t2[5], sgn));
    w[6] = cmulT<T>(v6, tw_sgnT<T>(t2[6], sgn));
    w[7] = cmulT<T>(v7, tw_sgnT<T>(t2[7], sgn));
    wave_sync();
}
// pass 3: lane = (k1, c) = (lane >> 3, lane & 7); out[d] = X[k1 + 8 c + 64 d].  Ends with the fence that lets the caller
// overwrite the buffer.
template <typename T>
__device__ __forceinline__ void fft512_pass3T(const cx<T>* buf, int lane, T sgn, cx<T> (&out)[8]) {
    const int k1 = lane >> 3, c = lane & 7;
    const cx<T>* r = buf + k1 * 72 + c;
    cx<T> v0 = r[0], v1 = r[9], v2 = r[18], v3 = r[27], v4 = r[36], v5 = r[45], v6 = r[54], v7 = r[63];
    wave_sync();
    dft8T<T>(v0, v1, v2, v3, v4, v5, v6, v7, sgn);
    out[0] = v0; out[1] = v1; out[2] = v2; out[3] = v3; out[4] = v4; out[5] = v5; out[6] = v6; out[7] = v7;
}

// the full transform, natural order in LDS in and out (element i at SP(i))
template <typename T>
__device__ __forceinline__ void fft512T(cx<T>* buf, const cx<T>* __restrict__ tw1, const cx<T>* __restrict__ tw2, int lane, T sgn) {
    const cx<T>* p1 = buf + SP(lane);
    fft512_pass1T<T>(buf, tw1, lane, sgn, p1[0], p1[72], p1[144], p1[216], p1[288], p1[360], p1[432], p1[504]);
    fft512_pass2T<T>(buf, tw2, lane, sgn);
    cx<T> v[8];
    fft512_pass3T<T>(buf, lane, sgn, v);
    cx<T>* w = buf + (lane >> 3) + 9 * (lane & 7);  // SP(k1 + 8 c + 64 d) = k1 + 9 c + 72 d
    w[0] = v[0]; w[72] = v[1]; w[144] = v[2]; w[216] = v[3]; w[288] = v[4]; w[360] = v[5]; w[432] = v[6]; w[504] = v[7];
    wave_sync();
}
// input from registers (v_j = x[lane + 64 j]), output to LDS in natural order
template <typename T>
__device__ __forceinline__ void fft512T_regin(cx<T>* buf, const cx<T>* __restrict__ tw1, const cx<T>* __restrict__ tw2, int lane, T sgn,
                                              const cx<T> (&in)[8]) {
    fft512_pass1T<T>(buf, tw1, lane, sgn, in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7]);
    fft512_pass2T<T>(buf, tw2, lane, sgn);
    cx<T> v[8];
    fft512_pass3T<T>(buf, lane, sgn, v);
    cx<T>* w = buf + (lane >> 3) + 9 * (lane & 7);
    w[0] = v[0]; w[72] = v[1]; w[144] = v[2]; w[216] = v[3]; w[288] = v[4]; w[360] = v[5]; w[432] = v[6]; w[504] = v[7];
    wave_sync();
}

// registers in (v_j = x[lane + 64 j]), registers out (lane (k1, c) = (lane >> 3, lane & 7): out[d] = X[k1 + 8 c + 64 d] -- "transform
// order"); the buffer is free again on return
template <typename T>
__device__ __forceinline__ void fft512T_regs(cx<T>* buf, const cx<T>* __restrict__ tw1, const cx<T>* __restrict__ tw2, int lane, T sgn,
                                             const cx<T> (&in)[8], cx<T> (&out)[8]) {
    fft512_pass1T<T>(buf, tw1, lane, sgn, in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7]);
    fft512_pass2T<T>(buf, tw2, lane, sgn);
    fft512_pass3T<T>(buf, lane, sgn, out);
}
// element k = k1 + 8 c + 64 d of a spectrum kept in transform order sits at 64 d + lane, lane = 8 k1 + c
__host__ __device__ __forceinline__ int fft512_transform_pos(int k) { return (k & ~63) + 8 * (k & 7) + ((k >> 3) & 7); }

// The TRANSPOSED network (round 5): the three passes run backwards, each one transposed.  The DFT matrix is symmetric, so
// this is the same transform with the roles of the two layouts swapped: input in transform order (lane (k1, c):
// v[d] = G[k1 + 8 c + 64 d]), output in natural order (v[j] = g[lane + 64 j]), both in registers.  After a spectrum that
// was produced in transform order by fft512T_regs, the inverse needs no exchange in front of it and none behind it: two LDS
// round trips instead of five (the adjoint of the log-mel front-end, k_audionet.hip).
//   T3: DFT-8 over d -> b, times W64^(b c), to 72 k1 + 9 b + c        (the pass-2 gather pattern, as a scatter)
//   T2: lane (k1, b) reads c = 0..7, DFT-8 over c -> a, to 72 k1 + 9 a + b  (the pass-2 scatter / gather patterns swapped)
//   T1: lane n2 reads SP(n2) + 72 k1, times W512^(k1 n2), DFT-8 over k1 -> j
template <typename T>
__device__ __forceinline__ void fft512T_transposed(cx<T>* buf, const cx<T>* __restrict__ tw1, const cx<T>* __restrict__ tw2, int lane, T sgn,
                                                   cx<T> (&v)[8]) {
    const int k1 = lane >> 3, c = lane & 7;
    dft8T<T>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], sgn);
    {
        const cx<T>* t2 = tw2 + 9 * c;
        cx<T>* w = buf + k1 * 72 + c;
        w[0] = v[0];
#pragma unroll
        for (int b = 1; b < 8; ++b) w[9 * b] = cmulT<T>(v[b], tw_sgnT<T>(t2[b], sgn));
    }
    wave_sync();
    {
        const cx<T>* r = buf + k1 * 72 + 9 * c;  // this lane as (k1, b = lane & 7)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = r[i];
    }
    wave_sync();
    dft8T<T>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], sgn);
    {
        cx<T>* w = buf + k1 * 72 + c;
#pragma unroll
        for (int a = 0; a < 8; ++a) w[9 * a] = v[a];
    }
    wave_sync();
    {
        const cx<T>* p1 = buf + SP(lane);
        v[0] = p1[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) v[j] = cmulT<T>(p1[72 * j], tw_sgnT<T>(tw1[(j - 1) * 64 + lane], sgn));
    }
    wave_sync();
    dft8T<T>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], sgn);
}

#pragma clang fp contract(fast)

}  // namespace sg
