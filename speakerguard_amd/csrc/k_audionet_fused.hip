// The AudioNet CNN of one pass as TWO launches instead of ~22 ("whole-CNN-per-utterance fused
// fwd/bwd, channels <= 128: LDS-resident"):
//   an_cnn_fwd_kernel   5x5 pre-filter -> conv2..conv8 (+ folded BatchNorm, ReLU, MaxPool)     model/audionet_csine.py:176-207
//   an_cnn_bwd_kernel   d loss/d conv8 pre-activation -> d loss/d features (what autograd derives for the same lines)
//
// One block (8 waves, one per CU: the two activation buffers fill the LDS) per (utterance, time slice).  A layer's input sits
// in LDS -- written there by the previous layer's epilogue, never read back from memory --, its weights stream from the L2
// (k4-packed: a lane's W operand of a k-group is one coalesced 16-byte load, the whole stack is 0.6 MB and stays
// L2-resident), its output goes to the other LDS buffer and, for the backward pass's masks, to memory once.  The time
// axis of an utterance is cut into S slices so that B x S blocks fill the chip at small batches (64 utterances: 4
// slices); a slice recomputes the halo its receptive field reaches into (1-2 frames per layer, doubled by every pool
// below), writes only the rows it owns, and S = 1 from 256 utterances up.
//
// ARITHMETIC: every output element is the same float32 fmaf chain as in the per-layer kernels (k_conv_gemm.hip,
// restated in oracle/conv_chain.c): taps ascending, chunks of 32 channels ascending, k-groups of 8 ascending, inside a
// group 0, 4, 1, 5, 2, 6, 3, 7 (v_mfma_f32_32x32x2_f32 is a sequential fused multiply-add over its two k values); bias,
// ReLU, pooling, masks as in their epilogues and in an_pool_*_kernel.  So this path is BIT-IDENTICAL to the per-layer
// launch sequence it replaces (tests/test_gpu_audionet.py compares the two; SG_AN_FUSED=0 selects the old sequence), and
// a slice's halo rows equal its neighbour's own rows bit for bit: the result does not depend on S.
//
// LDS image of an activation tile [rows][C], C in {32, 64, 128}: row-major, the 16-byte slots of a row permuted so that
// the 16 lanes a ds_read_b128 is served for -- 16 consecutive rows, same channel group -- hit 16 different slots:
//   C = 32  (128-byte rows: consecutive rows alternate between the two halves of the 64 banks): slot ^ ((row >> 1) & 7)
//   C >= 64 (rows a multiple of 256 bytes: every row starts at bank 0): slot16 ^ (row & 15) inside each 64-channel piece
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include <atomic>

#include "loss_device.h"
#include "sg_internal.h"

namespace sg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kFzThreads = 512, kFzWaves = 8;
constexpr int kFzHeadFloats = 4096;  // LDS scratch of the in-kernel head (fz_head): 3 x 1024 class slots + 872 small ones

__host__ __device__ constexpr int fz_min(int a, int b) { return a < b ? a : b; }
__host__ __device__ constexpr int fz_max(int a, int b) { return a > b ? a : b; }

template <int C>
__device__ __forceinline__ int fz_off(int R, int c) {
    if constexpr (C == 32) return R * 32 + ((((c >> 2) ^ (R >> 1)) & 7) << 2) + (c & 3);
    else return R * C + (c & ~63) + ((((c >> 2) ^ R) & 15) << 2) + (c & 3);
}

// ---------------------------------------------------------------------------------------------------------------
// Slices.  c0 .. c1: the slice's rows of conv8's output; every other tensor is cut at c * (product of the pool strides
// above it), the last slice taking the ragged rest -- "own" ranges: what the slice writes to memory (forward) / the rows of
// d features it produces (backward).  The ranges it has to COMPUTE follow from the taps and pools in between.
struct AnSlice {
    int olo[kAnConv], ohi[kAnConv];  // forward: rows of act[l] to compute | backward: rows of d act[l] needed
    int wlo[kAnConv], whi[kAnConv];  // own rows of act[l]
    int plo, phi, wplo, wphi;        // the same for the pre-filter output / its gradient
    int flo, fhi;                    // forward: feature rows to read | backward: own rows of d features
};

__host__ __device__ inline void an_slice_own(const int* Tout, int Fnet, int S, int s, AnSlice& r) {
    const int T8 = Tout[kAnConv - 1];
    const int c0 = (int)((long long)s * T8 / S), c1 = (int)((long long)(s + 1) * T8 / S);
    const bool last = s == S - 1;
    int m = 1;
    for (int l = kAnConv - 1; l >= 0; --l) {
        if (kAnPool[l]) m *= 2;
        r.wlo[l] = fz_min(c0 * m, Tout[l]);
        r.whi[l] = last ? Tout[l] : fz_min(c1 * m, Tout[l]);
    }
    r.wplo = fz_min(c0 * m, Fnet);
    r.wphi = last ? Fnet : fz_min(c1 * m, Fnet);
}

__host__ __device__ inline void an_slice_fwd(const int* Tin, const int* Tout, int Fnet, int S, int s, AnSlice& r) {
    an_slice_own(Tout, Fnet, S, s, r);
    int lo = r.wlo[kAnConv - 1], hi = r.whi[kAnConv - 1];
    for (int l = kAnConv - 1; l >= 0; --l) {
        r.olo[l] = lo;
        r.ohi[l] = hi;
        const int ilo = fz_max(0, lo - kAnPad[l]), ihi = fz_min(Tin[l], hi - kAnPad[l] + 2);  // valid input rows of conv l
        if (l > 0) {
            const int nlo = kAnPool[l - 1] ? 2 * ilo : ilo, nhi = kAnPool[l - 1] ? fz_min(2 * ihi, Tout[l - 1]) : ihi;
            lo = fz_min(nlo, r.wlo[l - 1]);
            hi = fz_max(nhi, r.whi[l - 1]);
        } else {
            r.plo = fz_min(ilo, r.wplo);
            r.phi = fz_max(ihi, r.wphi);
            r.flo = fz_max(0, r.plo - 2);
            r.fhi = fz_min(Fnet, r.phi + 2);
        }
    }
}

// backward: own rows of d features [flo, fhi) -> rows of d pre, then bottom-up the rows of d act[l] every data gradient
// needs (d in[t] = sum_j W_j^T d out[t + pad - j]) and through the un-pooling (pooled row p <- rows 2p, 2p + 1)
__host__ __device__ inline void an_slice_bwd(const int* Tin, const int* Tout, int Fnet, int S, int s, AnSlice& r) {
    an_slice_own(Tout, Fnet, S, s, r);
    r.flo = r.wplo;
    r.fhi = r.wphi;
    r.plo = fz_max(0, r.flo - 2);
    r.phi = fz_min(Fnet, r.fhi + 2);
    int lo = r.plo, hi = r.phi;  // rows of the INPUT of conv l whose gradient is needed
    for (int l = 0; l < kAnConv; ++l) {
        r.wlo[l] = lo;  // (re-used: rows of d input-of-l this slice produces)
        r.whi[l] = hi;
        r.olo[l] = fz_max(0, lo + kAnPad[l] - 2);
        r.ohi[l] = fz_min(Tout[l], hi + kAnPad[l]);
        if (l + 1 < kAnConv) {
            lo = kAnPool[l] ? r.olo[l] / 2 : r.olo[l];
            hi = kAnPool[l] ? fz_min((r.ohi[l] + 1) / 2, Tin[l + 1]) : r.ohi[l];
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// acc[mi] += the 32 x 32 output block (rows R0[mi] + lane % 32 of the LDS input for tap offset 0, columns n0 + lane % 32)
// over 3 taps x K channels.  REV: tap j reads input row + (2 - j) (data gradient), else row + j.
template <int K, int MI, bool REV, bool RING>
__device__ __forceinline__ void fz_mac(const float* __restrict__ in, const float4* __restrict__ wl, int ldw, const int (&R0)[MI],
                                       f32x16 (&acc)[MI]) {
    constexpr int NCH = 3 * K / 32, CPT = K / 32;  // chunks of 32 k values; per tap
    const int lhi = (threadIdx.x & 63) >> 5;
    if constexpr (RING) {
        // Small slices (one 32-row tile per wave): a chunk is 16 MFMAs = 0.43 us, an L2 round trip under load 1-2 us, so
        // the W operands of chunk c sit in slot c % D of a register ring and are requested D chunks ahead; loops fully
        // unrolled (3 .. 12 chunks), every slot a compile-time register.  A/B on one box at 64 utterances (S = 4): 175 us
        // per pass pair against 191 with the one-chunk-ahead form below.  It takes ~240 registers (hipcc hoists the LDS
        // addresses of all chunks), which is why this variant of the kernels keeps the compact, fully checked epilogue.
        constexpr int D = MI == 3 ? 3 : MI == 2 ? 4 : 6;
        constexpr int DD = D < NCH ? D : NCH;
        float4 wr[DD][4];
#pragma unroll
        for (int d = 0; d < DD; ++d)
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) wr[d][kg] = wl[(size_t)(d * 8 + 2 * kg) * ldw];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int j = ch / CPT, kc = ch % CPT;
            float4 a[4][MI];
#pragma unroll
            for (int kg = 0; kg < 4; ++kg)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    a[kg][mi] = *reinterpret_cast<const float4*>(in + fz_off<K>(R0[mi] + (REV ? 2 - j : j), kc * 32 + 8 * kg + 4 * lhi));
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                const float4 w = wr[ch % DD][kg];
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const float wb = st == 0 ? w.x : st == 1 ? w.y : st == 2 ? w.z : w.w;
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const float4 x = a[kg][mi];
                        const float xa = st == 0 ? x.x : st == 1 ? x.y : st == 2 ? x.z : x.w;
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, wb, acc[mi], 0, 0, 0);
                    }
                }
            }
            if (ch + DD < NCH) {
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) wr[ch % DD][kg] = wl[(size_t)((ch + DD) * 8 + 2 * kg) * ldw];
            }
        }
    } else {
        // Two or three tiles per wave (whole utterances, or halves): a chunk is 32-48 MFMAs per wave, the next chunk's four
        // 16-byte W pieces requested before this chunk's multiplications are in time, and the ~130 registers of this form
        // leave room for the epilogue's straight paths (A/B at 512 utterances: 685 us per pass pair against 720).
        float4 wc[4], wn[4];
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) wc[kg] = wl[(size_t)(2 * kg) * ldw];
#pragma unroll 1
        for (int j = 0; j < 3; ++j) {
            int Rj[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) Rj[mi] = R0[mi] + (REV ? 2 - j : j);
#pragma unroll
            for (int kc = 0; kc < CPT; ++kc) {
                const int ch = j * CPT + kc;
                const int nx = ch + 1 < NCH ? ch + 1 : ch;  // past the end: the last chunk again (valid memory, never multiplied)
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) wn[kg] = wl[(size_t)(nx * 8 + 2 * kg) * ldw];
                float4 a[4][MI];
#pragma unroll
                for (int kg = 0; kg < 4; ++kg)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) a[kg][mi] = *reinterpret_cast<const float4*>(in + fz_off<K>(Rj[mi], kc * 32 + 8 * kg + 4 * lhi));
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
                    const float4 w = wc[kg];
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const float wb = st == 0 ? w.x : st == 1 ? w.y : st == 2 ? w.z : w.w;
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) {
                            const float4 x = a[kg][mi];
                            const float xa = st == 0 ? x.x : st == 1 ? x.y : st == 2 ? x.z : x.w;
                            acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, wb, acc[mi], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) wc[kg] = wn[kg];
            }
        }
    }
}

// The same contraction on 16 x 16 output blocks (v_mfma_f32_16x16x4_f32), for layers so thin that 32 x 32 blocks leave
// most of the block's waves without work (small time slices: conv6..conv8 of a quarter utterance have 1-4 such blocks for
// 8 waves; a single utterance cut into 16 slices is thin everywhere).  Four times as many units, each with a quarter of
// the dependent-MFMA chain: the k values an instruction consumes are its four 16-lane groups in order, so feeding lane
// group g the k values (0, 4, 1, 5)[g] and then (2, 6, 3, 7)[g] of a k-group reproduces the order -- and the bits -- of the
// 32 x 32 x 2 form (tools/native/mfma_order.hip; conv_gemm_s16_kernel does the same for the TDNN at batch 1-4).  Lane
// (i = lane % 16, g = lane / 16): A row R0 = tile row i, 16-byte piece 2 kg + (g & 1) of the k-group, components g >> 1 and
// 2 + (g >> 1); W column i of the block, the same piece and components.  W ring six chunks deep (a chunk is 8 MFMAs).
// Measured (one box, tools/audionet_cnn_bench.py / tools/an_trace.py): 245 -> 235 us per pass pair at 128 rows (S = 2, the
// configs[3] shard with EOT 2), no change at 64 rows (S = 4) and for a single utterance (S = 16: 52 us per launch either
// way).  Why not more: a 16-column block streams the same W rows as a 32-column one for half the outputs, and lane groups
// g and g + 2 fetch the same 16-byte pieces -- four times the W bytes per output through the CU's L1 (786 KB per layer and
// block at K = 384: 5 us at 64 B/clk, against 1.8 us for the shorter MFMA chain it buys); requesting all of a unit's W up
// front instead of six chunks ahead changed nothing.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int K, bool REV>
__device__ __forceinline__ void fz_mac16(const float* __restrict__ in, const float4* __restrict__ wl, int ldw, int R0, f32x4v& acc) {
    constexpr int NCH = 3 * K / 32, CPT = K / 32;
    const int g = (threadIdx.x & 63) >> 4, g1 = g & 1;
    const bool hi = (g >> 1) != 0;
    constexpr int D = NCH < 6 ? NCH : 6;
    float4 wr[D][4];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) wr[d][kg] = wl[(size_t)(d * 8 + 2 * kg) * ldw];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int j = ch / CPT, kc = ch % CPT;
        float4 a[4];
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) a[kg] = *reinterpret_cast<const float4*>(in + fz_off<K>(R0 + (REV ? 2 - j : j), kc * 32 + 8 * kg + 4 * g1));
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            const float4 w = wr[ch % D][kg], x = a[kg];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hi ? x.y : x.x, hi ? w.y : w.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hi ? x.w : x.z, hi ? w.w : w.z, acc, 0, 0, 0);
        }
        if (ch + D < NCH) {
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) wr[ch % D][kg] = wl[(size_t)((ch + D) * 8 + 2 * kg) * ldw];
        }
    }
}

// row of accumulator element e inside its tile: 32 x 32 blocks (16 values per lane) / 16 x 16 blocks (4 values per lane)
template <int NV>
__device__ __forceinline__ constexpr int fz_ro(int e) { return NV == 16 ? (e & 3) + 8 * (e >> 2) : e; }

// A wave's share of a layer: output column block wn of N / 32, and a contiguous share of the m-tiles, in units of up to
// three 32-row tiles (one W operand stream feeds all of them).  epi(values per lane, tile's first row, tile height, the
// lane's first row, the lane's column, accumulator, pre's value) consumes a finished tile (rows relative to the layer's first).
// pre(column) runs before the unit's multiplications and its result is handed to epi: what the epilogue needs from memory (the
// bias of the wave's columns) is requested there, so that its latency -- 1-2 us from the L2, once per unit, 4.7 us per
// epilogue with the ReLU outputs' stores behind it in the first version -- passes under the MFMAs.
template <int K, int N, bool REV, bool SMALL, typename Pre, typename Epi>
__device__ __forceinline__ void fz_layer(const float* __restrict__ in, const float* __restrict__ wq, int n_out, Pre&& pre, Epi&& epi) {
    constexpr int NT = N / 32, G = kFzWaves / NT;
    static_assert(NT >= 1 && NT <= kFzWaves && kFzWaves % NT == 0, "column blocks must divide the waves");
    const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mt = (n_out + 31) >> 5;
    if (mt * NT * 2 <= kFzWaves) {
        // thin layer: at most half of the waves would get a 32 x 32 block -> 16 x 16 blocks, dealt round-robin
        constexpr int NT16 = N / 16;
        const int l15 = lane & 15, g = lane >> 4;
        const int units = ((n_out + 15) >> 4) * NT16;
        for (int u = wid; u < units; u += kFzWaves) {
            const int m16 = u / NT16, n16 = u - m16 * NT16;
            const int col = n16 * 16 + l15;
            const float4* wl = reinterpret_cast<const float4*>(wq) + (size_t)(g & 1) * N + col;
            f32x4v acc = {0.f, 0.f, 0.f, 0.f};
            const auto pv = pre(col);
            fz_mac16<K, REV>(in, wl, N, min(m16 * 16 + l15, n_out - 1), acc);
            epi(std::integral_constant<int, 4>{}, m16 * 16, 16, m16 * 16 + 4 * g, col, acc, pv);
        }
        return;
    }
    const int wn = wid % NT, wg = wid / NT;
    const int m_begin = wg * mt / G, m_end = (wg + 1) * mt / G;
    const int n0 = wn * 32, col = n0 + l31;
    const float4* wl = reinterpret_cast<const float4*>(wq) + (size_t)lhi * N + n0 + l31;
    constexpr std::integral_constant<int, 16> t16{};
    for (int m = m_begin; m < m_end; m += 3) {
        const int cnt = min(3, m_end - m);
        // rows past the layer's last output row are computed on a clamped (valid) input row and dropped by the epilogue
        if (cnt == 3) {
            const int R0[3] = {min(m * 32 + l31, n_out - 1), min(m * 32 + 32 + l31, n_out - 1), min(m * 32 + 64 + l31, n_out - 1)};
            f32x16 acc[3];
#pragma unroll
            for (int mi = 0; mi < 3; ++mi)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;
            const auto pv = pre(col);
            fz_mac<K, 3, REV, SMALL>(in, wl, N, R0, acc);
#pragma unroll
            for (int mi = 0; mi < 3; ++mi) epi(t16, (m + mi) * 32, 32, (m + mi) * 32 + 4 * lhi, col, acc[mi], pv);
        } else if (cnt == 2) {
            const int R0[2] = {min(m * 32 + l31, n_out - 1), min(m * 32 + 32 + l31, n_out - 1)};
            f32x16 acc[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;
            const auto pv = pre(col);
            fz_mac<K, 2, REV, SMALL>(in, wl, N, R0, acc);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) epi(t16, (m + mi) * 32, 32, (m + mi) * 32 + 4 * lhi, col, acc[mi], pv);
        } else {
            const int R0[1] = {min(m * 32 + l31, n_out - 1)};
            f32x16 acc[1];
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][e] = 0.f;
            const auto pv = pre(col);
            fz_mac<K, 1, REV, SMALL>(in, wl, N, R0, acc);
            epi(t16, m * 32, 32, m * 32 + 4 * lhi, col, acc[0], pv);
        }
    }
}

#define FZ_STAMP(i)                                                                                           \
    if (p.trace && threadIdx.x == 0) p.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = wall_clock64();

__device__ __forceinline__ void fz_zero(float* buf, int floats) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x * 4; i < floats; i += kFzThreads * 4) *reinterpret_cast<float4*>(buf + i) = z;
}

// ---------------------------------------------------------------------------------------------------------------
// forward layer L: input (Cin channels) in `in` -- LDS row 0 = virtual input row olo[L] - pad --, output to memory (own rows)
// and, for the next layer, to `out` (LDS row 0 = virtual input row olo[L + 1] - pad[L + 1]; rows outside the valid range
// stay zero: the convolution's zero padding)
template <int L, bool SMALL>
__device__ __forceinline__ void fz_fwd_layer(const AnFusedArgs& p, const AnSlice& r, int row, const float* in, float* out) {
    constexpr int CIN = kAnCin[L], COUT = kAnCout[L];
    constexpr bool POOL = kAnPool[L], LAST = L == kAnConv - 1;
    constexpr int CNEXT = LAST ? 32 : kAnCin[L + 1 < kAnConv ? L + 1 : L];
    const int olo = r.olo[L], n_out = r.ohi[L] - r.olo[L], Tout = p.Tout[L];
    int nbase = 0, nrows = 0;  // the next layer's LDS window in its virtual input rows
    if constexpr (!LAST) {
        nbase = r.olo[L + 1] - kAnPad[L + 1];
        nrows = r.ohi[L + 1] - r.olo[L + 1] + 2;
        fz_zero(out, nrows * CNEXT);
    }
    __syncthreads();
    if constexpr (L == 2) { FZ_STAMP(10) }
    float* act = p.act[L] + (size_t)row * Tout * COUT;
    float* pool = POOL ? p.pool[L] + (size_t)row * (Tout / 2) * COUT : nullptr;
    const float* bias = p.bias[L];
    // Epilogue cost is instruction count: the wave that shares the SIMD is still multiplying, and every VALU / branch
    // instruction here takes matrix-pipe issue slots from it (the first version -- per-element bounds checks, 64-bit
    // addresses, own-row tests: ~560 instructions per tile -- took 4.6 us per tile next to 10.4 us of MFMAs).  So: a tile
    // whose 32 rows are all valid and all inside the next layer's window (wave-uniform tests) takes the straight path --
    // one 32-bit offset per lane, compile-time row strides --, only ragged tiles check per element; and every computed
    // row goes to memory, halo rows included (they equal the neighbour slice's own rows bit for bit), which drops the
    // own-row tests.
    fz_layer<CIN, COUT, false, SMALL>(in, p.wq[L], n_out, [&](int col) __attribute__((always_inline)) { return bias[col]; },
                               [&](auto nvt, int m0, int th, int r0, int col, const auto& acc, float bv) __attribute__((always_inline)) {
        // nvt: accumulator values per lane (16: a 32 x 32 block, 4: a 16 x 16 block); m0 / th: the tile's first row and
        // height; r0: the lane's first row (rows relative to olo), its values are rows r0 + fz_ro(e)
        constexpr int NV = decltype(nvt)::value;
        if constexpr (L == 2) { FZ_STAMP(11) }
        float v[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) v[e] = fmaxf(acc[e] + bv, 0.f);
        const bool full = !SMALL && m0 + th <= n_out;  // (SMALL: the compact, fully checked form only -- see fz_mac)
        const unsigned g0 = (unsigned)((olo + r0) * COUT + col);
        if (full) {
#pragma unroll
            for (int e = 0; e < NV; ++e) act[g0 + fz_ro<NV>(e) * COUT] = v[e];
        } else {
#pragma unroll
            for (int e = 0; e < NV; ++e)
                if (r0 + fz_ro<NV>(e) < n_out) act[g0 + fz_ro<NV>(e) * COUT] = v[e];
        }
        if constexpr (!POOL && !LAST) {
            const int q0 = olo + r0 - nbase, qt = olo + m0 - nbase;  // lane's / tile's first row in the next window
            if (full && qt >= 0 && qt + th <= nrows) {
#pragma unroll
                for (int e = 0; e < NV; ++e) out[fz_off<CNEXT>(q0 + fz_ro<NV>(e), col)] = v[e];
            } else {
#pragma unroll
                for (int e = 0; e < NV; ++e) {
                    const int q = q0 + fz_ro<NV>(e);
                    if (r0 + fz_ro<NV>(e) < n_out && q >= 0 && q < nrows) out[fz_off<CNEXT>(q, col)] = v[e];
                }
            }
        }
        if constexpr (POOL) {  // rows t, t + 1 (olo, r0 and fz_ro(e) are even for even e): MaxPool1d(2)
            const int p0 = (olo + r0) >> 1, pt = (olo + m0) >> 1;  // lane's / tile's first pooled row
            const unsigned gp0 = (unsigned)(p0 * COUT + col);
            const bool pfull = full && pt + th / 2 <= Tout / 2;
            if (pfull) {
#pragma unroll
                for (int e = 0; e < NV; e += 2) pool[gp0 + (fz_ro<NV>(e) >> 1) * COUT] = fmaxf(v[e], v[e + 1]);
            } else {
#pragma unroll
                for (int e = 0; e < NV; e += 2)
                    if (r0 + fz_ro<NV>(e) + 1 < n_out && p0 + (fz_ro<NV>(e) >> 1) < Tout / 2) pool[gp0 + (fz_ro<NV>(e) >> 1) * COUT] = fmaxf(v[e], v[e + 1]);
            }
            if (pfull && pt - nbase >= 0 && pt - nbase + th / 2 <= nrows) {
#pragma unroll
                for (int e = 0; e < NV; e += 2) out[fz_off<CNEXT>(p0 - nbase + (fz_ro<NV>(e) >> 1), col)] = fmaxf(v[e], v[e + 1]);
            } else {
#pragma unroll
                for (int e = 0; e < NV; e += 2) {
                    const int pr = p0 + (fz_ro<NV>(e) >> 1), q = pr - nbase;
                    if (r0 + fz_ro<NV>(e) + 1 < n_out && pr < Tout / 2 && q >= 0 && q < nrows) out[fz_off<CNEXT>(q, col)] = fmaxf(v[e], v[e + 1]);
                }
            }
        }
    });
    if constexpr (L == 2) { FZ_STAMP(12) }
    __syncthreads();
}

template <bool SMALL>
__device__ __forceinline__ void fz_forward_body(const AnFusedArgs& p, int row, float* bufA, float* bufB) {
    FZ_STAMP(0)
    AnSlice r;
    an_slice_fwd(p.Tin, p.Tout, p.Fnet, p.S, blockIdx.x, r);
    // ---- 5x5 pre-filter: feature rows [flo, fhi) -> bufB (plain [rows][32]); pre rows [plo, phi) -> memory (own rows) and
    //      bufA as conv2's input window
    // The feature rows are staged with a ZERO border (two rows above and below the pre rows, two columns left and right:
    // [rows + 4][36]) so that the 25 taps are unconditional LDS reads -- a tap outside the image multiplies a staged zero,
    // which adds exactly what the guarded form `w * (ok ? v : 0)` of an_prefilter_kernel adds (the first version clamped
    // and selected per tap: 15 us of a 180 us block at 300 frames).
    const int F = p.Fnet, np = r.phi - r.plo;
    constexpr int PW = kAnMel + 4;
    fz_zero(bufB, ((np + 4) * PW + 3) & ~3);
    const int base0 = r.olo[0] - kAnPad[0], rows0 = r.ohi[0] - r.olo[0] + 2;
    fz_zero(bufA, rows0 * 32);
    __syncthreads();
    {
        // staged row q <-> feature row plo - 2 + q; rows outside [flo, fhi) = outside the image stay zero
        const float* src = p.feats + (size_t)row * F * 32;
        for (int i = threadIdx.x; i < (r.fhi - r.flo) * 8; i += kFzThreads) {
            const int t = r.flo + (i >> 3), c4 = (i & 7) * 4;
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)t * 32 + c4);
            float* d = bufB + (t - (r.plo - 2)) * PW + 2 + c4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    __syncthreads();
    {
        float w[25];
#pragma unroll
        for (int i = 0; i < 25; ++i) w[i] = p.w25[i];
        float* pre = p.pre + (size_t)row * F * 32;
        for (int idx = threadIdx.x; idx < np * 32; idx += kFzThreads) {
            const int q0 = idx >> 5, m = idx & 31, t = r.plo + q0;
            const float* wnd = bufB + q0 * PW + m;  // tap (i, j) = staged[q0 + j][m + i]
            float acc = p.pre_bias;
#pragma unroll
            for (int i = 0; i < 5; ++i)      // mel offset
#pragma unroll
                for (int j = 0; j < 5; ++j)  // time offset
                    acc = fmaf(w[i * 5 + j], wnd[j * PW + i], acc);  // explicit, like an_prefilter_kernel
            if (t >= r.wplo && t < r.wphi) pre[(size_t)t * 32 + m] = acc;
            const int q = t - base0;
            if (q >= 0 && q < rows0) bufA[fz_off<32>(q, m)] = acc;
        }
    }
    __syncthreads();
    FZ_STAMP(1)
    fz_fwd_layer<0, SMALL>(p, r, row, bufA, bufB);
    FZ_STAMP(2)
    fz_fwd_layer<1, SMALL>(p, r, row, bufB, bufA);
    FZ_STAMP(3)
    fz_fwd_layer<2, SMALL>(p, r, row, bufA, bufB);
    FZ_STAMP(4)
    fz_fwd_layer<3, SMALL>(p, r, row, bufB, bufA);
    FZ_STAMP(5)
    fz_fwd_layer<4, SMALL>(p, r, row, bufA, bufB);
    FZ_STAMP(6)
    fz_fwd_layer<5, SMALL>(p, r, row, bufB, bufA);
    FZ_STAMP(7)
    fz_fwd_layer<6, SMALL>(p, r, row, bufA, bufB);
    FZ_STAMP(8)
}

template <bool SMALL>
__global__ __launch_bounds__(kFzThreads, 1) void an_cnn_fwd_kernel(AnFusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) float fz_lds[];
    fz_forward_body<SMALL>(p, blockIdx.y, fz_lds, fz_lds + p.buf_floats);
}

// ---------------------------------------------------------------------------------------------------------------
// backward layer L: d act[L] (Cout channels) in `in` -- LDS row 0 = row olo[L] + pad - 2 ... i.e. the window of rows
// [wlo[L] + pad - 2, whi[L] + pad) of d act[L], zero outside the tensor --; the data gradient gives d (input of conv L)
// rows [wlo[L], whi[L]); un-pooled / masked with the forward activations it becomes d act[L - 1] in `out` (LDS window
// of the next data gradient), or d pre for L = 0.
template <int L, bool SMALL>
__device__ __forceinline__ void fz_bwd_layer(const AnFusedArgs& p, const AnSlice& r, int row, const float* in, float* out, const float* wq) {
    constexpr int K = kAnCout[L], N = kAnCin[L];
    constexpr bool FIRST = L == 0;
    constexpr bool INPOOL = !FIRST && kAnPool[L > 0 ? L - 1 : 0];  // conv L reads the pooled output of block L - 1
    const int ilo = r.wlo[L], n_out = r.whi[L] - r.wlo[L];          // rows of d input-of-L to produce
    // next window: rows [wlo[L-1] + pad[L-1] - 2, ...) of d act[L - 1]; for L = 0: d pre rows [plo, phi) plain window
    int nbase, nrows;
    if constexpr (FIRST) {
        nbase = r.plo;
        nrows = r.phi - r.plo;
    } else {
        nbase = r.wlo[L - 1] + kAnPad[L - 1] - 2;
        nrows = r.whi[L - 1] - r.wlo[L - 1] + 2;
    }
    constexpr int PW = kAnMel + 4;  // L = 0: d pre goes to a zero-bordered plain image [rows + 4][36] (see the forward pre-filter)
    fz_zero(out, FIRST ? ((nrows + 4) * PW + 3) & ~3 : nrows * N);
    __syncthreads();
    const int Tprev = FIRST ? p.Fnet : p.Tout[L > 0 ? L - 1 : 0];
    const float* aprev = FIRST ? nullptr : p.act[L > 0 ? L - 1 : 0] + (size_t)row * Tprev * N;
    // (epilogue structure as in the forward layer: wave-uniform straight path for whole tiles inside the next window, 32-bit
    // offsets, per-element checks only on ragged tiles)
    fz_layer<K, N, true, SMALL>(in, wq, n_out, [](int) __attribute__((always_inline)) { return 0; },
                         [&](auto nvt, int m0, int th, int r0, int col, const auto& acc, int) __attribute__((always_inline)) {
        constexpr int NV = decltype(nvt)::value;  // (interface: see the forward layer)
        const bool full = !SMALL && m0 + th <= n_out;  // all rows of the tile are rows of d input-of-L to produce (SMALL: checked form only)
        if constexpr (FIRST) {
            const int q0 = ilo + r0 - nbase;
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                const int q = q0 + fz_ro<NV>(e);
                if (r0 + fz_ro<NV>(e) < n_out && q >= 0 && q < nrows) out[(q + 2) * PW + 2 + col] = acc[e];
            }
        } else if constexpr (INPOOL) {
            // pooled row pr <- rows 2 pr (first maximum wins a tie, like torch) / 2 pr + 1 of act[L - 1], ReLU mask applied
            const int pr0 = ilo + r0, q0 = 2 * pr0 - nbase, qt = 2 * (ilo + m0) - nbase;
            float a0[NV], a1[NV];
            if (full && ilo + m0 + th <= Tprev / 2 && qt >= 0 && qt + 2 * th <= nrows) {
                const unsigned g0 = (unsigned)(2 * pr0 * N + col);
#pragma unroll
                for (int e = 0; e < NV; ++e) {
                    a0[e] = aprev[g0 + 2 * fz_ro<NV>(e) * N];
                    a1[e] = aprev[g0 + (2 * fz_ro<NV>(e) + 1) * N];
                }
#pragma unroll
                for (int e = 0; e < NV; ++e) {
                    const bool first = !(a1[e] > a0[e]);
                    out[fz_off<N>(q0 + 2 * fz_ro<NV>(e), col)] = (first && a0[e] > 0.f) ? acc[e] : 0.f;
                    out[fz_off<N>(q0 + 2 * fz_ro<NV>(e) + 1, col)] = (!first && a1[e] > 0.f) ? acc[e] : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < NV; ++e) {
                    const int pr = min(pr0 + fz_ro<NV>(e), Tprev / 2 - 1);
                    a0[e] = aprev[(unsigned)(2 * pr * N + col)];
                    a1[e] = aprev[(unsigned)((2 * pr + 1) * N + col)];
                }
#pragma unroll
                for (int e = 0; e < NV; ++e) {
                    const int pr = pr0 + fz_ro<NV>(e);
                    if (r0 + fz_ro<NV>(e) < n_out && pr < Tprev / 2) {
                        const bool first = !(a1[e] > a0[e]);
                        const float g0 = (first && a0[e] > 0.f) ? acc[e] : 0.f, g1 = (!first && a1[e] > 0.f) ? acc[e] : 0.f;
                        const int q = 2 * pr - nbase;
                        if (q >= 0 && q < nrows) out[fz_off<N>(q, col)] = g0;
                        if (q + 1 >= 0 && q + 1 < nrows) out[fz_off<N>(q + 1, col)] = g1;
                    }
                }
            }
        } else {
            const int t0 = ilo + r0, q0 = t0 - nbase, qt = ilo + m0 - nbase;
            float mk[NV];
            if (full && ilo + m0 + th <= Tprev && qt >= 0 && qt + th <= nrows) {
                const unsigned g0 = (unsigned)(t0 * N + col);
#pragma unroll
                for (int e = 0; e < NV; ++e) mk[e] = aprev[g0 + fz_ro<NV>(e) * N];
#pragma unroll
                for (int e = 0; e < NV; ++e) out[fz_off<N>(q0 + fz_ro<NV>(e), col)] = mk[e] > 0.f ? acc[e] : 0.f;
            } else {
#pragma unroll
                for (int e = 0; e < NV; ++e) mk[e] = aprev[(unsigned)(min(t0 + fz_ro<NV>(e), Tprev - 1) * N + col)];
#pragma unroll
                for (int e = 0; e < NV; ++e) {
                    const int q = q0 + fz_ro<NV>(e);
                    if (r0 + fz_ro<NV>(e) < n_out && q >= 0 && q < nrows) out[fz_off<N>(q, col)] = mk[e] > 0.f ? acc[e] : 0.f;
                }
            }
        }
    });
    __syncthreads();
}

// The head of the network for the block's utterance (AnHeadArgs): scratch in `scr` (>= 6 K floats of the idle LDS buffer), the
// window of d loss / d conv8 pre-activation rows [base, base + rows) -> `win` ([rows][32], fz_off<32>, zero outside the tensor).
// Phase by phase an_tail_kernel's arithmetic (k_audionet.hip) on the first 256 threads where that kernel's 256 matter for
// the order of a sum: the same bits.
__device__ __forceinline__ void fz_head(const AnFusedArgs& p, int row, bool writes, float* scr, float* win, int base, int rows) {
    const AnHeadArgs& h = p.head;
    const int tid = threadIdx.x, T8 = p.Tout[kAnConv - 1], S = h.S;
    float* sc = scr;                    // [1024]
    float* dsc = scr + 1024;            // [1024]
    float* ex = scr + 2048;             // [1024]
    float* bc = scr + 3072;             // [8]
    float* emb = scr + 3080;            // [32]
    int* arg = reinterpret_cast<int*>(scr + 3112);   // [32]
    float* demb = scr + 3144;           // [32]
    float* pmx = scr + 3176;            // [8][32]
    int* pat = reinterpret_cast<int*>(scr + 3432);   // [8][32]
    float* part = scr + 3688;           // [8][32]
    const float* a = p.act[kAnConv - 1] + (size_t)row * T8 * 32;
    fz_zero(win, rows * 32);
    if (tid < 256) {  // x.max(2): first maximum; eight time slices per channel, merged in time order with a strict >
        const int c = tid & 31, sl = tid >> 5;
        const int t0 = (int)((long long)T8 * sl / 8), t1 = (int)((long long)T8 * (sl + 1) / 8);
        float mx = -INFINITY;
        int at = t0;
        for (int t = t0; t < t1; ++t) {
            const float v = a[(size_t)t * 32 + c];
            if (v > mx) { mx = v; at = t; }
        }
        pmx[sl * 32 + c] = mx;
        pat[sl * 32 + c] = at;
    }
    __syncthreads();
    if (tid < 32) {
        float mx = pmx[tid];
        int at = pat[tid];
#pragma unroll
        for (int i = 1; i < 8; ++i)
            if (pmx[i * 32 + tid] > mx) { mx = pmx[i * 32 + tid]; at = pat[i * 32 + tid]; }
        emb[tid] = mx;
        arg[tid] = at;
        if (writes && h.emb_out) h.emb_out[(size_t)row * 32 + tid] = mx;
    }
    __syncthreads();
    for (int s = tid; s < S; s += kFzThreads) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) acc += h.fc_w[(size_t)s * 32 + c] * emb[c];
        acc += h.fc_b[s];
        sc[s] = acc;
        dsc[s] = 0.f;
        if (writes && h.scores_out) h.scores_out[(size_t)row * S + s] = acc;
    }
    __syncthreads();
    {
        int64_t dec = 0;
        const int64_t yy = h.y ? h.y[row] : 0;
        const float loss = loss_and_dscores_block(sc, dsc, ex, bc, S, h.threshold, yy, h.y != nullptr, h.ls, &dec, tid, kFzThreads,
                                                  h.ls.coef_dev ? h.ls.coef_dev + (size_t)(h.coef_rows > 0 ? row % h.coef_rows : row) * S : nullptr);
        if (writes && tid == 0) {
            if (h.dec_out) h.dec_out[row] = dec;
            if (h.dec_trace) h.dec_trace[row] = dec;
            if (h.y && h.success) h.success[row] = h.ls.targeted ? (dec == yy) : (dec != yy);
            if (h.loss_out) h.loss_out[row] = loss;
            if (h.loss_trace) h.loss_trace[row] = loss;
        }
    }
    __syncthreads();
    if (tid < 256) {  // d emb[c] = sum_s dsc[s] fc_w[s][c]: 8 class-strided partial sums per channel, combined in order
        const int c = tid & 31, q = tid >> 5;
        float acc = 0.f;
        for (int s = q; s < S; s += 8) acc += dsc[s] * h.fc_w[(size_t)s * 32 + c];
        part[q * 32 + c] = acc;
    }
    __syncthreads();
    if (tid < 32) {
        float r = part[tid];
#pragma unroll
        for (int i = 1; i < 8; ++i) r += part[i * 32 + tid];
        demb[tid] = r;
    }
    __syncthreads();
    // gradient wrt the pre-activation of conv8: only the arg-max frame of each channel, if it is > 0
    for (int idx = tid; idx < rows * 32; idx += kFzThreads) {
        const int q = idx >> 5, c = idx & 31, t = base + q;
        if (t >= 0 && t < T8) win[fz_off<32>(q, c)] = (t == arg[c] && emb[c] > 0.f) ? demb[c] : 0.f;
    }
    __syncthreads();
}

// BOTH: the backward half of the one-launch form (weights from p.wq_bwd, no stage stamps: the forward's own them)
template <bool SMALL, bool HEAD, bool BOTH>
__device__ __forceinline__ void fz_backward_body(const AnFusedArgs& p, int row, float* bufA, float* bufB) {
#define FZ_BSTAMP(i) if constexpr (!BOTH) { FZ_STAMP(i) }
    FZ_BSTAMP(0)
    AnSlice r;
    an_slice_bwd(p.Tin, p.Tout, p.Fnet, p.S, blockIdx.x, r);
    // ---- d act[6] window: rows [wlo[6] + pad - 2, whi[6] + pad) of (T8, 32), zero outside -- from memory, or from the head
    {
        constexpr int L = kAnConv - 1;
        const int T8 = p.Tout[L], base = r.wlo[L] + kAnPad[L] - 2, rows = r.whi[L] - r.wlo[L] + 2;
        if constexpr (HEAD) {
            fz_head(p, row, blockIdx.x == 0, bufB, bufA, base, rows);
        } else {
            fz_zero(bufA, rows * 32);
            __syncthreads();
            const float* src = p.dtop + (size_t)row * T8 * 32;
            for (int idx = threadIdx.x; idx < rows * 32; idx += kFzThreads) {
                const int q = idx >> 5, c = idx & 31, t = base + q;
                if (t >= 0 && t < T8) bufA[fz_off<32>(q, c)] = src[(size_t)t * 32 + c];
            }
            __syncthreads();
        }
    }
    const float* const* wq = BOTH ? p.wq_bwd : p.wq;
    FZ_BSTAMP(1)
    fz_bwd_layer<6, SMALL>(p, r, row, bufA, bufB, wq[6]);
    FZ_BSTAMP(2)
    fz_bwd_layer<5, SMALL>(p, r, row, bufB, bufA, wq[5]);
    FZ_BSTAMP(3)
    fz_bwd_layer<4, SMALL>(p, r, row, bufA, bufB, wq[4]);
    FZ_BSTAMP(4)
    fz_bwd_layer<3, SMALL>(p, r, row, bufB, bufA, wq[3]);
    FZ_BSTAMP(5)
    fz_bwd_layer<2, SMALL>(p, r, row, bufA, bufB, wq[2]);
    FZ_BSTAMP(6)
    fz_bwd_layer<1, SMALL>(p, r, row, bufB, bufA, wq[1]);
    FZ_BSTAMP(7)
    fz_bwd_layer<0, SMALL>(p, r, row, bufA, bufB, wq[0]);
    FZ_BSTAMP(8)
    // ---- transposed 5x5 pre-filter: d pre rows [plo, phi) in bufB (zero-bordered plain image) -> own rows of d features
    {
        float w[25];
#pragma unroll
        for (int i = 0; i < 25; ++i) w[i] = p.w25[i];
        constexpr int PW = kAnMel + 4;
        const int F = p.Fnet;
        float* dst = p.dfeats + (size_t)row * F * 32;
        for (int idx = threadIdx.x; idx < (r.fhi - r.flo) * 32; idx += kFzThreads) {
            const int t = r.flo + (idx >> 5), m = idx & 31;
            // d pre[t - j + 2][m - i + 2] = staged[(t - j + 2) - plo + 2][(m - i + 2) + 2]
            const float* wnd = bufB + (t - r.plo + 4) * PW + m + 4;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) acc = fmaf(w[i * 5 + j], wnd[-j * PW - i], acc);
            dst[(size_t)t * 32 + m] = acc;
        }
    }
    FZ_BSTAMP(9)
#undef FZ_BSTAMP
}

template <bool SMALL, bool HEAD>
__global__ __launch_bounds__(kFzThreads, 1) void an_cnn_bwd_kernel(AnFusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) float fz_lds[];
    fz_backward_body<SMALL, HEAD, false>(p, blockIdx.y, fz_lds, fz_lds + p.buf_floats);
}

// Forward, head and backward of one whole utterance per block in ONE launch (S = 1: no other block holds rows of the
// utterance, so nothing crosses a block).  The backward reads the activations the forward wrote a moment ago through this
// CU's own L1 / the XCD's L2 (its masks); between the two halves the block's stores are complete (vmcnt at the barrier) and
// no line of them was cached here before the forward wrote it.
__global__ __launch_bounds__(kFzThreads, 1) void an_cnn_fwdbwd_kernel(AnFusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) float fz_lds[];
    fz_forward_body<false>(p, blockIdx.y, fz_lds, fz_lds + p.buf_floats);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    fz_backward_body<false, true, true>(p, blockIdx.y, fz_lds, fz_lds + p.buf_floats);
}

// ---------------------------------------------------------------------------------------------------------------
// Host side: slices per utterance and LDS demand.
static int fz_plan(const int* Tin, const int* Tout, int Fnet, int rows, int num_cus, int* S_out, int* buf_floats_out) {
    const int T8 = Tout[kAnConv - 1];
    if (T8 < 1) return -1;
    const int cus = num_cus > 0 ? num_cus : 256;
    auto demand = [&](int S) {
        int need = 0;
        for (int s = 0; s < S; ++s) {
            AnSlice f, b;
            an_slice_fwd(Tin, Tout, Fnet, S, s, f);
            an_slice_bwd(Tin, Tout, Fnet, S, s, b);
            need = std::max(need, (f.phi - f.plo + 4) * (kAnMel + 4) + 4);  // zero-bordered staging of the pre-filter's input
            need = std::max(need, (b.phi - b.plo + 4) * (kAnMel + 4) + 4);  // ... and of its transpose
            for (int l = 0; l < kAnConv; ++l) {
                need = std::max(need, (f.ohi[l] - f.olo[l] + 2) * kAnCin[l]);   // forward input window of conv l
                need = std::max(need, (b.whi[l] - b.wlo[l] + 2) * kAnCout[l]);  // backward: d act[l] window
            }
        }
        return (need + 3) & ~3;
    };
    // cost model: rounds of blocks on the chip x time of a block, in conv8 rows: its share + 12 -- a block's time is not
    // proportional to its rows (tools/an_trace.py, forward: 51 us at 2.2 rows, 79 us at 8.75, 163 us at 35: ~42 us + 3.6 us
    // per row: halo, the layers' barriers and epilogues, the weight stream's start).  Until round 5 the constant was 4 and
    // 129..255 utterances were cut too finely (192 utterances: 0.77 ms per PGD step against 0.67 at 256).
    int best = 0;
    double best_cost = 0.0;
    const int smax = std::min(T8, 16);
    constexpr int kLdsBudget = 160 * 1024 - 512;
    for (int S = 1; S <= smax; ++S) {
        if ((size_t)demand(S) * 2 * sizeof(float) > (size_t)kLdsBudget) continue;
        const long blocks = (long)rows * S;
        const double cost = (double)((blocks + cus - 1) / cus) * ((double)T8 / S + 12.0);
        if (!best || cost < best_cost - 1e-9) {
            best = S;
            best_cost = cost;
        }
    }
    if (!best) return -1;  // even the finest cut does not fit (very long utterances): the per-layer path takes over
    *S_out = best;
    *buf_floats_out = demand(best);
    return 0;
}

// The kernels ask for up to 160 KB of dynamic LDS (gfx950's per-CU LDS; the Makefile's ARCH is overridable): the opt-in is
// made once per DEVICE, and a device that refuses it (less LDS) runs the per-layer sequence instead of failing every call.
static bool fz_device_ok() {
    static std::atomic<int> state[kMaxDevices];  // 0 unknown, 1 opted in, -1 refused
    const int d = sg_device_slot();
    int st = state[d].load(std::memory_order_relaxed);
    if (st == 0) {
        const void* fns[7] = {reinterpret_cast<const void*>(an_cnn_fwd_kernel<false>), reinterpret_cast<const void*>(an_cnn_fwd_kernel<true>),
                              reinterpret_cast<const void*>(an_cnn_bwd_kernel<false, false>), reinterpret_cast<const void*>(an_cnn_bwd_kernel<true, false>),
                              reinterpret_cast<const void*>(an_cnn_bwd_kernel<false, true>), reinterpret_cast<const void*>(an_cnn_bwd_kernel<true, true>),
                              reinterpret_cast<const void*>(an_cnn_fwdbwd_kernel)};
        st = 1;
        for (const void* fn : fns)
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) st = -1;
        if (st < 0) (void)hipGetLastError();  // the refusal is handled here: the per-layer path runs
        state[d].store(st, std::memory_order_relaxed);
    }
    return st > 0;
}

bool an_fused_supported(const int* Tin, const int* Tout, int Fnet, int rows, int num_cus) {
    int S, bf;
    if (rows < 1 || rows > 65535) return false;  // gridDim.y = rows
    return fz_plan(Tin, Tout, Fnet, rows, num_cus, &S, &bf) == 0 && fz_device_ok();
}

hipError_t launch_an_cnn_fused(AnFusedArgs a, int rows, int num_cus, bool backward, int force_slices, hipStream_t s) {
    int S = 0, bf = 0;
    if (fz_plan(a.Tin, a.Tout, a.Fnet, rows, num_cus, &S, &bf) != 0) return hipErrorNotSupported;
    if (force_slices > 0) {  // tests: the result must not depend on the cut
        const int keep = S;
        S = std::min(force_slices, a.Tout[kAnConv - 1]);
        int need = 0;
        for (int sl = 0; sl < S; ++sl) {
            AnSlice f, b;
            an_slice_fwd(a.Tin, a.Tout, a.Fnet, S, sl, f);
            an_slice_bwd(a.Tin, a.Tout, a.Fnet, S, sl, b);
            need = std::max(need, std::max((f.phi - f.plo + 4) * (kAnMel + 4) + 4, (b.phi - b.plo + 4) * (kAnMel + 4) + 4));
            for (int l = 0; l < kAnConv; ++l)
                need = std::max(need, std::max((f.ohi[l] - f.olo[l] + 2) * kAnCin[l], (b.whi[l] - b.wlo[l] + 2) * kAnCout[l]));
        }
        bf = (need + 3) & ~3;
        if ((size_t)bf * 2 * sizeof(float) > 160 * 1024 - 512) {
            S = keep;
            if (fz_plan(a.Tin, a.Tout, a.Fnet, rows, num_cus, &S, &bf) != 0) return hipErrorNotSupported;
        }
    }
    if (backward && a.head.on && bf < kFzHeadFloats) bf = kFzHeadFloats;  // the head's scratch lives in the second buffer
    a.S = S;
    a.buf_floats = bf;
    const size_t lds = (size_t)bf * 2 * sizeof(float);
    if (rows > 65535 || !fz_device_ok()) return hipErrorNotSupported;
    // two builds of each kernel (fz_mac): small slices -- three or more per utterance, one 32-row tile per wave -- take the
    // deep W ring, whole / half utterances the compact multiply loop with the straight-path epilogues.  Same bits.
    const bool small = S >= 3;
    static const char* trace_file = sg_tune_env("SG_AN_TRACE");  // tuning aid: dump the per-block stage timestamps of every launch
    static PerDeviceScratch trace_buf;
    const size_t nblk = (size_t)S * rows;
    a.trace = trace_file ? static_cast<unsigned long long*>(trace_buf.get(nblk * 16 * 8)) : nullptr;
    if (a.trace) (void)hipMemsetAsync(a.trace, 0, nblk * 16 * 8, s);
    if (backward && a.head.on) {
        if (small) hipLaunchKernelGGL((an_cnn_bwd_kernel<true, true>), dim3(S, rows), dim3(kFzThreads), lds, s, a);
        else hipLaunchKernelGGL((an_cnn_bwd_kernel<false, true>), dim3(S, rows), dim3(kFzThreads), lds, s, a);
    } else if (backward) {
        if (small) hipLaunchKernelGGL((an_cnn_bwd_kernel<true, false>), dim3(S, rows), dim3(kFzThreads), lds, s, a);
        else hipLaunchKernelGGL((an_cnn_bwd_kernel<false, false>), dim3(S, rows), dim3(kFzThreads), lds, s, a);
    } else {
        if (small) hipLaunchKernelGGL(an_cnn_fwd_kernel<true>, dim3(S, rows), dim3(kFzThreads), lds, s, a);
        else hipLaunchKernelGGL(an_cnn_fwd_kernel<false>, dim3(S, rows), dim3(kFzThreads), lds, s, a);
    }
    if (a.trace) {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> h(nblk * 16);
        (void)hipMemcpy(h.data(), a.trace, h.size() * 8, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(trace_file, "a")) {
            fprintf(f, "%s S=%d rows=%d\n", backward ? "bwd" : "fwd", S, rows);
            for (size_t b = 0; b < nblk; ++b) {
                for (int i = 0; i < 13; ++i) fprintf(f, "%llu ", h[b * 16 + i]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    return hipGetLastError();
}

// LDS floats per buffer that whole utterances (S = 1) need; 0: they do not fit
static int fz_whole_utterance_floats(const int* Tin, const int* Tout, int Fnet) {
    AnSlice f, b;
    an_slice_fwd(Tin, Tout, Fnet, 1, 0, f);
    an_slice_bwd(Tin, Tout, Fnet, 1, 0, b);
    int need = std::max((f.phi - f.plo + 4) * (kAnMel + 4) + 4, (b.phi - b.plo + 4) * (kAnMel + 4) + 4);
    for (int l = 0; l < kAnConv; ++l)
        need = std::max(need, std::max((f.ohi[l] - f.olo[l] + 2) * kAnCin[l], (b.whi[l] - b.wlo[l] + 2) * kAnCout[l]));
    need = (need + 3) & ~3;
    return (size_t)need * 2 * sizeof(float) > 160 * 1024 - 512 ? 0 : need;
}

// the cut the fused launches will use: the planner's, or `force_slices` (tests) when that fits; 0: not supported
int an_fused_slices(const int* Tin, const int* Tout, int Fnet, int rows, int num_cus, int force_slices) {
    int S = 0, bf = 0;
    if (rows < 1 || rows > 65535 || Tout[kAnConv - 1] < 1 || !fz_device_ok()) return 0;
    if (force_slices == 1 && fz_whole_utterance_floats(Tin, Tout, Fnet) > 0) return 1;
    if (fz_plan(Tin, Tout, Fnet, rows, num_cus, &S, &bf) != 0) return 0;
    return force_slices > 1 ? std::min(force_slices, Tout[kAnConv - 1]) : S;
}

hipError_t launch_an_cnn_fwdbwd(AnFusedArgs a, int rows, int num_cus, int force_slices, hipStream_t s) {
    if (an_fused_slices(a.Tin, a.Tout, a.Fnet, rows, num_cus, force_slices) != 1) return hipErrorNotSupported;
    int bf = fz_whole_utterance_floats(a.Tin, a.Tout, a.Fnet);
    if (bf <= 0 || !a.head.on) return hipErrorNotSupported;
    if (bf < kFzHeadFloats) bf = kFzHeadFloats;
    a.S = 1;
    a.buf_floats = bf;
    a.trace = nullptr;
    hipLaunchKernelGGL(an_cnn_fwdbwd_kernel, dim3(1, rows), dim3(kFzThreads), (size_t)bf * 2 * sizeof(float), s, a);
    return hipGetLastError();
}

}  // namespace sg
