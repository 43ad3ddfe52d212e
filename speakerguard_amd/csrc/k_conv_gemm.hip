// Implicit-GEMM dilated 1-D convolution on f32-input MFMA (gfx950).
//
// Replaces torch.nn.functional.conv1d + autograd's conv data-gradient for the x-vector TDNN
// (reference model/_xv_plda/xvecTDNN.py:16-33,49-53; backward = the loss.backward() call at
// adaptive_attack/EOT.py:35 restricted to d/d-input -- weight gradients are never formed).
//
// Activations are channel-last ("frame-major"): row r = b*T + t holds the C channels of frame
// t of utterance b.  For output row (b, t) and tap j the K-slice is the contiguous channel
// vector of input row (b, t + j*tap_step), so no im2col buffer exists: the A tile of a K-chunk
// is BM rows x 32 channels fetched straight from the activation tensor with one row offset per
// tap.  The same kernel computes the data gradient by running over d(out) with tap_step = -dil,
// zero rows where t + j*tap_step falls outside the utterance, and tap-transposed weights.
//
//   C[r][n] = epi( sum_j sum_c A[row(r) + j*tap_step][c] * W[j*Kc + c][n] )
//
// Kernels in this file (all share one k order, so every launch strategy gives bit-identical results):
//   * gemm_segment<BM,BN,WM,WN> + conv_gemm_kernel: 4 waves, one block per output tile (64x128 or 128x32), operands
//     fed by ds_read_b32 from a K-major LDS image.  Used for tdnn1 (short K), fc1, split-K launches, small batches
//     and the AudioNet stack.
//   * gemm_segment8: 8 waves, 128x128, role-split staging, b32-fed (stream-K fallback when no packed weights exist).
//   * gemm_segment_q<WMW>: quad-fed 128x128 (8 waves) or 256x128 (16 waves, the default of the TDNN layers):
//     ds_read_b128 operands, k4-packed weights, buffer-load staging with hardware zero fill.
//   * conv_gemm_streamk_kernel: persistent blocks with equal (tile, K-chunk) ranges and a bit-exact hand-off of
//     split tiles.
// v_mfma_f32_32x32x2_f32 is exact fp32 (an fmaf chain), 64 cycles per instruction per SIMD.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

#include "sg_internal.h"

namespace sg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;

// ------------------------------------------------------------------------------------------
// One K-segment of one output tile: acc += A[m0.., chunks c_begin..c_end) * W[.., n0..).
// Ends with a barrier, so the LDS stages can be reused immediately by the caller.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void gemm_segment(const ConvGemmArgs& p, float* smem, int m0, int n0, int c_begin,
                                             int c_end, f32x16 (&acc)[BM / WM / 32][BN / WN / 32]) {
    constexpr int MI = BM / WM / 32;
    constexpr int NI = BN / WN / 32;
    constexpr int A_QUADS = BM / 4;            // row quads in the A tile
    constexpr int B_F4 = BK * BN / 4;          // float4 in the W tile
    constexpr int B_PER_THREAD = (B_F4 + 255) / 256;
    static_assert(A_QUADS * 8 <= 256, "A tile too tall");
    static_assert(B_PER_THREAD <= 4, "W tile too wide");

    float* As = smem;                 // [2][BK][BM]
    float* Bs = smem + 2 * BK * BM;   // [2][BK][BN]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- A staging map: lane -> (row quad, 4-float k group); 8 lanes of a group share a k group
    const int a_r4 = wid * 8 + (lane & 7);   // row quad index (valid if < A_QUADS)
    const int a_c4 = lane >> 3;              // which float4 of the 32-float slice
    const bool a_on = a_r4 < A_QUADS;
    const float* a_ptr[4];
    int a_t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = m0 + a_r4 * 4 + i;
        if (a_on && r < p.M) {
            const int b = r / p.Tc;
            const int t = r - b * p.Tc;
            a_ptr[i] = p.A + (size_t)(b * p.Ta + t) * p.lda + a_c4 * 4;
            a_t[i] = t;
        } else {
            a_ptr[i] = p.A;
            a_t[i] = -(1 << 28);
        }
    }
    const int kchunks = p.Kc / BK;
    // W tile: thread -> float4 f = tid + i*256 of the [BK][BN] chunk image
    const float* b_ptr[B_PER_THREAD];
#pragma unroll
    for (int i = 0; i < B_PER_THREAD; ++i) {
        const int f = (tid + i * 256) % B_F4;
        b_ptr[i] = p.W + (size_t)(f / (BN / 4)) * p.ldw + n0 + (f % (BN / 4)) * 4;
    }

    // Staging registers are named scalars on purpose: as arrays captured by the helper lambdas hipcc's
    // promote-alloca pass moved the W-tile registers into LDS (+16 KB, an LDS round trip per chunk).
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    bool ok0, ok1, ok2, ok3;
    rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);

    // Loads are branch-free: an out-of-range tap row (or a thread that stages nothing) reads the first
    // row and is zeroed when the registers are written to LDS.  A conditional load makes hipcc wait
    // vmcnt(0) at every branch join, which serialised four global round trips per chunk in front of
    // the MFMAs.  Per chunk only a wave-uniform element offset is added to per-thread base pointers.
#define SG_LOAD_A(i, r, okv)                                                        \
    {                                                                               \
        const int tt = a_t[i] + off;                                                \
        okv = tt >= 0 && tt < p.Ta;                                                 \
        r = *reinterpret_cast<const float4*>(okv ? a_ptr[i] + a_off : p.A);         \
    }
#define SG_LOAD_B(i, r) \
    if (i < B_PER_THREAD) r = *reinterpret_cast<const float4*>(b_ptr[i < B_PER_THREAD ? i : 0] + b_off);
#define SG_STORE_B(i, r)                                                            \
    if (i < B_PER_THREAD && (B_F4 % 256 == 0 || tid + i * 256 < B_F4))              \
        *reinterpret_cast<float4*>(Bs + buf * BK * BN + (tid + i * 256) * 4) = r;

    auto load_chunk = [&](int c) {
        const int j = c / kchunks;
        const int kc = (c - j * kchunks) * BK;
        const int off = p.tap_base + j * p.tap_step;
        const long a_off = (long)off * p.lda + kc;
        const long b_off = (long)(j * p.Kc + kc) * p.ldw;
        SG_LOAD_A(0, ra0, ok0) SG_LOAD_A(1, ra1, ok1) SG_LOAD_A(2, ra2, ok2) SG_LOAD_A(3, ra3, ok3)
        SG_LOAD_B(0, rb0) SG_LOAD_B(1, rb1) SG_LOAD_B(2, rb2) SG_LOAD_B(3, rb3)
    };
    auto store_chunk = [&](int buf) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 r0 = ok0 ? ra0 : z, r1 = ok1 ? ra1 : z, r2 = ok2 ? ra2 : z, r3 = ok3 ? ra3 : z;
        if (a_on) {
            float* a = As + buf * BK * BM + (a_c4 * 4) * BM + a_r4 * 4;
            // register transpose: r_i = 4 k-values of row i  ->  one float4 of 4 rows per k
            *reinterpret_cast<float4*>(a + 0 * BM) = make_float4(r0.x, r1.x, r2.x, r3.x);
            *reinterpret_cast<float4*>(a + 1 * BM) = make_float4(r0.y, r1.y, r2.y, r3.y);
            *reinterpret_cast<float4*>(a + 2 * BM) = make_float4(r0.z, r1.z, r2.z, r3.z);
            *reinterpret_cast<float4*>(a + 3 * BM) = make_float4(r0.w, r1.w, r2.w, r3.w);
        }
        SG_STORE_B(0, rb0) SG_STORE_B(1, rb1) SG_STORE_B(2, rb2) SG_STORE_B(3, rb3)
    };
#undef SG_LOAD_A
#undef SG_LOAD_B
#undef SG_STORE_B

    // Software pipeline over K-chunks, two LDS stages, one register stage:
    //   chunk c computes from LDS stage c&1; in the MIDDLE of its MFMA stream the registers holding
    //   chunk c+1 (loaded half a chunk earlier) are written to the other LDS stage and the loads of
    //   chunk c+2 are issued.  The staging instructions (address VALU, ds_write, global_load) thus
    //   issue in the shadow of this wave's own MFMAs; only the barrier remains at the chunk boundary.
    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk(0);
        if (c_begin + 1 < c_end) load_chunk(c_begin + 1);
    }
    __syncthreads();

    const int a_rd = wm * (BM / WM) + l31;
    const int b_rd = wn * (BN / WN) + l31;
    constexpr int KS_STAGE = 5;  // k-step after which the next chunk is staged
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        __builtin_amdgcn_sched_barrier(0);
        // k order shared by every kernel in this file (so results do not depend on which one runs):
        // MFMA step ks = 4 kg + s consumes k = 8 kg + 4 lhi + s -- see gemm_segment_q, whose 16-byte
        // operand reads dictate it.  KROW(ks) is the k-row of lane half 0.
#define KROW(ks) (8 * ((ks) >> 2) + ((ks) & 3))
        const float* a_s = As + buf * BK * BM + lhi * 4 * BM + a_rd;
        const float* b_s = Bs + buf * BK * BN + lhi * 4 * BN + b_rd;
        // register double-buffered operand fetch: the ds_reads of k-step ks+1 are issued before the
        // MFMAs of ks, so their LDS latency hides under MI*NI x 64 cycles of matrix pipe
        float av[2][MI], bv[2][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) av[0][mi] = a_s[mi * 32];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bv[0][ni] = b_s[ni * 32];
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            if (ks + 1 < BK / 2) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) av[(ks + 1) & 1][mi] = a_s[KROW(ks + 1) * BM + mi * 32];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bv[(ks + 1) & 1][ni] = b_s[KROW(ks + 1) * BN + ni * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks & 1][mi], bv[ks & 1][ni], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == KS_STAGE) {
                if (c + 1 < c_end && !(p.ablate & 2)) store_chunk(buf ^ 1);
                if (c + 2 < c_end && !(p.ablate & 1)) load_chunk(c + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!(p.ablate & 4)) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// 8-wave variant: 128 x 128 tile, 512 threads as 2 (M) x 4 (N) waves, each wave 64 x 32.
// Staging is role-split: waves 0-3 fetch and transpose the A chunk (4 rows x 4 k per thread),
// waves 4-7 copy the W chunk -- 4 global loads + 4 LDS stores per thread per chunk, half of the
// 4-wave 64 x 128 block, and a third less L2 traffic per MAC.  64 KB of LDS -> two blocks per CU
// = 4 waves per SIMD.  Loads stay branch-free: both roles issue the same four loads through
// role-selected pointers.
__device__ __forceinline__ void gemm_segment8(const ConvGemmArgs& p, float* smem, int m0, int n0, int c_begin,
                                              int c_end, f32x16 (&acc)[2][1]) {
    constexpr int BM = 128, BN = 128;
    float* As = smem;                 // [2][BK][BM]
    float* Bs = smem + 2 * BK * BM;   // [2][BK][BN]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int l31 = lane & 31, lhi = lane >> 5;
    const bool role_a = wid < 4;

    const int a_r4 = (wid & 3) * 8 + (lane & 7);  // A role: row quad 0..31
    const int a_c4 = lane >> 3;                   //         float4 of the 32-float slice
    const float* ptr[4];
    int a_t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (role_a) {
            const int r = m0 + a_r4 * 4 + i;
            if (r < p.M) {
                const int b = r / p.Tc;
                const int t = r - b * p.Tc;
                ptr[i] = p.A + (size_t)(b * p.Ta + t) * p.lda + a_c4 * 4;
                a_t[i] = t;
            } else {
                ptr[i] = p.A;
                a_t[i] = -(1 << 28);
            }
        } else {
            const int f = (tid - 256) + i * 256;  // float4 index in the [32][128] W chunk
            ptr[i] = p.W + (size_t)(f >> 5) * p.ldw + n0 + (f & 31) * 4;
            a_t[i] = 0;
        }
    }
    const int kchunks = p.Kc / BK;
    float4 r0, r1, r2, r3;
    bool ok0, ok1, ok2, ok3;
#define SG_LD(i, r, okv)                                                              \
    {                                                                                 \
        const int tt = a_t[i] + off;                                                  \
        okv = !role_a || (tt >= 0 && tt < p.Ta);                                      \
        r = *reinterpret_cast<const float4*>(okv ? ptr[i] + eoff : p.A);              \
    }
    auto load_chunk = [&](int c) {
        const int j = c / kchunks;
        const int kc = (c - j * kchunks) * BK;
        const int off = role_a ? p.tap_base + j * p.tap_step : 0;
        const long eoff = role_a ? (long)off * p.lda + kc : (long)(j * p.Kc + kc) * p.ldw;
        SG_LD(0, r0, ok0) SG_LD(1, r1, ok1) SG_LD(2, r2, ok2) SG_LD(3, r3, ok3)
    };
#undef SG_LD
    auto store_chunk = [&](int buf) {
        if (role_a) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 q0 = ok0 ? r0 : z, q1 = ok1 ? r1 : z, q2 = ok2 ? r2 : z, q3 = ok3 ? r3 : z;
            float* a = As + buf * BK * BM + (a_c4 * 4) * BM + a_r4 * 4;
            *reinterpret_cast<float4*>(a + 0 * BM) = make_float4(q0.x, q1.x, q2.x, q3.x);
            *reinterpret_cast<float4*>(a + 1 * BM) = make_float4(q0.y, q1.y, q2.y, q3.y);
            *reinterpret_cast<float4*>(a + 2 * BM) = make_float4(q0.z, q1.z, q2.z, q3.z);
            *reinterpret_cast<float4*>(a + 3 * BM) = make_float4(q0.w, q1.w, q2.w, q3.w);
        } else {
            float* b = Bs + buf * BK * BN + (tid - 256) * 4;
            *reinterpret_cast<float4*>(b + 0 * 1024) = r0;
            *reinterpret_cast<float4*>(b + 1 * 1024) = r1;
            *reinterpret_cast<float4*>(b + 2 * 1024) = r2;
            *reinterpret_cast<float4*>(b + 3 * 1024) = r3;
        }
    };
    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk(0);
        if (c_begin + 1 < c_end) load_chunk(c_begin + 1);
    }
    __syncthreads();
    const int a_rd = wm * 64 + l31;
    const int b_rd = wn * 32 + l31;
    constexpr int KS_STAGE = 5;
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        __builtin_amdgcn_sched_barrier(0);
        const float* a_s = As + buf * BK * BM + lhi * 4 * BM + a_rd;
        const float* b_s = Bs + buf * BK * BN + lhi * 4 * BN + b_rd;
        float av[2][2], bv[2];
        av[0][0] = a_s[0];
        av[0][1] = a_s[32];
        bv[0] = b_s[0];
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            if (ks + 1 < BK / 2) {
                av[(ks + 1) & 1][0] = a_s[KROW(ks + 1) * BM];
                av[(ks + 1) & 1][1] = a_s[KROW(ks + 1) * BM + 32];
                bv[(ks + 1) & 1] = b_s[KROW(ks + 1) * BN];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks & 1][0], bv[ks & 1], acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks & 1][1], bv[ks & 1], acc[1][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == KS_STAGE) {
                if (c + 1 < c_end && !(p.ablate & 2)) store_chunk(buf ^ 1);
                if (c + 2 < c_end && !(p.ablate & 1)) load_chunk(c + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!(p.ablate & 4)) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Quad-fed tiles (8 or 16 waves): every MFMA operand fetch is one ds_read_b128 that serves FOUR
// k-steps (12 LDS instructions per 32 MFMAs instead of 32), and staging is a plain 16-byte copy.
//
// Why: on gfx950 the f32 MFMA shares its issue/datapath with the VALU -- tools/native/mfma_mix.hip
// measures ~4.7 matrix-pipe cycles lost per VALU instruction and ~7 per ds_read_b32 issued by ANY
// wave of the SIMD, while a b128-fed loop runs at 154 of 157 TFLOP/s.  The b32-fed loop above plus
// its register transposes / zero-selects tops out near 120.
//
// How: the two k-values an MFMA step consumes may be ANY two of the chunk as long as A and W agree.
// Step s of k-group kg takes k = 8 kg + 4 lhi + s (lhi = lane / 32), so a lane's A operands of steps
// s = 0..3 are 4 consecutive floats of its row (activations stay row-major in LDS, no transpose) and
// its W operands are 4 consecutive k of its column, contiguous because the weights are pre-packed
// k4-major at load time (Wq[k/4][n][k%4]).
// LDS image: A[BM rows][8 slots of 16 B], slot c of row r stored at c ^ ((r >> 1) & 7): conflict-free
// for the 16-lane groups ds_read_b128 is served in and for the 8-lane groups of ds_write_b128;
// W[8 k4-groups][128 n][4].  Out-of-range tap rows are zero-filled by the buffer-load bounds check
// instead of being zeroed in registers.
// WMW = waves along M: 2 -> 8 waves, 128 x 128 tile, 64 KB of LDS (two blocks per CU); 4 -> 16 waves, 256 x 128
// tile, 96 KB (one block owns the CU).  Every wave computes 64 x 32.
template <int WMW>
__device__ __forceinline__ void gemm_segment_q(const ConvGemmArgs& p, float* smem, int m0, int n0, int c_begin,
                                               int c_end, f32x16 (&acc)[2][1], unsigned long long* tr = nullptr) {
    constexpr int BM = 64 * WMW, BN = 128;
    constexpr int A_STAGE = BM * BK, B_STAGE = BK * BN;   // floats per LDS stage
    constexpr int ROLE_T = 128 * WMW;                     // threads per staging role
    constexpr int LW = 8 / WMW;                           // W float4 per W-role thread and chunk (A role: 4)
    constexpr int ST_STRIDE = 512 * WMW;                  // LDS floats between a thread's consecutive float4
    float* As = smem;                   // [2][BM][32]
    float* Bs = smem + 2 * A_STAGE;     // [2][8][BN][4]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int l31 = lane & 31, lhi = lane >> 5;
    const bool role_a = wid < 2 * WMW;
    const int ts = role_a ? tid : tid - ROLE_T;  // index inside the role

    // Staging loads are raw buffer loads: address = descriptor base + per-lane byte offset (VGPR) + a
    // wave-uniform byte offset (SGPR).  The K position inside a tap goes into the SGPR, so a chunk's four
    // loads cost no VALU at all; a lane whose tap row falls outside its utterance (or whose row is past M)
    // carries an out-of-range VGPR offset and the hardware returns zeros.  Per-lane offsets are rebuilt
    // only when the chunk stream moves to the next tap.
    constexpr unsigned kOob = 0x80000000u;
    const int kchunks = p.Kc / BK;
    const __amdgpu_buffer_rsrc_t rsrc =
        role_a ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000)
               : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Wq), 0, p.w_bytes, 0x00020000);
    unsigned row_off[4];  // A role: byte offset of (row, tap 0, k = 4 c4); W role: byte offset in the chunk image
    int a_t[4];
    int st_off;  // LDS float offset of this thread's first staged float4 in stage 0 (the others: + i * ST_STRIDE)
    if (role_a) {
        const int c4 = ts & 7;
        // one integer division per thread and segment; the other three rows are 16 WMW apart
        int b = (m0 + (ts >> 3)) / p.Tc;
        int t = (m0 + (ts >> 3)) - b * p.Tc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = m0 + (ts >> 3) + 16 * WMW * i;
            row_off[i] = ((unsigned)(b * p.Ta + t) * (unsigned)p.lda + (unsigned)(c4 * 4)) * 4u;
            a_t[i] = r < p.M ? t : -(1 << 28);
            t += 16 * WMW;
            while (t >= p.Tc) {
                t -= p.Tc;
                ++b;
            }
        }
        st_off = (ts >> 3) * 32 + ((c4 ^ ((ts >> 4) & 7)) << 2);  // row stride 32 floats; rows + 16 WMW i keep the swizzle
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = ts + i * ROLE_T;  // float4 index in the [8][128] chunk image (i < LW)
            row_off[i] = i < LW ? (unsigned)(((f >> 7) * p.ldw + n0 + (f & 127)) * 16) : kOob;
            a_t[i] = 0;
        }
        st_off = 2 * A_STAGE + ts * 4;
    }
    float* st_ptr0 = smem + st_off;                                    // stage 0
    float* st_ptr1 = st_ptr0 + (role_a ? A_STAGE : B_STAGE);           // stage 1
    // load stream state (chunks are requested in increasing order)
    int ld_j = c_begin / kchunks;
    int ld_kc = (c_begin - ld_j * kchunks) * BK;
    unsigned voff[4];
    auto set_tap = [&](int j) {
        const int off = p.tap_base + j * p.tap_step;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool okv = (unsigned)(a_t[i] + off) < (unsigned)p.Ta;
            voff[i] = !role_a ? row_off[i] : okv ? row_off[i] + (unsigned)(off * p.lda * 4) : kOob;
        }
    };
    set_tap(ld_j);
    i32x4 r0, r1, r2, r3;
    r2 = r3 = i32x4{0, 0, 0, 0};
    auto load_chunk = [&]() {
        // W role: all taps are one contiguous k range -> everything in the SGPR offset
        const int soff = role_a ? ld_kc * 4 : (ld_j * p.Kc + ld_kc) * p.ldw * 4;
        r0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[0], soff, 0);
        r1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[1], soff, 0);
        if (LW == 4 || role_a) {
            r2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[2], soff, 0);
            r3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[3], soff, 0);
        }
        ld_kc += BK;
        if (ld_kc == p.Kc) {
            ld_kc = 0;
            ++ld_j;
            if (role_a) set_tap(ld_j);
        }
    };
    auto store_chunk = [&](float* d) {
        *reinterpret_cast<i32x4*>(d + 0 * ST_STRIDE) = r0;
        *reinterpret_cast<i32x4*>(d + 1 * ST_STRIDE) = r1;
        if (LW == 4 || role_a) {
            *reinterpret_cast<i32x4*>(d + 2 * ST_STRIDE) = r2;
            *reinterpret_cast<i32x4*>(d + 3 * ST_STRIDE) = r3;
        }
    };
    if (c_begin < c_end) {
        load_chunk();
        store_chunk(st_ptr0);
        if (c_begin + 1 < c_end) load_chunk();
    }
    __syncthreads();
    if (tr && threadIdx.x == 0) tr[13] = __builtin_amdgcn_s_memrealtime();       // prologue done
    // Per-lane LDS read pointers are fixed for the whole segment (one per k-group, the XOR swizzle folded in);
    // the stage offset is a compile-time constant of the body below, so it lands in the ds_read / ds_write
    // immediate and a chunk issues no address VALU at all (every VALU instruction costs matrix-pipe cycles).
    const int sw = (l31 >> 1) & 7;
    const float* a_base = As + (wm * 64 + l31) * 32;
    const float* a_kg0 = a_base + (((0 * 2 + lhi) ^ sw) << 2);
    const float* a_kg1 = a_base + (((1 * 2 + lhi) ^ sw) << 2);
    const float* a_kg2 = a_base + (((2 * 2 + lhi) ^ sw) << 2);
    const float* a_kg3 = a_base + (((3 * 2 + lhi) ^ sw) << 2);
    const float* b_base = Bs + (lhi * BN + wn * 32 + l31) * 4;
    auto chunk = [&](int c, auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr int AOFF = BUF * A_STAGE;  // this chunk's stages
        constexpr int BOFF = BUF * B_STAGE;
        __builtin_amdgcn_sched_barrier(0);
        float4 a0 = *reinterpret_cast<const float4*>(a_kg0 + AOFF);
        float4 a1 = *reinterpret_cast<const float4*>(a_kg0 + AOFF + 32 * 32);
        float4 b0 = *reinterpret_cast<const float4*>(b_base + BOFF);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            float4 na0, na1, nb0;
            if (kg < 3) {
                const float* an = kg == 0 ? a_kg1 : kg == 1 ? a_kg2 : a_kg3;
                na0 = *reinterpret_cast<const float4*>(an + AOFF);
                na1 = *reinterpret_cast<const float4*>(an + AOFF + 32 * 32);
                nb0 = *reinterpret_cast<const float4*>(b_base + BOFF + (kg + 1) * 2 * BN * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, acc[1][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, acc[1][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, acc[1][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, acc[1][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kg == 1) {
                if (c + 1 < c_end && !(p.ablate & 2)) store_chunk(BUF ? st_ptr0 : st_ptr1);  // the other stage
                if (c + 2 < c_end && !(p.ablate & 1)) load_chunk();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kg < 3) { a0 = na0; a1 = na1; b0 = nb0; }
        }
        if (!(p.ablate & 4)) __syncthreads();
    };
    int c = c_begin;
    for (; c + 1 < c_end; c += 2) {
        chunk(c, std::integral_constant<int, 0>{});
        chunk(c + 1, std::integral_constant<int, 1>{});
        if (tr && threadIdx.x == 0 && c == c_begin) tr[14] = __builtin_amdgcn_s_memrealtime();  // two chunks done
    }
    if (c < c_end) chunk(c, std::integral_constant<int, 0>{});
}

// 4-wave 64 x 128 quad-fed tile for the one-block-per-tile launches (small batches, tdnn1 forward, the AudioNet
// stack): the same operand scheme and k order as gemm_segment_q, 48 KB of LDS so three blocks share a CU.
// Staging roles: waves 0-1 copy the A chunk (64 rows x 8 float4: 4 per thread), waves 2-3 the W chunk (8 k4-groups
// x 128 columns: 8 per thread).
// MI = row fragments per wave: 2 -> 64 x 128 tile; 1 -> 32 x 128 tile (half the MFMAs per chunk and wave: the k chain of
// a tile, which is what a nearly empty chip waits for at batch <= 8, is half as long).
template <int MI>
__device__ __forceinline__ void gemm_segment_q1(const ConvGemmArgs& p, float* smem, int m0, int n0, int c_begin,
                                                int c_end, f32x16 (&acc)[MI][1]) {
    constexpr int BM = 32 * MI, BN = 128;
    constexpr int LA = 2 * MI;  // A float4 per A-role thread and chunk (rows 16 i + ts / 8)
    constexpr int A_STAGE = BM * BK, B_STAGE = BK * BN;
    constexpr int ST_STRIDE = 512;  // LDS floats between a thread's consecutive float4 (16 A rows / one W k4-group)
    float* As = smem;                   // [2][BM][32]
    float* Bs = smem + 2 * A_STAGE;     // [2][8][BN][4]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid;
    const int l31 = lane & 31, lhi = lane >> 5;
    const bool role_a = wid < 2;
    const int ts = role_a ? tid : tid - 128;
    constexpr unsigned kOob = 0x80000000u;
    const int kchunks = p.Kc / BK;
    const __amdgpu_buffer_rsrc_t rsrc =
        role_a ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000)
               : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Wq), 0, p.w_bytes, 0x00020000);
    unsigned row_off[8];
    int a_t[4];
    int st_off;
    if (role_a) {
        const int c4 = ts & 7;
        int b = (m0 + (ts >> 3)) / p.Tc;
        int t = (m0 + (ts >> 3)) - b * p.Tc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = m0 + (ts >> 3) + 16 * i;
            row_off[i] = i < LA ? ((unsigned)(b * p.Ta + t) * (unsigned)p.lda + (unsigned)(c4 * 4)) * 4u : kOob;
            a_t[i] = (i < LA && r < p.M) ? t : -(1 << 28);
            t += 16;
            while (t >= p.Tc) {
                t -= p.Tc;
                ++b;
            }
        }
#pragma unroll
        for (int i = 4; i < 8; ++i) row_off[i] = kOob;
        st_off = (ts >> 3) * 32 + ((c4 ^ ((ts >> 4) & 7)) << 2);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) row_off[i] = (unsigned)((i * p.ldw + n0 + ts) * 16);  // k4-group i, column ts
#pragma unroll
        for (int i = 0; i < 4; ++i) a_t[i] = 0;
        st_off = 2 * A_STAGE + ts * 4;
    }
    float* st_ptr0 = smem + st_off;
    float* st_ptr1 = st_ptr0 + (role_a ? A_STAGE : B_STAGE);
    int ld_j = c_begin / kchunks;
    int ld_kc = (c_begin - ld_j * kchunks) * BK;
    // both roles run the same eight loads: the A role's slots 4..7 carry an out-of-range offset (the bounds check
    // returns zeros without touching memory) -- role-dependent control flow around the staging registers made
    // hipcc keep them in scratch
    unsigned voff[8];
#pragma unroll
    for (int i = 4; i < 8; ++i) voff[i] = row_off[i];
    auto set_tap = [&](int j) {
        const int off = p.tap_base + j * p.tap_step;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool okv = (unsigned)(a_t[i] + off) < (unsigned)p.Ta;
            voff[i] = !role_a ? row_off[i] : okv ? row_off[i] + (unsigned)(off * p.lda * 4) : kOob;
        }
    };
    set_tap(ld_j);
    i32x4 r0, r1, r2, r3, r4, r5, r6, r7;
    auto load_chunk = [&]() {
        const int soff = role_a ? ld_kc * 4 : (ld_j * p.Kc + ld_kc) * p.ldw * 4;
        r0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[0], soff, 0);
        r1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[1], soff, 0);
        r2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[2], soff, 0);
        r3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[3], soff, 0);
        r4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[4], soff, 0);
        r5 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[5], soff, 0);
        r6 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[6], soff, 0);
        r7 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[7], soff, 0);
        ld_kc += BK;
        if (ld_kc == p.Kc) {
            ld_kc = 0;
            ++ld_j;
            if (role_a) set_tap(ld_j);
        }
    };
    auto store_chunk = [&](float* d) {
        *reinterpret_cast<i32x4*>(d + 0 * ST_STRIDE) = r0;
        *reinterpret_cast<i32x4*>(d + 1 * ST_STRIDE) = r1;
        if (MI == 2 || !role_a) {
            *reinterpret_cast<i32x4*>(d + 2 * ST_STRIDE) = r2;
            *reinterpret_cast<i32x4*>(d + 3 * ST_STRIDE) = r3;
        }
        if (!role_a) {
            *reinterpret_cast<i32x4*>(d + 4 * ST_STRIDE) = r4;
            *reinterpret_cast<i32x4*>(d + 5 * ST_STRIDE) = r5;
            *reinterpret_cast<i32x4*>(d + 6 * ST_STRIDE) = r6;
            *reinterpret_cast<i32x4*>(d + 7 * ST_STRIDE) = r7;
        }
    };
    if (c_begin < c_end) {
        load_chunk();
        store_chunk(st_ptr0);
        if (c_begin + 1 < c_end) load_chunk();
    }
    __syncthreads();
    const int sw = (l31 >> 1) & 7;
    const float* a_base = As + l31 * 32;
    const float* a_kg0 = a_base + (((0 * 2 + lhi) ^ sw) << 2);
    const float* a_kg1 = a_base + (((1 * 2 + lhi) ^ sw) << 2);
    const float* a_kg2 = a_base + (((2 * 2 + lhi) ^ sw) << 2);
    const float* a_kg3 = a_base + (((3 * 2 + lhi) ^ sw) << 2);
    const float* b_base = Bs + (lhi * BN + wn * 32 + l31) * 4;
    auto chunk = [&](int c, auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr int AOFF = BUF * A_STAGE;
        constexpr int BOFF = BUF * B_STAGE;
        __builtin_amdgcn_sched_barrier(0);
        float4 a0 = *reinterpret_cast<const float4*>(a_kg0 + AOFF);
        float4 a1 = a0;
        if (MI == 2) a1 = *reinterpret_cast<const float4*>(a_kg0 + AOFF + 32 * 32);
        float4 b0 = *reinterpret_cast<const float4*>(b_base + BOFF);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            float4 na0, na1, nb0;
            if (kg < 3) {
                const float* an = kg == 0 ? a_kg1 : kg == 1 ? a_kg2 : a_kg3;
                na0 = *reinterpret_cast<const float4*>(an + AOFF);
                na1 = na0;
                if (MI == 2) na1 = *reinterpret_cast<const float4*>(an + AOFF + 32 * 32);
                nb0 = *reinterpret_cast<const float4*>(b_base + BOFF + (kg + 1) * 2 * BN * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[0][0], 0, 0, 0);
            if (MI == 2) acc[MI - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, acc[MI - 1][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[0][0], 0, 0, 0);
            if (MI == 2) acc[MI - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, acc[MI - 1][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[0][0], 0, 0, 0);
            if (MI == 2) acc[MI - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, acc[MI - 1][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[0][0], 0, 0, 0);
            if (MI == 2) acc[MI - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, acc[MI - 1][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kg == 1) {
                if (c + 1 < c_end) store_chunk(BUF ? st_ptr0 : st_ptr1);
                if (c + 2 < c_end) load_chunk();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kg < 3) { a0 = na0; a1 = na1; b0 = nb0; }
        }
        __syncthreads();
    };
    int c = c_begin;
    for (; c + 1 < c_end; c += 2) {
        chunk(c, std::integral_constant<int, 0>{});
        chunk(c + 1, std::integral_constant<int, 1>{});
    }
    if (c < c_end) chunk(c, std::integral_constant<int, 0>{});
}

// ------------------------------------------------------------------------------------------
// Wave-specialised quad-fed segment for the ONE-BLOCK-PER-CU stream-K kinds (8 ... 32 utterances per GPU: the shards of
// a batch of 64 cut over 8 / 4 / 2 GPUs), where a SIMD holds one computing wave (two at 32) and nothing hides that
// wave's own issue gaps.  What round 3 measured on the two-stage kernels above at these sizes (rocprofv3 per-layer times,
// PMC passes, tools/native/mfma_chain_chip.hip, tools/native/mfma_burst.hip):
//   * memory is not the limit: L2 hit rate 88 %, 68 MB of fabric traffic per tdnn3 launch at 8 utterances, and a
//     register ring three chunks deep changed nothing;
//   * a dependent chain of v_mfma_f32_32x32x2_f32 runs at 64-67 cycles per instruction on all 1024 SIMDs at once, one
//     wave each (150 TFLOP/s), and 8 ds_read_b128 + one s_barrier per 16 MFMAs cost < 4 %;
//   * but the STAGING issued by the same wave -- 5 buffer loads, the wait for them, 5 ds_write_b128 per chunk -- stretches
//     a 1024-cycle chunk to 1550 (micro-benchmark) ... 1800 cycles (the kernel): 57 % of peak at 8 utterances, 66 % at 16.
// So the roles are split between waves: the first 4 WM waves of a block (one or two per SIMD) only fetch operands from LDS
// and multiply -- per chunk: one barrier, the 16-byte operand reads of the WHOLE next chunk into a second register
// set, then 16 (MI = 1) or 32 (MI = 2) MFMAs back to back -- and four more waves (one per SIMD) only stage: per chunk
// the LDS stores of chunk c+2 from a two-deep register ring, the buffer loads of chunk c+4, one barrier.  Three LDS
// stages.  The micro-benchmark of this split: 1320 cycles per chunk (77 %) against 1550 (66 %) for one wave doing both.
// Same LDS images, same operand values, same k order as gemm_segment_q / gemm_segment_q1: bit-identical results.
// WM computing waves along M x 4 along N; a computing wave owns (32 MI) x 32 of the (32 MI WM) x 128 tile.
// Loads past the segment's last chunk carry out-of-range offsets (the bounds check returns zeros without touching
// memory), so the staging loop has no conditional loads (a load behind a branch makes hipcc drain vmcnt at the join).
// Every wave of the block executes the same number of barriers: one after the prologue, one per chunk.  The two sides are
// separate functions (ws_stage_segment / ws_compute_segment) and the kernel branches on the role ONCE, at its top: with
// the branch inside one segment function the staging waves kept the (dead) accumulators alive through their loop -- 64
// registers at 64 x 64 wave tiles -- and the 168-register budget of three waves per SIMD spilled.
// The staging waves' side of a segment (threads NC .. NC + 255 of the block).
// BG: the computing waves fetch their W operands from the L2 themselves (ws_compute_segment); only the A tile is staged.
template <int WM, int WNC, int MI, int NI, bool BG>
__device__ __forceinline__ void ws_stage_segment(const ConvGemmArgs& p, float* smem, int m0, int n0, int c_begin, int c_end) {
    static_assert(WNC * NI == 4, "a tile is 128 columns wide");
    constexpr int BM = 32 * MI * WM, BN = 128, NC = 64 * WM * WNC;
    constexpr int A_STAGE = BM * BK, B_STAGE = BG ? 0 : BK * BN, STAGE = A_STAGE + B_STAGE;  // floats; stage s = [A | W] at s * STAGE
    const int n = c_end - c_begin;
    // ============================================================ staging waves
    constexpr int LA = BM * 8 / 256, LW = BG ? 0 : 1024 / 256, LWD = LW > 0 ? LW : 1, APASS = 32;  // float4 per thread and chunk; A rows per pass
    const int tid = (int)threadIdx.x - NC;
    constexpr unsigned kOob = 0x80000000u;
    const int kchunks = p.Kc / BK;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Wq), 0, p.w_bytes, 0x00020000);
    const int c4 = tid & 7, r0 = tid >> 3;
    // per-thread rows m0 + r0 + 32 i: only their per-tap byte offsets are kept (a_voff); (utterance, frame) of the first row
    // are, the others are re-derived when the chunk stream moves to the next tap (every Kc / 32 chunks) -- eight rows per
    // thread at 256-row tiles, and the staging waves share the 168 registers of three waves per SIMD
    unsigned a_voff[LA], w_voff[LWD];
    const int row0 = m0 + r0;
    const int b0 = row0 / p.Tc, t0 = row0 - b0 * p.Tc;
#pragma unroll
    for (int i = 0; i < LW; ++i) {
        const int f = tid + i * 256;  // float4 index in the [8 k4-groups][128 columns] chunk image
        w_voff[i] = (unsigned)(((f >> 7) * p.ldw + n0 + (f & 127)) * 16);
    }
    float* st_a = smem + r0 * 32 + ((c4 ^ ((r0 >> 1) & 7)) << 2);  // + i * APASS * 32: rows 16 apart keep the swizzle
    float* st_w = smem + A_STAGE + tid * 4;                          // + i * 1024
    int ld_j = c_begin / kchunks;
    int ld_kc = (c_begin - ld_j * kchunks) * BK;
    int ld_left = n;
    auto set_tap = [&](int j) __attribute__((always_inline)) {
        const int off = p.tap_base + j * p.tap_step;
        int b = b0, t = t0;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const bool okv = row0 + APASS * i < p.M && (unsigned)(t + off) < (unsigned)p.Ta;
            a_voff[i] = okv ? ((unsigned)(b * p.Ta + t + off) * (unsigned)p.lda + (unsigned)(c4 * 4)) * 4u : kOob;
            t += APASS;
            while (t >= p.Tc) {
                t -= p.Tc;
                ++b;
            }
        }
    };
    set_tap(ld_j);
    // Register ring: slot = chunk % D, D chunks of loads in flight.  With two (round 3, first version) a load had two
    // chunk times to land -- 0.85 us at 32-row tiles while all 256 CUs pull ~7 TB/s out of the L2s: the staging waves were
    // latency-bound again (0.66 us per chunk).  The staging waves hold no accumulators or operands, so the ring is cheap.
    constexpr int D = WM == 1 ? 6 : 2;  // taller tiles: a chunk is >= 1.7 us of compute, and three waves per SIMD leave 168 registers
    i32x4 ra[D][LA], rw[D][LWD];
    auto issue = [&](i32x4 (&a)[LA], i32x4 (&w)[LWD]) __attribute__((always_inline)) {
        const int soff_a = ld_kc * 4, soff_w = (ld_j * p.Kc + ld_kc) * p.ldw * 4;
#pragma unroll
        for (int i = 0; i < LA; ++i) a[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, a_voff[i], soff_a, 0);
#pragma unroll
        for (int i = 0; i < LW; ++i) w[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff[i], soff_w, 0);
        ld_kc += BK;
        --ld_left;
        if (ld_left <= 0) {  // the segment's chunks are all requested: further loads return zeros
#pragma unroll
            for (int i = 0; i < LA; ++i) a_voff[i] = kOob;
#pragma unroll
            for (int i = 0; i < LW; ++i) w_voff[i] = kOob;
            ld_kc = 0;
        } else if (ld_kc == p.Kc) {
            ld_kc = 0;
            ++ld_j;
            set_tap(ld_j);
        }
    };
    auto store = [&](const i32x4 (&a)[LA], const i32x4 (&w)[LWD], int stage_off) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LA; ++i) *reinterpret_cast<i32x4*>(st_a + stage_off + i * APASS * 32) = a[i];
#pragma unroll
        for (int i = 0; i < LW; ++i) *reinterpret_cast<i32x4*>(st_w + stage_off + i * 1024) = w[i];
    };
    // prologue: D + 2 chunks requested, chunks 0 and 1 staged
#pragma unroll
    for (int d = 0; d < D; ++d) issue(ra[d], rw[d]);
    store(ra[0], rw[0], 0);
    issue(ra[0], rw[0]);
    store(ra[1], rw[1], STAGE);
    issue(ra[1], rw[1]);
    __syncthreads();  // barrier 0
    // iteration i: chunk i + 2 goes from ring slot (i + 2) % D to stage (i + 2) % 3 (last read two barriers ago), the slot
    // is refilled with chunk i + 2 + D.  Six iterations per trip: slot and stage are compile-time; one back edge, no exits
    // inside (never-taken structurizer edges would make the waitcnt insertion wait for the loads issued one iteration ago).
    auto step = [&](auto jtag) __attribute__((always_inline)) {
        constexpr int J = decltype(jtag)::value;
        store(ra[(J + 2) % D], rw[(J + 2) % D], ((J + 2) % 3) * STAGE);
        issue(ra[(J + 2) % D], rw[(J + 2) % D]);
        __syncthreads();
    };
    const int trips = n / 6;
    for (int q = 0; q < trips; ++q) {
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
    }
    const int rest = n - 6 * trips;
    if (rest >= 1) step(std::integral_constant<int, 0>{});
    if (rest >= 2) step(std::integral_constant<int, 1>{});
    if (rest >= 3) step(std::integral_constant<int, 2>{});
    if (rest >= 4) step(std::integral_constant<int, 3>{});
    if (rest >= 5) step(std::integral_constant<int, 4>{});
}

// The computing waves' side of a segment (threads 0 .. NC - 1): acc += the segment's chunks.
// BG (the 64-row kind, sk_bg): the W operands do not pass through LDS.  profiles/r03_staging_cost.txt: at small tiles the
// staging costs the computing waves up to 16 points of the MFMA peak, and 4-6 of them are the LDS WRITES of the W tile
// alone (16 KB per chunk whatever the tile height; the reads of the same bytes, the staging instruction count -- LDS-DMA
// was tried -- and the barrier cost nothing).  A computing wave's W operand of a k-group IS a coalesced 1 KB piece of the
// k4-packed weights (lane (column, k half) holds W[2 kg + half][column] as one float4), so each wave requests its own
// 32-column slice straight from the L2: 4 x 16-byte buffer loads per chunk where it had 4 x ds_read_b128, into a register
// ring DB chunks deep, refilled in the middle of the MFMA run like the A reads.  Same values in the same k order:
// bit-identical.  Pays at 64-row tiles (32 MFMAs per 4 loads: -1.7 % over the eight layers at 16 utterances); at 32-row
// tiles (16 MFMAs per 4 loads) it is 8 % slower than staging W through LDS, so kind 7 keeps the LDS path.
template <int WM, int WNC, int MI, int NI, bool BG>
__device__ __forceinline__ void ws_compute_segment(const ConvGemmArgs& p, float* smem, int n0, int c_begin, int c_end,
                                                   f32x16 (&acc)[MI][NI]) {
    static_assert(WNC * NI == 4, "a tile is 128 columns wide");
    constexpr int BM = 32 * MI * WM, BN = 128;
    constexpr int A_STAGE = BM * BK, B_STAGE = BG ? 0 : BK * BN, STAGE = A_STAGE + B_STAGE;
    const int n = c_end - c_begin;
    // ================================================================ computing waves
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WNC, wn = wid % WNC;
    const int l31 = lane & 31, lhi = lane >> 5;
    // per-lane operand addresses: A row, slot of k-group kg XOR-swizzled; W column
    const int sw = (l31 >> 1) & 7;
    const float* a_base = smem + (wm * 32 * MI + l31) * 32;
    const float* a_kg[4] = {a_base + (((0 * 2 + lhi) ^ sw) << 2), a_base + (((1 * 2 + lhi) ^ sw) << 2),
                            a_base + (((2 * 2 + lhi) ^ sw) << 2), a_base + (((3 * 2 + lhi) ^ sw) << 2)};
    const float* b_base = smem + A_STAGE + (lhi * BN + wn * 32 * NI + l31) * 4;
    // Operand sets: the 16-byte pieces of KGS k-groups -- a whole chunk for a 32 x 32 wave tile, half a chunk for the larger
    // ones (a whole chunk of 64 x 32 operands in two sets is 96 registers; three waves per SIMD leave 168 in all).
    constexpr int PARTS = MI * NI >= 4 ? 4 : MI * NI >= 2 ? 2 : 1, KGS = 4 / PARTS;  // 64 x 64 wave tiles: one k-group (16 MFMAs) per part
    float4 oa[2][MI][KGS], ob[2][NI][KGS];  // [set][fragment][k-group of the part]
    auto read_set = [&](float4 (&a)[MI][KGS], float4 (&b)[NI][KGS], int stage_off, int kg0) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < KGS; ++g) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi][g] = *reinterpret_cast<const float4*>(a_kg[kg0 + g] + stage_off + mi * 32 * 32);
            if constexpr (!BG) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) b[ni][g] = *reinterpret_cast<const float4*>(b_base + stage_off + (kg0 + g) * 2 * BN * 4 + ni * 32 * 4);
            }
        }
    };
    // BG: the W ring.  Slot c % DB holds chunk c's four k-groups; chunk c + DB - 1 is requested in the middle of chunk c
    // (into the slot chunk c - 1 has just left).  The cursor is ONE scalar: chunk c of a tile is rows 32 c .. 32 c + 31 of
    // the k4-packed weights whatever the tap, i.e. a byte offset that grows by 32 ldw 4 per chunk, and requests past the
    // segment's end repeat its last chunk (valid memory, never multiplied) -- no vector instruction besides the loads
    // themselves: in a wave that shares its SIMD's issue with MFMAs every VALU instruction costs ~20 MFMA cycles
    // (profiles/r03_staging_cost.txt), the first version of this (tap / k cursor in VGPRs, 7 VALU per chunk) was 9 % slower.
    constexpr int DB = !BG ? 1 : MI == 1 ? 6 : 3;
    float4 obr[DB][NI][4];
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Wq), 0, p.w_bytes, 0x00020000);
    unsigned w_voff[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) w_voff[ni] = (unsigned)((lhi * p.ldw + n0 + (wn * NI + ni) * 32 + l31) * 16);
    const int w_step = __builtin_amdgcn_readfirstlane(BK * p.ldw * 4), kg_step = __builtin_amdgcn_readfirstlane(2 * p.ldw * 16);
    int w_soff = __builtin_amdgcn_readfirstlane(c_begin * BK * p.ldw * 4);
    const int w_last = __builtin_amdgcn_readfirstlane((c_end - 1) * BK * p.ldw * 4);
    auto issue_w = [&](float4 (&w)[NI][4]) __attribute__((always_inline)) {
        const int soff = min(w_soff, w_last);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                w[ni][kg] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff[ni], soff + kg * kg_step, 0));
        w_soff += w_step;
    };
    if constexpr (BG) {
#pragma unroll
        for (int d = 0; d < DB - 1; ++d) issue_w(obr[d]);
    }
    __syncthreads();  // barrier 0: chunks 0 and 1 are staged
    read_set(oa[0], ob[0], 0, 0);
    // iteration i, part h: the MFMAs of the part, and half way through them the operand reads of the NEXT part (of this
    // chunk, or the first of chunk i + 1 in stage (i + 1) % 3, complete since the previous barrier) into the other register
    // set; after the last part the barrier.  k order: step s of k-group kg takes k = 8 kg + 4 (lane / 32) + s.
    auto chunk = [&](auto jtag) __attribute__((always_inline)) {
        constexpr int J = decltype(jtag)::value;
#pragma unroll
        for (int h = 0; h < PARTS; ++h) {
            const int cur = PARTS == 1 ? J % 2 : h % 2;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int idx = 0; idx < 4 * KGS; ++idx) {  // the part's (k-group, step) pairs in k order
                const int g = idx >> 2, st = idx & 3;
                if (idx == 2 * KGS) {
                    // the reads sit in the MIDDLE of the part's MFMA chain: straight after a barrier release every
                    // non-MFMA instruction costs ~22 cycles (MI355X_MICROARCH.md) and the staging waves issue their own
                    // burst there; tools/native/mfma_burst.hip: 1170 cycles per chunk against 1330 with the reads first
                    __builtin_amdgcn_sched_barrier(0);
                    if (h + 1 < PARTS) read_set(oa[1 - cur], ob[1 - cur], (J % 3) * STAGE, KGS * (h + 1));
                    else read_set(oa[1 - cur], ob[1 - cur], ((J + 1) % 3) * STAGE, 0);
                    if constexpr (BG) {
                        if (h == PARTS - 1) issue_w(obr[(J + DB - 1) % DB]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const float4 x = oa[cur][mi][g];
                    const float xa = st == 0 ? x.x : st == 1 ? x.y : st == 2 ? x.z : x.w;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const float4 y = BG ? obr[J % DB][ni][h * KGS + g] : ob[cur][ni][g];
                        const float yb = st == 0 ? y.x : st == 1 ? y.y : st == 2 ? y.z : y.w;
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, yb, acc[mi][ni], 0, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    const int trips = n / 6;
    for (int q = 0; q < trips; ++q) {
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        chunk(std::integral_constant<int, 2>{});
        chunk(std::integral_constant<int, 3>{});
        chunk(std::integral_constant<int, 4>{});
        chunk(std::integral_constant<int, 5>{});
    }
    const int rest = n - 6 * trips;
    if (rest >= 1) chunk(std::integral_constant<int, 0>{});
    if (rest >= 2) chunk(std::integral_constant<int, 1>{});
    if (rest >= 3) chunk(std::integral_constant<int, 2>{});
    if (rest >= 4) chunk(std::integral_constant<int, 3>{});
    if (rest >= 5) chunk(std::integral_constant<int, 4>{});
}

// Walk the accumulator fragments of a tile.  C/D layout of the 32x32 MFMA: col = lane & 31,
// row = (e&3) + 8*(e>>2) + 4*(lane>>5).  fn(mi, ni, row0, col) handles one 16-value fragment.
template <int BM, int BN, int WM, int WN, int EPI>
__device__ __forceinline__ void tile_store(const ConvGemmArgs& p, float* C, int m0, int n0,
                                           f32x16 (&acc)[BM / WM / 32][BN / WN / 32]) {
    constexpr int MI = BM / WM / 32;
    constexpr int NI = BN / WN / 32;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    // Per 32-row fragment (wave-uniform decision): rows all inside M -- every fragment of an interior tile, and of the last
    // row of tiles all those above M (M is a multiple of 32 for most shapes of the model, so the slow form rarely runs; it
    // used to serve every fragment of a partial tile and made the workers that own the last tiles finish ~10 us late) --
    // take the fast form: no row checks, and every address is a wave-uniform 64-bit base (one per fragment row, SALU) plus
    // ONE per-lane 32-bit offset, which the compiler folds into the saddr form of global_load/store -- no per-element VALU
    // address arithmetic.  Rows all outside M: nothing to store.  Otherwise the general form (~10 instructions per element).
    const unsigned lane_off = (unsigned)(4 * lhi * p.ldc + l31) * 4u;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int rowu = m0 + wm * (BM / WM) + mi * 32;  // uniform
        if (rowu + 32 <= p.M) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int colu = n0 + wn * (BN / WN) + ni * 32;  // uniform
                const size_t base = (size_t)rowu * p.ldc + colu;
                float bias = 0.f;
                if (EPI == EPI_BIAS_RELU) bias = p.bias[colu + l31];
                f32x16 mk;
                if (EPI == EPI_RELU_MASK) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const char* mb = reinterpret_cast<const char*>(p.mask + base + (size_t)((e & 3) + 8 * (e >> 2)) * p.ldc);
                        mk[e] = *reinterpret_cast<const float*>(mb + lane_off);
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[mi][ni][e];
                    if (EPI == EPI_BIAS_RELU) v = fmaxf(v + bias, 0.f);
                    if (EPI == EPI_RELU_MASK) v = mk[e] > 0.f ? v : 0.f;
                    char* cb = reinterpret_cast<char*>(C + base + (size_t)((e & 3) + 8 * (e >> 2)) * p.ldc);
                    *reinterpret_cast<float*>(cb + lane_off) = v;
                }
            }
        } else if (rowu < p.M) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int col = n0 + wn * (BN / WN) + ni * 32 + l31;
                const int row0 = rowu + 4 * lhi;
                float bias = 0.f;
                if (EPI == EPI_BIAS_RELU) bias = p.bias[col];
                f32x16 mk;
                if (EPI == EPI_RELU_MASK) {
                    // all 16 mask loads in flight at once, rows clamped instead of branched around (a
                    // guarded load makes hipcc wait vmcnt(0) per element: 64 serial L2 round trips)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = min(row0 + (e & 3) + 8 * (e >> 2), p.M - 1);
                        mk[e] = p.mask[(size_t)row * p.ldc + col];
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + (e & 3) + 8 * (e >> 2);
                    float v = acc[mi][ni][e];
                    if (EPI == EPI_BIAS_RELU) v = fmaxf(v + bias, 0.f);
                    if (EPI == EPI_RELU_MASK) v = mk[e] > 0.f ? v : 0.f;
                    if (row < p.M) C[(size_t)row * p.ldc + col] = v;
                }
            }
        }
    }
}

template <int MI, int NI>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[MI][NI]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
}

// ------------------------------------------------------------------------------------------
// Data-parallel launch: one block per output tile (x optional split-K slabs along blockIdx.z).
template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvGemmArgs p, int mtiles, int ntiles) {
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
    // XCD-aware tile order: the dispatcher places block b on XCD b % 8; give each XCD a contiguous
    // run of tiles so the N-tiles that share an A row-panel hit the same L2.
    const int nblk = mtiles * ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8;
        const int xcd = bid % 8, loc = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / ntiles, nt = bid % ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int c_begin = blockIdx.z * p.chunks_per_split;
    const int c_end = min(p.total_chunks, c_begin + p.chunks_per_split);
    f32x16 acc[BM / WM / 32][BN / WN / 32];
    acc_zero(acc);
    gemm_segment<BM, BN, WM, WN>(p, smem, m0, n0, c_begin, c_end, acc);
    tile_store<BM, BN, WM, WN, EPI>(p, p.C + (size_t)blockIdx.z * p.split_stride, m0, n0, acc);
}

// One block per tile, quad-fed 64 x 128 (MI = 2) or 32 x 128 (MI = 1) (needs the k4-packed weights; no split-K).
template <int EPI, int MI>
__global__ __launch_bounds__(256, 3) void conv_gemm_q_kernel(ConvGemmArgs p, int mtiles, int ntiles) {
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (32 * MI + 128)];
    const int nblk = mtiles * ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8;
        const int xcd = bid % 8, loc = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / ntiles, nt = bid % ntiles;
    const int m0 = mt * 32 * MI, n0 = nt * 128;
    f32x16 acc[MI][1];
    acc_zero(acc);
    gemm_segment_q1<MI>(p, smem, m0, n0, 0, p.total_chunks, acc);
    tile_store<32 * MI, 128, 1, 4, EPI>(p, p.C, m0, n0, acc);
}

// ------------------------------------------------------------------------------------------
// Stream-K launch (128 x 128 tiles, 8 waves): `workers` persistent blocks, all co-resident (2 per CU), each
// owning an equal contiguous range of (tile, K-chunk) iterations, so every block does the same
// number of MFMAs and the launch has no tail (the data-parallel launch of 1080 tiles on 768
// slots idles ~16 % of the matrix pipes in its last round).  A tile whose chunks straddle two
// workers is combined deterministically: the worker that owns the tile's FIRST chunks computes
// them first thing and parks the accumulators in its slab; the worker that owns the LAST chunks
// handles them last, starting from the parked accumulators (same fmaf chain as an unsplit tile,
// so the result is bit-identical to the data-parallel launch) and runs the epilogue.
// All workers are co-resident (grid = resident slots), and a worker parks its head piece before anything
// else, so a waiter never waits on work that depends on it (spins are bounded regardless).  Hand-off = agent-scope release/acquire on one flag
// per worker (cdna guide G16).  A flag holds the EPOCH of the launch that parked the slab (a process-wide
// launch counter passed as a kernel argument), so nothing has to be cleared between launches.
// KIND 0: 8 waves, b32-fed 128x128 (no packed weights); 1: 8 waves, quad-fed 128x128; 2: 16 waves, quad-fed 256x128;
// 3: 4 waves, quad-fed 64x128 (small batches: one block per CU); 4: 4 waves, quad-fed 32x128 (batch 8).
// 5 / 6 / 7: the shapes of 1 / 3 / 4 with the roles split between waves (gemm_segment_ws: the tile's waves only multiply,
// four more waves only stage; three LDS stages) -- what the one-block-per-CU launches of 8 ... 32 utterances per GPU use.
// 8: 256x128 with the roles split: eight computing waves of 64 x 64 (two per SIMD) + four staging waves -- the long-K layers of
// the full batch (the all-in-one 16-wave kind 2 keeps the K = 512 layers, whose 16-chunk tiles feel the split's longer prologue).
// 9: 128x128 as four computing waves of 64 x 64 + four staging waves (two waves per SIMD).
constexpr bool sk_deep(int kind) { return kind >= 5; }
constexpr int sk_wm(int kind) { return (kind == 2 || kind == 8) ? 4 : (kind == 3 || kind == 4 || kind == 6 || kind == 7) ? 1 : 2; }  // waves along M (9: 2)
constexpr int sk_wn(int kind) { return (kind == 8 || kind == 9) ? 2 : 4; }                                                                   // waves along N
constexpr int sk_bm(int kind) { return (kind == 4 || kind == 7) ? 32 : 64 * sk_wm(kind); }
constexpr int sk_threads(int kind) { return 64 * sk_wm(kind) * sk_wn(kind) + (sk_deep(kind) ? 256 : 0); }  // deep: + 4 staging waves
constexpr int sk_min_waves(int kind) { return (kind == 5 || kind == 8) ? 3 : sk_deep(kind) ? 2 : sk_wm(kind) == 1 ? 1 : 4; }  // per SIMD: sets the VGPR budget
// kinds whose computing waves take their W operands straight from the L2 (ws_compute_segment, BG): the 32- and 64-row tiles
#ifndef SG_WS_BG
#define SG_WS_BG 0x40  // bit k: kind k (64-row tiles; at 32-row tiles the four loads per 16 MFMAs cost more than the LDS writes saved: -3 %)
#endif
constexpr bool sk_bg(int kind) { return sk_deep(kind) && ((SG_WS_BG >> kind) & 1); }
constexpr size_t sk_lds_bytes(int kind) {
    return (size_t)(sk_deep(kind) ? 3 : 2) * BK * (sk_bm(kind) + (sk_bg(kind) ? 0 : 128)) * sizeof(float);
}
template <int EPI, int KIND>
__global__ __launch_bounds__(sk_threads(KIND), sk_min_waves(KIND)) void conv_gemm_streamk_kernel(ConvGemmArgs p, int ntiles, int tiles,
                                                                                      int iters_per_worker, float* slabs,
                                                                                      unsigned* flags, unsigned epoch) {
    constexpr int WM = sk_wm(KIND), WN = sk_wn(KIND);
    constexpr int BM = sk_bm(KIND), BN = 128;
    constexpr int MI = BM / WM / 32, NI = BN / WN / 32;  // 2x1 (or 1x1) fragments per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 * BK * (BM + BN) floats
    const int C = p.total_chunks;
    const long total = (long)tiles * C;
    // Worker id: block b runs on XCD b % 8; XCD x takes the CONTIGUOUS workers [x*G, (x+1)*G), so the tiles an
    // XCD's L2 (4 MB) sees at any time are neighbours (shared A row panels with m-tile-major numbering, shared W
    // panels with n-tile-major numbering) and a hand-off partner is in the same XCD.  With w = blockIdx.x every
    // XCD touched every panel at unrelated k positions (stream-K ranges drift by ipw - C chunks per worker).
    // Measured (sum over six big layers): w = blockIdx.x 124.0, contiguous + n-major 134.2, contiguous + m-major
    // 135.4 TFLOP/s (tdnn4 101 -> 119); the two tile orders differ by < 1 %: the kernel is MFMA-bound, the PMC
    // passes in profiles/r01_pmc_tdnn3.json show the fabric traffic either way is absorbed by the Infinity Cache.
    // sk_xcd: 0 = blockIdx order, 1 = contiguous + n-tile-major, 2 = contiguous + m-tile-major (default).
    const int per_xcd = gridDim.x >> 3;
    const int w = (p.sk_xcd && (gridDim.x & 7) == 0) ? (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    const int mtiles = tiles / ntiles;
    const long it_begin = (long)w * iters_per_worker;
    const long it_end = min(total, it_begin + iters_per_worker);
    if (it_begin >= it_end) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // Wave-specialised kinds: the waves past the tile's WM x WN only stage operands.  They run the worker's schedule of
    // segments on their own -- same segments in the same order, and the same barriers as the computing waves execute around
    // the hand-off -- and hold no accumulators.
    if constexpr (sk_deep(KIND)) {
        if (__builtin_amdgcn_readfirstlane(wid) >= WM * WN) {
            const int first_tile = (int)(it_begin / C), first_c0 = (int)(it_begin % C);
            const int last_tile = (int)((it_end - 1) / C), last_c1 = (int)((it_end - 1) % C) + 1;
            const bool head_piece = last_c1 < C, tail_piece = first_c0 > 0;
            auto origin = [&](int tile, int& m0, int& n0) {
                if (p.sk_xcd == 1) {
                    m0 = (tile % mtiles) * BM;
                    n0 = (tile / mtiles) * BN;
                } else {
                    m0 = (tile / ntiles) * BM;
                    n0 = (tile % ntiles) * BN;
                }
            };
            int m0, n0;
            if (head_piece && !(last_tile == first_tile && tail_piece)) {
                origin(last_tile, m0, n0);
                ws_stage_segment<WM, WN, MI, NI, sk_bg(KIND)>(p, smem, m0, n0, 0, last_c1);
                __syncthreads();  // the computing waves' barrier between parking the slab and publishing the flag
            }
            for (int tile = tail_piece ? first_tile + 1 : first_tile; tile < (head_piece ? last_tile : last_tile + 1); ++tile) {
                origin(tile, m0, n0);
                ws_stage_segment<WM, WN, MI, NI, sk_bg(KIND)>(p, smem, m0, n0, 0, C);
            }
            if (tail_piece) {
                origin(first_tile, m0, n0);
                __syncthreads();  // the computing waves' barrier after the flag wait
                ws_stage_segment<WM, WN, MI, NI, sk_bg(KIND)>(p, smem, m0, n0, first_c0, first_tile == last_tile ? last_c1 : C);
            }
            return;
        }
    }
    constexpr bool computing = true;
    f32x16 acc[MI][NI];
    auto segment = [&](int m0, int n0, int c0, int c1, unsigned long long* tr = nullptr) __attribute__((always_inline)) {
        if constexpr (KIND == 0) gemm_segment8(p, smem, m0, n0, c0, c1, acc);
        else if constexpr (sk_deep(KIND)) ws_compute_segment<WM, WN, MI, NI, sk_bg(KIND)>(p, smem, n0, c0, c1, acc);
        else if constexpr (KIND == 3) gemm_segment_q1<2>(p, smem, m0, n0, c0, c1, acc);
        else if constexpr (KIND == 4) gemm_segment_q1<1>(p, smem, m0, n0, c0, c1, acc);
        else gemm_segment_q<WM>(p, smem, m0, n0, c0, c1, acc, tr);
    };
    // Parked accumulators travel as 16-byte pieces: piece q of fragment (mi, ni) of wave wid, lane-interleaved, so
    // a wave's store/load instruction covers 1 KB contiguous.  Stores and loads are write-through / L1-bypassing
    // (sc1) raw buffer accesses: no release fence (which would write back every dirty line of the XCD's L2, i.e.
    // the previous layer's output) and no acquire fence are needed around them -- cdna guide G16 R1, price list
    // "publish-large": 3.0 vs 8.2 us per 64 KB published.
    const __amdgpu_buffer_rsrc_t slab_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(slabs, 0, (int)min((size_t)gridDim.x * BM * BN * sizeof(float), (size_t)0x7FFFFFFF), 0x00020000);
    auto piece_off = [&](int wk, int mi, int ni, int q) {
        return (unsigned)((((size_t)wk * (BM * BN / 4)) + ((((wid * MI + mi) * NI + ni) * 4 + q) * 64 + lane)) * 16);
    };

    const int first_tile = (int)(it_begin / C), first_c0 = (int)(it_begin % C);
    const int last_tile = (int)((it_end - 1) / C), last_c1 = (int)((it_end - 1) % C) + 1;
    const bool head_piece = last_c1 < C;                   // my range stops inside last_tile
    const bool tail_piece = first_c0 > 0;                  // my range starts inside first_tile
    auto tile_origin = [&](int tile, int& m0, int& n0) {
        if (p.sk_xcd == 1) {
            m0 = (tile % mtiles) * BM;
            n0 = (tile / mtiles) * BN;
        } else {
            m0 = (tile / ntiles) * BM;
            n0 = (tile % ntiles) * BN;
        }
    };

    // phase timestamps for tools/sk_trace.py (100 MHz wall clock; slot 15 = hardware id of the block's first wave)
#define SG_STAMP(i) \
    if (p.trace && threadIdx.x == 0) p.trace[(size_t)w * 16 + (i)] = __builtin_amdgcn_s_memrealtime();
    if (p.trace && threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p.trace[(size_t)w * 16 + 15] = ((unsigned long long)xcc << 32) | hw;
        for (int i = 1; i < 15; ++i) p.trace[(size_t)w * 16 + i] = 0;
    }
    SG_STAMP(0)
    // (wave-specialised kinds: slots 13 / 14 = the shader clock counter at the worker's start and end -- tools/sk_trace.py
    // prints the clock the CU actually ran at)
    if (sk_deep(KIND) && p.trace && threadIdx.x == 0) p.trace[(size_t)w * 16 + 13] = __builtin_readcyclecounter();
    // 1. the head piece of my last tile (chunks [0, last_c1)): park it for worker w+1
    if (head_piece && !(last_tile == first_tile && tail_piece)) {
        int m0, n0;
        tile_origin(last_tile, m0, n0);
        acc_zero(acc);
        segment(m0, n0, 0, last_c1);
        SG_STAMP(1)
        if (computing) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), slab_rsrc, piece_off(w, mi, ni, q), 0, 16);
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains its write-through stores ...
        // A/B knob (SG_ABLATE bit 16): additionally the classic agent-scope release fence (writes back the XCD's dirty L2
        // lines); the default relies on the sc1 write-through semantics alone, see DESIGN.md section 4
        if (p.ablate & 16) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();                                   // ... before ONE lane publishes the flag
        if (threadIdx.x == 0) {
            // (ablate bit 8 = fault injection for tests/test_gpu_conv.py: the flag is never published)
            if (!(p.ablate & 8)) __hip_atomic_store(flags + w, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        SG_STAMP(2)
    }
    // 2. whole tiles
    const int whole_begin = tail_piece ? first_tile + 1 : first_tile;
    const int whole_end = head_piece ? last_tile : last_tile + 1;
    for (int tile = whole_begin; tile < whole_end; ++tile) {
        int m0, n0;
        tile_origin(tile, m0, n0);
        acc_zero(acc);
        segment(m0, n0, 0, C);
        SG_STAMP(3 + 2 * min(tile - whole_begin, 2))
        if (computing) tile_store<BM, BN, WM, WN, EPI>(p, p.C, m0, n0, acc);
        SG_STAMP(4 + 2 * min(tile - whole_begin, 2))
    }
    // 3. the tail piece of my first tile (chunks [first_c0, C)): RESUME from the accumulators worker
    //    w-1 parked, so every output element sees exactly the fmaf chain of an unsplit tile -- the
    //    result is bit-identical to the one-block-per-tile launch and does not depend on the batch.
    if (tail_piece) {
        int m0, n0;
        tile_origin(first_tile, m0, n0);
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(flags + w - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > ((p.ablate & 8) ? (1u << 12) : (1u << 26))) {
                    // bounded (a partner that never became resident must not hang the GPU); the tile is then wrong,
                    // so raise the context's health word: the next API call / sg_sync fails with SG_ERR_HIP
                    if (p.err_word) __hip_atomic_fetch_or(p.err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
        if (p.ablate & 16) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        SG_STAMP(9)
        // the slab was stored write-through (sc1) before the flag; sc1 loads bypass this CU's L1, so no acquire fence
        if (computing) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc, piece_off(w - 1, mi, ni, q), 0, 16));
                        acc[mi][ni][4 * q] = v.x; acc[mi][ni][4 * q + 1] = v.y; acc[mi][ni][4 * q + 2] = v.z; acc[mi][ni][4 * q + 3] = v.w;
                    }
        }
        const int c1 = first_tile == last_tile ? last_c1 : C;  // (host guarantees == C, see launcher)
        SG_STAMP(10)
        segment(m0, n0, first_c0, c1, p.trace ? p.trace + (size_t)w * 16 : nullptr);
        SG_STAMP(11)
        if (computing) tile_store<BM, BN, WM, WN, EPI>(p, p.C, m0, n0, acc);
        SG_STAMP(12)
    }
    if (sk_deep(KIND) && p.trace && threadIdx.x == 0) p.trace[(size_t)w * 16 + 14] = __builtin_readcyclecounter();
#undef SG_STAMP
}

template <int BM, int BN, int WM, int WN>
static hipError_t launch_tile(const ConvGemmArgs& a, int epi, int splits, hipStream_t s) {
    const int mtiles = (a.M + BM - 1) / BM;
    const int ntiles = a.N / BN;
    dim3 grid(mtiles * ntiles, 1, splits);
    switch (epi) {
        case EPI_NONE:
            hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, EPI_NONE>), grid, dim3(256), 0, s, a, mtiles, ntiles);
            break;
        case EPI_BIAS_RELU:
            hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, EPI_BIAS_RELU>), grid, dim3(256), 0, s, a, mtiles, ntiles);
            break;
        case EPI_RELU_MASK:
            hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, EPI_RELU_MASK>), grid, dim3(256), 0, s, a, mtiles, ntiles);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Nearly empty chip (batch 1-4: the reference's default batch_size is 1 and real utterances differ in length, so one
// utterance per call is the common drop-in case): a 32 x 128 tile keeps ONE CU busy for the whole k range -- 114 k
// matrix-pipe cycles for tdnn3, 48 us, while 36 tiles leave 220 CUs idle.  Smaller tiles need a smaller MFMA:
// v_mfma_f32_16x16x4_f32 -- and tools/native/mfma_order.hip shows that both f32 MFMA shapes ARE a sequential fmaf chain
// over k (bit for bit, 20480 outputs, K up to 3584), so a 16 x 16 block fed the k values in the order the 32x32x2 kernels
// consume them -- per k-group of 8: (0, 4, 1, 5) then (2, 6, 3, 7) -- gives the same bits with a quarter of the
// per-tile work: 544 independent waves at batch 1 instead of 144.
// A block is four waves = 32 x 32 outputs.  Per chunk of 32 k the block stages 32 rows x 32 k of A and 32 k x 32 columns
// of the k4-packed W in LDS (one 16-byte load of each per thread, coalesced; the same images and the same XOR swizzle
// as the quad-fed kernels), through a register ring D chunks deep -- a chunk is only ~300 cycles of MFMA chain, so the
// loads run ~2000 cycles ahead -- and three LDS stages, so that a wave fetches the NEXT chunk's operands from
// LDS while the MFMA chain of the current one runs.  Lane (row l % 16, k-half h = (l / 16) & 1, element pair
// e = l / 32) reads the 16 bytes A[row][8 kg + 4 h ..] and W[2 kg + h][col l % 16][..] of a k-group and feeds elements
// e and 2 + e of each to the k-group's two MFMAs.  (A first version without LDS, every lane loading its operands
// straight from L2, was slower than the 32-row tiles: 16 rows x 2 halves = 32 cache lines per load instruction keep
// the CU's texture path busier than the MFMA chain.)  Out-of-range tap rows come back as zeros from the buffer bounds
// check, as everywhere.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int EPI, int D, bool TRACE>
__global__ __launch_bounds__(256) void conv_gemm_s16_kernel(ConvGemmArgs p, int ntile32) {
    static_assert(D % 3 == 0, "the LDS stage of a chunk is taken from its ring slot");
    constexpr int A_STAGE = 32 * 32, STAGE = A_STAGE + 8 * 32 * 4;  // floats: A 4 KB + W 4 KB
    __shared__ __attribute__((aligned(16))) float smem[3 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int bm0 = ((int)blockIdx.x / ntile32) * 32, bn0 = ((int)blockIdx.x % ntile32) * 32;
    constexpr unsigned kOob = 0x80000000u;
    // ---- staging role of this thread: A float4 (row tid / 8, slot tid % 8), W float4 (k4-group tid / 32, column tid % 32)
    const int srow = bm0 + (tid >> 3), c4 = tid & 7;
    const bool srow_ok = srow < p.M;
    const int sb = srow_ok ? srow / p.Tc : 0;
    const int st = srow_ok ? srow - sb * p.Tc : 0;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Wq), 0, p.w_bytes, 0x00020000);
    const unsigned a_base = ((unsigned)(sb * p.Ta + st) * (unsigned)p.lda + (unsigned)(4 * c4)) * 4u;
    const unsigned w_voff = (unsigned)(((tid >> 5) * p.ldw + bn0 + (tid & 31)) * 16);
    auto tap_voff = [&](int j) {
        const int off = p.tap_base + j * p.tap_step;
        const bool ok = srow_ok && (unsigned)(st + off) < (unsigned)p.Ta;
        return ok ? a_base + (unsigned)(off * p.lda * 4) : kOob;
    };
    float* st_a = smem + (tid >> 3) * 32 + ((c4 ^ ((tid >> 4) & 7)) << 2);  // stage 0; row r slot c at c ^ ((r >> 1) & 7)
    float* st_w = smem + A_STAGE + tid * 4;
    const int C = p.total_chunks;
    int ld_j = 0, ld_k = 0;  // load cursor: tap, k inside the tap
    unsigned a_voff = tap_voff(0), w_v = w_voff;
    i32x4 ra[D], rw[D];
    // Loads, LDS stores and operand reads are issued UNCONDITIONALLY (past the last chunk they carry out-of-range
    // offsets: the bounds check returns zeros without touching memory): a load behind a branch makes hipcc lose count of
    // what is in flight and wait vmcnt(0) before every LDS store -- the whole L2 latency once per chunk, measured 50 us
    // for tdnn3 at batch 1 against 78 us for the 32-row tiles and ~17 us of MFMA chain.  Only the MFMAs are guarded.
    auto issue = [&](i32x4& a, i32x4& w) {
        a = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, a_voff, ld_k * 4, 0);
        w = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_v, ((ld_j * p.Kc + ld_k) >> 2) * p.ldw * 16, 0);
        ld_k += BK;
        if (ld_k == p.Kc) {
            ld_k = 0;
            ++ld_j;
            const bool more = ld_j < p.taps;
            a_voff = more ? tap_voff(ld_j) : kOob;
            w_v = more ? w_voff : kOob;
        }
    };
    // A lane needs elements e and 2 + e of a 16-byte group (k = 4 h + e and 4 h + 2 + e).  The groups are staged with their
    // middle elements swapped -- (k0, k2, k1, k3) -- so that the pair is one aligned 8-byte read: half the LDS bytes of a
    // 16-byte read (the four waves of a block hit the LDS together after every barrier: with 16-byte reads it was busy
    // 320 of a chunk's ~770 cycles and the MFMA chain waited behind it) and no select instructions in front of the MFMAs.
    auto store = [&](const i32x4& a, const i32x4& w, int stage) {
        *reinterpret_cast<i32x4*>(st_a + stage * STAGE) = i32x4{a.x, a.z, a.y, a.w};
        *reinterpret_cast<i32x4*>(st_w + stage * STAGE) = i32x4{w.x, w.z, w.y, w.w};
    };
    // ---- compute role: wave (wm, wn) owns rows 16 wm .., columns 16 wn .. of the block
    const int wm = wid >> 1, wn = wid & 1;
    const int l16 = lane & 15, g = lane >> 4, h = g & 1, e = g >> 1;
    const int arow = 16 * wm + l16;
    const int sw = (arow >> 1) & 7;
    const float* rd_a = smem + arow * 32;
    const float* rd_w = smem + A_STAGE + (h * 32 + 16 * wn + l16) * 4;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // operands (k = 4 h + e, 4 h + 2 + e) of the chunk being multiplied and of the next one: two register sets that swap
    // roles every chunk (the set is a compile-time function of the unrolled ring slot: no copies)
    f32x2 oa[2][4], ow[2][4];
    const float* rd_ae = rd_a + 2 * e;
    const float* rd_we = rd_w + 2 * e;
    auto read_one = [&](f32x2& a, f32x2& w, int kg, int stage) {
        a = *reinterpret_cast<const f32x2*>(rd_ae + stage * STAGE + (((2 * kg + h) ^ sw) << 2));
        w = *reinterpret_cast<const f32x2*>(rd_we + stage * STAGE + kg * 2 * 32 * 4);
    };
    // prologue: D chunks in flight, chunks 0 and 1 staged, operands of chunk 0 in registers
#pragma unroll
    for (int j = 0; j < D; ++j) issue(ra[j], rw[j]);
    store(ra[0], rw[0], 0);
    issue(ra[0], rw[0]);
    store(ra[1], rw[1], 1);
    issue(ra[1], rw[1]);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) read_one(oa[0][kg], ow[0][kg], kg, 0);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    long long t_chain = 0, t_bar = 0, t_all0 = 0;  // SG_S16_TRACE (tuning aid): cycles of block 0, wave 0
    const bool tracing = TRACE && p.trace != nullptr && blockIdx.x == 0 && tid == 0;
    if (tracing) t_all0 = __builtin_readcyclecounter();
    // The loop runs to the next multiple of D and is free of branches: past the last chunk the staged operands are the
    // zeros of the out-of-range loads, and fmaf(0, 0, acc) == acc exactly (an accumulator that starts at +0 never becomes
    // -0: x + (-x) rounds to +0), so the surplus MFMAs change nothing.  One basic block lets hipcc keep exact counts of the
    // loads in flight AND lets every other instruction be placed behind an MFMA of the dependent chain (44 cycles each,
    // 32 of them busy), where it issues for free: the hand-over of chunk c + 2 from its ring slot to LDS and the slot's
    // refill first, then the LDS reads of the next chunk's operands (reading first made the four waves of a block collide in
    // the LDS right after the barrier and stretched the chain from 460 to 610 cycles).
    static_assert(D % 2 == 0, "the operand register set of a chunk is taken from its ring slot");
    for (int c0 = 0; c0 < C; c0 += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            // chunk c = c0 + j (c % 3 == j % 3, c % 2 == j % 2, c % D == j): its operands are in set j % 2.  The operands of
            // chunk c + 1 (staged during iteration c - 1, published by the barrier that ended it) go to the other set; chunk
            // c + 2 goes from its ring slot to the LDS stage chunk c - 1 was read from; the slot is refilled with chunk c + 2 + D.
            f32x2 (&ca)[4] = oa[j & 1];
            f32x2 (&cw)[4] = ow[j & 1];
            f32x2 (&na)[4] = oa[(j + 1) & 1];
            f32x2 (&nw)[4] = ow[(j + 1) & 1];
            const long long s0 = TRACE ? __builtin_readcyclecounter() : 0;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[0].x, cw[0].x, acc, 0, 0, 0);
            store(ra[(j + 2) % D], rw[(j + 2) % D], (j + 2) % 3);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[0].y, cw[0].y, acc, 0, 0, 0);
            issue(ra[(j + 2) % D], rw[(j + 2) % D]);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[1].x, cw[1].x, acc, 0, 0, 0);
            read_one(na[0], nw[0], 0, (j + 1) % 3);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[1].y, cw[1].y, acc, 0, 0, 0);
            read_one(na[1], nw[1], 1, (j + 1) % 3);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[2].x, cw[2].x, acc, 0, 0, 0);
            read_one(na[2], nw[2], 2, (j + 1) % 3);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[2].y, cw[2].y, acc, 0, 0, 0);
            read_one(na[3], nw[3], 3, (j + 1) % 3);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[3].x, cw[3].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[3].y, cw[3].y, acc, 0, 0, 0);
            const long long s1 = TRACE ? __builtin_readcyclecounter() : 0;
            // LDS-only barrier (__syncthreads() would also wait for every global load in flight).  Waiting only for the two
            // LDS writes (lgkmcnt(8): "the eight reads may still fly") is NOT safe: hipcc merges pairs of the reads into
            // ds_read2st64_b64, so fewer than eight instructions follow the writes and the count would pass too early.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (TRACE) {
                const long long s2 = __builtin_readcyclecounter();
                t_chain += s1 - s0;
                t_bar += s2 - s1;
            }
        }
    }
    if (tracing) {
        p.trace[0] = (unsigned long long)(__builtin_readcyclecounter() - t_all0);
        p.trace[1] = (unsigned long long)t_chain;
        p.trace[2] = (unsigned long long)t_bar;
        p.trace[3] = (unsigned long long)C;
    }
    const int m0 = bm0 + 16 * wm, col = bn0 + 16 * wn + l16;
    float bias = 0.f;
    if (EPI == EPI_BIAS_RELU) bias = p.bias[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ro = m0 + 4 * g + r;
        if (ro < p.M) {
            float v = acc[r];
            if (EPI == EPI_BIAS_RELU) v = fmaxf(v + bias, 0.f);
            if (EPI == EPI_RELU_MASK) v = p.mask[(size_t)ro * p.ldc + col] > 0.f ? v : 0.f;
            p.C[(size_t)ro * p.ldc + col] = v;
        }
    }
}

static hipError_t launch_s16(const ConvGemmArgs& a_in, int epi, hipStream_t s) {
    constexpr int R = 6;  // chunks in flight per thread (multiple of 3)
    ConvGemmArgs a = a_in;
    static const bool tr_on = sg_tune_env("SG_S16_TRACE") != nullptr;  // tuning aid: cycle split of block 0 (synchronises)
    static PerDeviceScratch tr_buf;
    unsigned long long* tr_dev = tr_on ? static_cast<unsigned long long*>(tr_buf.get(64)) : nullptr;
    a.trace = tr_dev;
    const int ntile32 = a.N / 32;
    dim3 grid(((a.M + 31) / 32) * ntile32);
#define SG_S16(EPI) \
    if (tr_on) hipLaunchKernelGGL((conv_gemm_s16_kernel<EPI, R, true>), grid, dim3(256), 0, s, a, ntile32); \
    else hipLaunchKernelGGL((conv_gemm_s16_kernel<EPI, R, false>), grid, dim3(256), 0, s, a, ntile32);
    switch (epi) {
        case EPI_NONE: SG_S16(EPI_NONE) break;
        case EPI_BIAS_RELU: SG_S16(EPI_BIAS_RELU) break;
        case EPI_RELU_MASK: SG_S16(EPI_RELU_MASK) break;
        default: return hipErrorInvalidValue;
    }
#undef SG_S16
    if (tr_on && tr_dev && hipStreamSynchronize(s) == hipSuccess) {
        unsigned long long h[4];
        if (hipMemcpy(h, tr_dev, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[3])
            fprintf(stderr, "s16 M=%d N=%d chunks %llu: %llu cycles in all, per chunk %.0f = chain section %.0f + wait/barrier %.0f + rest %.0f\n",
                    a.M, a.N, h[3], h[0], (double)h[0] / h[3], (double)h[1] / h[3], (double)h[2] / h[3],
                    ((double)h[0] - h[1] - h[2]) / h[3]);
    }
    return hipGetLastError();
}

template <int MI>
static hipError_t launch_tile_q_mi(const ConvGemmArgs& a, int epi, hipStream_t s) {
    const int mtiles = (a.M + 32 * MI - 1) / (32 * MI), ntiles = a.N / 128;
    dim3 grid(mtiles * ntiles);
    switch (epi) {
        case EPI_NONE: hipLaunchKernelGGL((conv_gemm_q_kernel<EPI_NONE, MI>), grid, dim3(256), 0, s, a, mtiles, ntiles); break;
        case EPI_BIAS_RELU: hipLaunchKernelGGL((conv_gemm_q_kernel<EPI_BIAS_RELU, MI>), grid, dim3(256), 0, s, a, mtiles, ntiles); break;
        case EPI_RELU_MASK: hipLaunchKernelGGL((conv_gemm_q_kernel<EPI_RELU_MASK, MI>), grid, dim3(256), 0, s, a, mtiles, ntiles); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

static hipError_t launch_tile_q(const ConvGemmArgs& a, int epi, hipStream_t s) {
    // A nearly empty chip (batch <= 8: fewer 64-row tiles than CUs) waits for the sequential k chain of one tile;
    // 32-row tiles halve the MFMAs per wave and chunk, so the chain -- and the launch -- takes half as long.
    static const int small_rows = [] {
        const char* e = sg_tune_env("SG_TILE32");  // 0 = always 64-row tiles (tuning aid)
        return e ? atoi(e) : 1;
    }();
    const int cus = a.num_cus > 0 ? a.num_cus : 256;
    if (small_rows && a.force == 0 && a.total_chunks >= 8 && ((a.M + 63) / 64) * (a.N / 128) < cus) return launch_tile_q_mi<1>(a, epi, s);
    return launch_tile_q_mi<2>(a, epi, s);
}

// Persistent workers = resident blocks: 512 8-wave blocks (64 KB LDS, two per CU) or 256 16-wave blocks (96 KB, one
// per CU).  tools/sk_trace.py shows why the second exists: of two blocks sharing a CU the first-dispatched one runs
// ~1.3x faster (the SIMD arbiter favours the older waves; s_setprio did not change it) and the other finishes the
// kernel alone at ~60 % efficiency.  One block per CU removes the pair, and the 256-row tile stages 25 % fewer
// bytes per MAC.

template <int EPI, int KIND>
static void launch_streamk_kind(const ConvGemmArgs& a, int workers, int ntiles, int tiles, int ipw, float* slabs,
                                unsigned* flags, unsigned epoch, hipStream_t s) {
    constexpr int threads = sk_threads(KIND);
    constexpr size_t lds = sk_lds_bytes(KIND);
    // per device: remember for which devices the > 64 KB dynamic-LDS opt-in has been made (a process may hold one
    // sg_ctx per GPU)
    static std::atomic<unsigned long long> done_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done_mask.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_streamk_kernel<EPI, KIND>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        done_mask.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((conv_gemm_streamk_kernel<EPI, KIND>), dim3(workers), dim3(threads), lds, s, a, ntiles,
                       tiles, ipw, slabs, flags, epoch);
}

// resident blocks per CU the runtime admits for a stream-K kernel (all workers must be co-resident: a waiter spins on
// its predecessor); queried once per kind, with the dynamic-LDS opt-in applied first
template <int KIND>
static int streamk_blocks_per_cu() {
    static const int n = [] {
        constexpr size_t lds = sk_lds_bytes(KIND);
        const void* fn = reinterpret_cast<const void*>(conv_gemm_streamk_kernel<EPI_NONE, KIND>);
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_gemm_streamk_kernel<EPI_NONE, KIND>, sk_threads(KIND), lds) !=
            hipSuccess)
            nb = 0;
        return nb;
    }();
    return n;
}

// returns hipErrorNotSupported when the shape does not qualify (caller falls back to the tile launch)
static hipError_t launch_streamk(const ConvGemmArgs& a, int epi, float* slabs, unsigned* flags, hipStream_t s) {
    static const int w16 = [] {
        const char* e = sg_tune_env("SG_STREAMK_W16");  // 0 = two 8-wave 128x128 blocks per CU
        return e ? atoi(e) : 1;
    }();
    int kind = !a.Wq ? 0 : ((w16 && a.force != 3) ? 2 : 1);
    // one worker per resident slot of this device (slabs / flags are sized for 256 CUs)
    const int cus = a.num_cus > 0 && a.num_cus < 256 ? a.num_cus : 256;
    int bm = kind == 2 ? 256 : 128, workers = kind == 2 ? cus : 2 * cus;
    // Medium batches (B = 32 at 3 s: 136 tiles of 256 rows for 256 CUs): fewer 256-row tiles than CUs would leave the
    // persistent launch to the 4-wave tile kernel at ~100 TFLOP/s.  128-row tiles still give every CU one: run the
    // 8-wave 128x128 kernel with ONE block per CU (2 waves per SIMD; the CU's MFMA rate is what a lone block needs).
    static const int mid = [] {
        const char* e = sg_tune_env("SG_STREAMK_MID");  // 0 = off (tuning aid)
        return e ? atoi(e) : 1;
    }();
    static const int mid9 = [] {
        const char* e = sg_tune_env("SG_STREAMK_MID9");  // 1 = 128-row tiles as four 64 x 64 computing waves (kind 9) instead of kind 5
        return e ? atoi(e) : 0;
    }();
    if (mid && kind == 2 && a.force == 0 && ((a.M + 255) / 256) * (a.N / 128) < cus) {
        // deep = 1 (default): the wave-specialised kinds 5 / 6 / 7 (ws_stage_segment / ws_compute_segment); 0: the all-in-one kinds 1 / 3 / 4
        static const int deep = [] {
            const char* e = sg_tune_env("SG_STREAMK_DEEP");
            return e ? atoi(e) : 1;
        }();
        if (((a.M + 127) / 128) * (a.N / 128) >= cus) {
            kind = deep ? (mid9 ? 9 : 5) : 1;
            bm = 128;
            workers = cus;
        } else if (((a.M + 63) / 64) * (a.N / 128) >= cus) {
            // small batches (B = 16): 4-wave 64x128 blocks, one per CU -- balances the 272 tiles a one-block-per-tile
            // launch would spread as 240 x 1 + 16 x 2
            kind = deep ? 6 : 3;
            bm = 64;
            workers = cus;
        } else if (((a.M + 31) / 32) * (a.N / 128) >= cus) {
            kind = deep ? 7 : 4;  // batch 8: 32-row tiles, half the k chain per tile, balanced over the CUs
            bm = 32;
            workers = cus;
        }
    }
    // the full batch (>= 256 tiles of 256 rows): kind 8, the 256-row tile with the roles split between waves (round 3)
    static const int ws256 = [] {
        const char* e = sg_tune_env("SG_STREAMK_WS");  // 0 = the all-in-one 16-wave kernel (kind 2)
        return e ? atoi(e) : 1;
    }();
    // measured per layer at 64 utterances (profiles/r03_layers.txt): tdnn2 / tdnn3 (80 / 112 chunks per tile) gain 1-3 % from
    // the split, tdnn4 / tdnn5 (16 / 48 chunks) lose 4-8 %
    static const int ws256_min_chunks = [] {
        const char* e = sg_tune_env("SG_STREAMK_WS_MINCHUNKS");
        return e ? atoi(e) : 64;
    }();

    if (ws256 && kind == 2 && a.force == 0) {
        if (a.total_chunks >= ws256_min_chunks) {
            kind = 8;
        } else {
            // the K = 512 / 1536 layers of the full batch: 128-row tiles on the wave-specialised kind 5 (eight 64x32 computing
            // waves + four staging waves, one block per CU) -- 3-4 % faster than the 16-wave all-in-one 256-row kernel on each
            // of tdnn4 / tdnn5 both ways (profiles/r03_layers.txt)
            kind = 5;
            bm = 128;
            workers = cus;
        }
    }
    static const int env_kind = [] {
        const char* e = sg_tune_env("SG_STREAMK_KIND");  // tuning aid: 5 .. 9 = this wave-specialised kind wherever the shape qualifies
        return e ? atoi(e) : 0;
    }();
    if (env_kind >= 5 && env_kind <= 9 && a.force == 0 && a.Wq) {
        const int kind0 = kind, bm0 = bm, workers0 = workers;
        kind = env_kind;
        bm = kind == 5 || kind == 9 ? 128 : kind == 6 ? 64 : kind == 7 ? 32 : 256;
        workers = cus;
        const long tl = (long)((a.M + bm - 1) / bm) * (a.N / 128);
        if (tl < workers || (tl * a.total_chunks + workers - 1) / workers < a.total_chunks) {  // does not qualify: leave the choice alone
            kind = kind0;
            bm = bm0;
            workers = workers0;
        }
    }
    if (a.force >= 6 && a.force <= 10 && a.Wq) {  // parity tests: the wave-specialised kinds on any shape that qualifies
        kind = a.force - 1;
        bm = kind == 5 || kind == 9 ? 128 : kind == 6 ? 64 : kind == 7 ? 32 : 256;
        workers = cus;
    }
    const int mtiles = (a.M + bm - 1) / bm, ntiles = a.N / 128;
    const int tiles = mtiles * ntiles;
    const long total = (long)tiles * a.total_chunks;
    const int ipw = (int)((total + workers - 1) / workers);
    // every range must span at least one full tile's worth of chunks, so a tile is shared by at most
    // two workers and never lies strictly inside one range.
    // Very short K (tdnn1 forward: 5 chunks) is faster one block per tile (measured).
    static const int min_chunks = [] {
        const char* e = sg_tune_env("SG_STREAMK_MINCHUNKS");
        return e ? atoi(e) : 16;  // >= 16 chunks (K >= 512): tdnn4 / tdnn5 gain 4-12 %, tdnn1 (5 chunks) loses
    }();
    if (!slabs || !flags || tiles < workers || ipw < a.total_chunks || a.total_chunks < min_chunks) return hipErrorNotSupported;
    // every worker has to be resident at once; if the runtime would admit fewer blocks than that, use the tile launch
    const int per_cu = kind == 2 ? streamk_blocks_per_cu<2>() : kind == 1 ? streamk_blocks_per_cu<1>()
                     : kind == 3 ? streamk_blocks_per_cu<3>() : kind == 4 ? streamk_blocks_per_cu<4>()
                     : kind == 5 ? streamk_blocks_per_cu<5>() : kind == 6 ? streamk_blocks_per_cu<6>()
                     : kind == 7 ? streamk_blocks_per_cu<7>() : kind == 8 ? streamk_blocks_per_cu<8>() : kind == 9 ? streamk_blocks_per_cu<9>() : streamk_blocks_per_cu<0>();
    if ((long)per_cu * cus < workers) return hipErrorNotSupported;
    if ((size_t)workers * bm * 128 > (size_t)512 * 128 * 128) return hipErrorNotSupported;  // slab capacity (sg_api.hip)
    static std::atomic<unsigned> launch_counter{0};
    unsigned epoch = ++launch_counter;
    if (epoch == 0) epoch = ++launch_counter;  // 0 is the value of never-written flags
    static const char* trace_file = sg_tune_env("SG_SK_TRACE");  // tuning aid: dump per-worker phase timestamps
    // ... of the launches [SG_SK_TRACE_SKIP, SG_SK_TRACE_SKIP + SG_SK_TRACE_COUNT) only (default: all of them): a traced launch
    // is followed by a synchronisation, so tracing every launch never sees the chip at the clock of a long loop
    static const long trace_skip = [] {
        const char* e = sg_tune_env("SG_SK_TRACE_SKIP");
        return e ? atol(e) : 0L;
    }();
    static const long trace_count = [] {
        const char* e = sg_tune_env("SG_SK_TRACE_COUNT");
        return e ? atol(e) : (1L << 60);
    }();
    static std::atomic<long> trace_seen{0};
    static PerDeviceScratch trace_buf;
    bool trace_this = trace_file != nullptr;
    if (trace_this) {
        const long idx = trace_seen.fetch_add(1, std::memory_order_relaxed);
        trace_this = idx >= trace_skip && idx - trace_skip < trace_count;
    }
    unsigned long long* trace_dev = trace_this ? static_cast<unsigned long long*>(trace_buf.get((size_t)workers * 16 * 8)) : nullptr;
    ConvGemmArgs at = a;
    at.trace = trace_dev;
    if (at.lose_counter && *at.lose_counter > 0) {  // fault injection: THIS stream-K launch publishes no hand-off flags
        at.ablate |= 8;
        --*at.lose_counter;
    }
    at.lose_counter = nullptr;
    dim3 grid(workers);
#define a at
#define SG_SK(EPI)                                                                                          \
    if (kind == 9) launch_streamk_kind<EPI, 9>(a, workers, ntiles, tiles, ipw, slabs, flags, epoch, s);       \
    else if (kind == 8) launch_streamk_kind<EPI, 8>(a, workers, ntiles, tiles, ipw, slabs, flags, epoch, s);  \
    else if (kind == 7) launch_streamk_kind<EPI, 7>(a, workers, ntiles, tiles, ipw, slabs, flags, epoch, s);  \
    else if (kind == 6) launch_streamk_kind<EPI, 6>(a, workers, ntiles, tiles, ipw, slabs, flags, epoch, s);  \
    else if (kind == 5) launch_streamk_kind<EPI, 5>(a, workers, ntiles, tiles, ipw, slabs, flags, epoch, s);  \
    else if (kind == 4) launch_streamk_kind<EPI, 4>(a, workers, ntiles, tiles, ipw, slabs, flags, epoch, s);  \
    else if (kind == 3) launch_streamk_kind<EPI, 3>(a, workers, ntiles, tiles, ipw, slabs, flags, epoch, s);  \
    else if (kind == 2) launch_streamk_kind<EPI, 2>(a, workers, ntiles, tiles, ipw, slabs, flags, epoch, s);  \
    else if (kind == 1) launch_streamk_kind<EPI, 1>(a, workers, ntiles, tiles, ipw, slabs, flags, epoch, s);  \
    else launch_streamk_kind<EPI, 0>(a, workers, ntiles, tiles, ipw, slabs, flags, epoch, s);
    switch (epi) {
        case EPI_NONE: SG_SK(EPI_NONE) break;
        case EPI_BIAS_RELU: SG_SK(EPI_BIAS_RELU) break;
        case EPI_RELU_MASK: SG_SK(EPI_RELU_MASK) break;
        default: return hipErrorInvalidValue;
    }
#undef SG_SK
#undef a
    if (trace_this && trace_dev) {
        std::vector<unsigned long long> h((size_t)workers * 16);
        if (hipStreamSynchronize(s) == hipSuccess &&
            hipMemcpy(h.data(), trace_dev, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE* f = fopen(trace_file, "ab")) {
                const int hdr[8] = {workers, at.M, at.N, at.total_chunks, ipw, tiles, epi, 0};
                fwrite(hdr, sizeof(hdr), 1, f);
                fwrite(h.data(), 8, h.size(), f);
                fclose(f);
            }
        }
    }
    return hipGetLastError();
}

// One-block-per-tile launches use 64 x 128 tiles: 48 KB of LDS lets three blocks share a CU (3 waves per SIMD),
// which hides the per-chunk staging/barrier bubble; 96- and 128-row 4-wave tiles measured 12 % and 22 % slower
// (profiles/r01_tile_rows_sweep.txt) and were removed.
__global__ void pack_k4_kernel(const float* __restrict__ w, int K, int N, float* __restrict__ wq) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)K * N) return;
    const int k = (int)(i / N), n = (int)(i % N);
    wq[((size_t)(k >> 2) * N + n) * 4 + (k & 3)] = w[i];
}

hipError_t launch_pack_k4(const float* w, int K, int N, float* wq, hipStream_t s) {
    const size_t n = (size_t)K * N;
    hipLaunchKernelGGL(pack_k4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, K, N, wq);
    return hipGetLastError();
}

int conv_gemm_tile_rows(int M, int N) {
    (void)M;
    (void)N;
    return 64;
}

hipError_t launch_conv_gemm(const ConvGemmArgs& a_in, int tile, int epi, int splits, hipStream_t s) {
#ifdef SG_EXP_ABLATE  // timing experiments (results become wrong): never in the shipped library
    static const int ablate = [] {
        const char* e = sg_tune_env("SG_ABLATE");
        return e ? atoi(e) : 0;
    }();
#else
    constexpr int ablate = 0;
#endif
    static const int use_streamk = [] {
        const char* e = sg_tune_env("SG_STREAMK");  // 0 = always one block per tile
        return e ? atoi(e) : 1;
    }();
    ConvGemmArgs a = a_in;
    a.ablate |= ablate;  // (a_in.ablate: the fault-injection bit of sg_debug_lose_handoffs)
    static const int sk_xcd = [] {
        const char* e = sg_tune_env("SG_STREAMK_XCD");  // tuning aid, see the kernel
        return e ? atoi(e) : 2;
    }();
    a.sk_xcd = sk_xcd;
    static const int use_quad = [] {
        const char* e = sg_tune_env("SG_QUADFEED");  // 0 = b32-fed 8-wave kernel even when packed weights exist
        return e ? atoi(e) : 1;
    }();
    if (!use_quad || a.force == 2 || (a.ldw % 4) || (a.Kc % 4)) a.Wq = nullptr;
    if (a.Kc % BK != 0 || a.M <= 0 || a.Tc <= 0) return hipErrorInvalidValue;
    // The staging loads address A and W through 32-bit buffer descriptors (num_records, per-lane byte offsets), and
    // the "this tap row is outside the utterance" marker is the offset 0x80000000, which must stay out of range:
    // an operand of 2 GiB or more is refused (the API layer reports the largest batch that fits, see check_dims).
    const size_t a_bytes = (size_t)(a.M / a.Tc) * a.Ta * a.lda * sizeof(float);
    const size_t w_bytes = (size_t)a.taps * a.Kc * a.ldw * sizeof(float);
    if (a_bytes >= 0x80000000ull || w_bytes >= 0x80000000ull) return hipErrorInvalidValue;
    a.a_bytes = (unsigned)a_bytes;
    a.w_bytes = (unsigned)w_bytes;
    // batch 1-4: so few 16 x 16 output blocks that every one can have a SIMD (almost) to itself
    static const long s16_max = [] {
        const char* e = sg_tune_env("SG_S16_MAX_BLOCKS");  // 0 = never (tuning aid)
        return e ? atol(e) : 2800L;
    }();
    if ((tile == 0 || tile == 2) && splits == 1 && a.Wq && (a.N % 32) == 0 &&
        (a.force == 5 || (a.force == 0 && a.total_chunks >= 4 && (long)((a.M + 15) / 16) * (a.N / 16) <= s16_max)))
        return launch_s16(a, epi, s);
    switch (tile) {
        case 0: {
            if (a.N % 128) return hipErrorInvalidValue;
            if (splits == 1 && use_streamk && !a.no_streamk && a.force != 1 && a.force != 4) {
                const hipError_t e = launch_streamk(a, epi, a.sk_slabs, a.sk_flags, s);
                if (e != hipErrorNotSupported) return e;
                if (a.force >= 6) return hipErrorInvalidValue;  // a forced wave-specialised kind never silently becomes a tile launch
            }
            if (a.Wq && splits == 1 && a.force != 1) return launch_tile_q(a, epi, s);
            return launch_tile<64, 128, 2, 2>(a, epi, splits, s);
        }
        case 1:
            if (a.N % 32) return hipErrorInvalidValue;
            return launch_tile<128, 32, 4, 1>(a, epi, splits, s);
        case 2:
            if (a.N % 128) return hipErrorInvalidValue;
            if (a.Wq && splits == 1 && a.force != 1) return launch_tile_q(a, epi, s);
            return launch_tile<64, 128, 2, 2>(a, epi, splits, s);
        default:
            return hipErrorInvalidValue;
    }
}

}  // namespace sg
