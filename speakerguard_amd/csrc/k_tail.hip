// Per-utterance tail of the x-vector system, forward and backward in one workgroup:
//   fc1 output (split-K partials + folded bias)          xvecTDNN.py:63
//   - emb_mean                                            xvector_extract.py:42 / iv_plda.py:419
//   LDA affine  A[:, :512] e + A[:, 512]                  iv_plda.py:423-435
//   length normalisation to sqrt(D) (norm is DETACHED: .item() at xvector_extract.py:33)
//   PLDA transform P (e - mu), normalisation factor sqrt(D / sum e^2/(psi+1))   plda.py:73-97
//   PLDA log-likelihood-ratio against each enrolled speaker                      plda.py:140-190
//   decision = argmax, rejected (-1) unless max > threshold                      iv_plda.py:182-194
//   loss (attack/utils.py:7-102) and d loss/d scores, then the chain back to d loss/d fc1-output.
// One block of 1024 threads per utterance; reductions are DPP wave sums + a 16-entry LDS pass.
// The four matrix-vector products are bound by the LATENCY of the (L2-resident) matrix loads, not by bandwidth
// (1.2 MB per utterance): un-unrolled they issued one dependent load at a time (157 us for the kernel), unrolled
// 8-16x with 256 threads 97 us; with the K range of every product split over the four 256-thread quarters of the block
// 35 us; now every thread owns four adjacent columns (16-byte loads) of one of 8-20 k slices (matvec_cols4): 26 us.
// Partial sums are combined in a fixed order through LDS.
#include <cstdio>
#include <cstdlib>

#include "loss_device.h"
#include "sg_internal.h"

namespace sg {

constexpr int kMaxD = 512;
constexpr int kMaxS = kLossMaxS;

struct TailModelDev {
    const float *fc1_b, *emb_mean, *lda, *lda_t, *plda_mean, *plda_p, *plda_pt, *plda_psi, *enroll, *pa;
    int D, S;
    int Dp;  // D rounded up to a multiple of 4: row stride of lda_t / plda_p / plda_pt (zero-padded); lda / pa rows are kLdaLd long
    float threshold, logdet_given, logdet_without;
};

constexpr int kTailThreads = 1024;
constexpr int kTailParts = kTailThreads / 256;
constexpr int kTailEnrCache = 4096;  // floats of enrolled embeddings kept in LDS (S x D <= this: 10 x 200 in the recipes)
static_assert(kTailThreads == 2 * kEmb && kTailParts >= 2, "step 1 splits the fc1 slabs over two half-blocks");

// Sum over the 64 lanes, the same value in every lane: DPP butterfly inside the rows of 16 lanes, the four row sums
// through scalar registers (k_mfcc.hip wave_sum) -- no LDS round trips.
__device__ __forceinline__ float tail_wave_sum(float v) {
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

// Block sum: sixteen wave sums, combined in wave order.  Ends with a barrier; valid on every thread.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = tail_wave_sum(v);
    __syncthreads();  // (red may still be read by an earlier phase)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < kTailThreads / 64; ++i) r += red[i];
    return r;
}

// out[n] = sum_k W[k * ld + n] * v[k] for n < 4 N4 (v, out, part in LDS; ld a multiple of 4 floats, W 16-byte aligned; columns
// past the matrix's own width are zero padding).  A thread owns FOUR adjacent columns (one 16-byte load per row) and one
// of NP = 1024 / N4 slices of the k range, so a product is 10-26 independent 16-byte loads per thread, all in flight at
// once, where the first version (one column per thread, four k slices) issued 50-128 dependent-batch 4-byte loads: the
// four products of a tail went from 20.9 to ~11 us (the 1.1 MB of matrices cross one CU's L1 at 64 B/clk: 7.4 us is the
// floor).  The NP partial sums of a column are combined in slice order.  Ends with a barrier.
constexpr int kTailPart = 4096;  // floats: NP x 4 N4 <= 1024 x 4
__device__ __forceinline__ void matvec_cols4(const float* __restrict__ W, int ld, int K, int N4, const float* v, float* out,
                                             float* part) {
    const int NP = kTailThreads / N4, N = 4 * N4;
    const int p = threadIdx.x / N4, cg = threadIdx.x - p * N4;
    const int R = (K + NP - 1) / NP, k0 = p * R;
    if (p < NP) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* col = W + 4 * cg;
        // rows past the slice / the matrix are clamped to a valid row and weighted 0: no load sits behind a branch
#pragma unroll 16
        for (int j = 0; j < R; ++j) {
            const int k = k0 + j, kk = min(k, K - 1);
            const float4 w = *reinterpret_cast<const float4*>(col + (size_t)kk * ld);
            const float x = k < K ? v[kk] : 0.f;
            acc.x += w.x * x;
            acc.y += w.y * x;
            acc.z += w.z * x;
            acc.w += w.w * x;
        }
        *reinterpret_cast<float4*>(part + p * N + 4 * cg) = acc;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += kTailThreads) {
        float r = part[n];
        for (int i = 1; i < NP; ++i) r += part[i * N + n];
        out[n] = r;
    }
    __syncthreads();
}

// Touches every 128-byte line of the back-end matrices once (see step 1 of the kernel).  A CU pulls only ~11 B/cycle
// from HBM (MI355X_MICROARCH.md, prologue burst), so the blocks that share an XCD (blockIdx % 8 -- a speed assumption
// only) split the lines between them: block b takes every nq-th line starting at (b / 8) % nq, nq = blocks per XCD (<= 8).
// Independent, clamped loads: all of a thread's lines are in flight at once.  Small batches get HELPER blocks (blockIdx >=
// B, launched up to 64 in all) that do nothing else, so that one utterance does not pull the 1.2 MB alone (11 -> 4 us).
__device__ __forceinline__ float touch_matrices(const TailModelDev& m, int want_grad, int b, int tid) {
    constexpr int NT = kTailThreads;
    const int D = m.D;
    const int nq = min(8, max(1, ((int)gridDim.x + 7) >> 3)), q = (b >> 3) % nq;
    const size_t n_ldat = (size_t)(kEmb + 1) * m.Dp, n_pa = (size_t)D * kLdaLd, n_p = (size_t)D * m.Dp;
    const size_t step = (size_t)NT * nq * 32, first = ((size_t)tid * nq + q) * 32;
    // Independent, clamped loads: the first line of every matrix for every thread is in flight at once
    const float pa = m.lda_t[min(first, n_ldat - 1)], pb = want_grad ? m.pa[min(first, n_pa - 1)] : 0.f;
    const float pc = m.plda_pt[min(first, n_p - 1)];
    float sink = 0.f;
    for (size_t j = first + step; j < n_ldat; j += step) sink += m.lda_t[j];
    for (size_t j = first + step; j < n_p; j += step) sink += m.plda_pt[j];
    if (want_grad)
        for (size_t j = first + step; j < n_pa; j += step) sink += m.pa[j];
    sink += (pa + pb) + pc;
    return sink;
}

// Round 6: the tail is a chain of a dozen short dependent phases, and what it waited for between the four matrix-vector
// products was latency, not work (phase times of block 0, SG_TAIL_TRACE: 29 us in all, 12.5 of them in the products):
//  * the per-dimension vectors (psi, PLDA mean, LDA offset row) and the enrolled embeddings are fetched into LDS at the
//    very start, together with the fc1 slabs -- every later phase that read them from global memory paid an L2 round
//    trip of its own (~1 us each: normalisation, factor, scores, d scores);
//  * block sums and the per-speaker dot products reduce on DPP instead of six ds_bpermute round trips each;
//  * decision + loss + d loss / d scores run on ONE WAVE with the classes across the lanes (loss_and_dscores_wave: the 2 S
//    expf of the cross-entropy side by side, the sum still in index order) -- one thread walking them was 3.5-4.5 us;
//  * the backward applies P^T and LDA^T as ONE product with the matrix (P A) folded at load time (sg_xv_load, in float64,
//    rounded once): d e1 = ratio (P A)^T dv, the product autograd computes as A^T (ratio (P^T dv)).
__global__ __launch_bounds__(kTailThreads) void tail_kernel(TailModelDev m, const float* __restrict__ fc1_part, int nsplit, int B,
                                                   const int64_t* __restrict__ y, sg_loss_spec ls, int want_grad,
                                                   float* __restrict__ tdnn_emb, float* __restrict__ emb_out,
                                                   float* __restrict__ scores_out, int64_t* __restrict__ dec_out,
                                                   float* __restrict__ loss_out, float* __restrict__ demb,
                                                   float* __restrict__ loss_trace, int64_t* __restrict__ dec_trace,
                                                   uint8_t* __restrict__ success, unsigned long long* __restrict__ trace, int coef_rows) {
#define TSTAMP(i) if (trace && threadIdx.x == 0 && blockIdx.x == 0) trace[i] = __builtin_amdgcn_s_memrealtime();
    TSTAMP(0)
    __shared__ float e1[kEmb];
    __shared__ float e2[kMaxD], e4[kMaxD], e5[kMaxD], dv[kMaxD];
    __shared__ float c_psi[kMaxD], c_pmean[kMaxD], c_off[kMaxD];
    __shared__ float c_enr[kTailEnrCache];
    __shared__ float sc[kMaxS], dsc[kMaxS];
    __shared__ float red[kTailThreads / 64];
    __shared__ int64_t c_y;
    __shared__ __attribute__((aligned(16))) float part[kTailPart];
    constexpr int NT = kTailThreads;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= B) {  // helper block of a small batch: warm the L2 of an XCD that has an utterance to serve, nothing else
        if ((b & 7) < B) {
            const float sink = touch_matrices(m, want_grad, b, tid);
            if (sink == 1.2345e38f && demb) demb[0] = sink;  // never true: keeps the loads alive
        }
        return;
    }
    const int D = m.D, S = m.S;
    const float sqrtD = sqrtf((float)D);
    const bool enr_lds = (size_t)S * D <= (size_t)kTailEnrCache;
    const float* enr = enr_lds ? c_enr : m.enroll;
    // 1. fc1 output = sum of the split-K slabs (+ folded bias), global-mean subtraction.  The block's two halves take
    //    half of the slabs each, all loads issued before the first add; halves are added in order.
    //    Interleaved with it: the matrices of the products below (1.2 MB at D = 200) were evicted from every XCD's
    //    L2 by the 2.6 ms of contractions since the last tail launch, and a mat-vec is a chain of dependent load
    //    batches, each of which would pay an HBM / Infinity-Cache round trip (measured 7.6 + 4.0 + 4.0 + 8.7 us for the
    //    four products).  Every 128-byte line of them is touched once here, AFTER the slab loads were issued (vmcnt
    //    retires in order): the misses overlap each other and the slab sum, the products then hit in L2 (14 us).
    {
        constexpr int kHalf = kFc1SplitK / 2;
        const int i = tid & (kEmb - 1), h = tid >> 9;
        float sv[kHalf];
        float v = 0.f;
        if (nsplit == kFc1SplitK) {
#pragma unroll
            for (int z = 0; z < kHalf; ++z) sv[z] = fc1_part[((size_t)(h * kHalf + z) * B + b) * kEmb + i];
        }
        // the small per-dimension vectors, the enrolled embeddings, the label and this thread's bias / mean entries: once,
        // all in flight together with the slabs, for every later phase
        const int dd = min(tid, D - 1);
        const float v_psi = m.plda_psi[dd], v_pm = m.plda_mean[dd], v_off = m.lda_t[(size_t)kEmb * m.Dp + dd];
        const float v_b = m.fc1_b[i], v_mean = m.emb_mean[i];
        const int64_t v_y = y ? y[b] : 0;
        float v_enr[kTailEnrCache / NT];
#pragma unroll
        for (int z = 0; z < kTailEnrCache / NT; ++z) v_enr[z] = enr_lds ? m.enroll[min(tid + z * NT, S * D - 1)] : 0.f;
        const float sink = touch_matrices(m, want_grad, b, tid);
        if (nsplit == kFc1SplitK) {
#pragma unroll
            for (int z = 0; z < kHalf; ++z) v += sv[z];
        } else if (h == 0) {
            for (int z = 0; z < nsplit; ++z) v += fc1_part[((size_t)z * B + b) * kEmb + i];
        }
        if (h == 1) part[i] = v;
        if (tid < kMaxD) {
            c_psi[tid] = v_psi;
            c_pmean[tid] = v_pm;
            c_off[tid] = v_off;
        }
        if (tid == 0) c_y = v_y;
#pragma unroll
        for (int z = 0; z < kTailEnrCache / NT; ++z) c_enr[tid + z * NT] = v_enr[z];
        if (sink == 1.2345e38f && demb) demb[0] = sink;  // never true: keeps the prefetch loads alive
        __syncthreads();
        if (h == 0) {  // halves added in order, then the folded bias, then the global mean (xvector_extract.py:42)
            float e = v + part[i];
            e += v_b;
            if (tdnn_emb) tdnn_emb[(size_t)b * kEmb + i] = e;
            e1[i] = e - v_mean;
        }
    }
    __syncthreads();
    TSTAMP(1)
    // 2. LDA (the offset column of the (D, 513) matrix is row kEmb of the transposed copy)
    matvec_cols4(m.lda_t, m.Dp, kEmb, m.Dp / 4, e1, e2, part);
    float n2 = 0.f;
    for (int d = tid; d < D; d += NT) {
        const float acc = e2[d] + c_off[d];
        e2[d] = acc;
        n2 += acc * acc;
    }
    TSTAMP(2)
    // 3. length normalisation (ratio is a constant for the backward pass)
    const float ratio = sqrtD / sqrtf(block_sum(n2, red));
    for (int d = tid; d < D; d += NT) e2[d] = e2[d] * ratio - c_pmean[d];
    __syncthreads();
    TSTAMP(3)
    // 4. PLDA transform + normalisation factor
    matvec_cols4(m.plda_pt, m.Dp, D, m.Dp / 4, e2, e4, part);
    TSTAMP(4)
    float qp = 0.f;
    for (int d = tid; d < D; d += NT) qp += e4[d] * e4[d] / (c_psi[d] + 1.f);
    const float q = block_sum(qp, red);
    const float fac = sqrtf((float)D / q);
    for (int d = tid; d < D; d += NT) {
        e5[d] = e4[d] * fac;
        if (emb_out) emb_out[(size_t)b * D + d] = e5[d];
    }
    __syncthreads();
    TSTAMP(5)
    // 5. PLDA scores: one wave per enrolled speaker
    {
        const int lane = tid & 63, wid = tid >> 6;
        const float l2pi = logf(2.f * 3.1415926f) * (float)D;  // plda.py:179 (cancels in the ratio)
        float s0 = 0.f;  // sum e5^2 / (psi + 1)
        for (int d = lane; d < D; d += 64) s0 += e5[d] * e5[d] * (1.f / (c_psi[d] + 1.f));
        s0 = tail_wave_sum(s0);
        const float without = -0.5f * (m.logdet_without + l2pi + s0);
        for (int s = wid; s < S; s += NT / 64) {
            float s1 = 0.f;
            for (int d = lane; d < D; d += 64) {
                const float psi = c_psi[d];
                const float r = psi / (psi + 1.f);
                const float df = e5[d] - r * enr[(size_t)s * D + d];
                s1 += df * df * (1.f / (1.f + r));
            }
            s1 = tail_wave_sum(s1);
            if (lane == 0) {
                const float given = -0.5f * (m.logdet_given + l2pi + s1);
                const float v = given - without;
                sc[s] = v;
                dsc[s] = 0.f;
                if (scores_out) scores_out[(size_t)b * S + s] = v;
            }
        }
    }
    __syncthreads();
    TSTAMP(6)
    // 6-8. decision, loss, d loss / d scores
    if (S <= 64) {
        if (tid < 64) {  // one wave, the classes across its lanes; nobody else touches sc / dsc / part until the barrier
            int64_t dec = 0;
            const float loss = loss_and_dscores_wave(sc, dsc, part, S, m.threshold, c_y, y != nullptr, ls, &dec, tid,
                                                     ls.coef_dev ? ls.coef_dev + (size_t)(coef_rows > 0 ? b % coef_rows : b) * S : nullptr);
            if (tid == 0) {
                if (dec_out) dec_out[b] = dec;
                if (dec_trace) dec_trace[b] = dec;
                if (y && success) success[b] = ls.targeted ? (dec == c_y) : (dec != c_y);
                if (loss_out) loss_out[b] = loss;
                if (loss_trace) loss_trace[b] = loss;
            }
        }
    } else {
        int64_t dec = 0;  // scratch: `part` (kTailPart floats >= kMaxS) and `red` (16 floats), both idle here
        const float loss = loss_and_dscores_block(sc, dsc, part, red, S, m.threshold, y ? y[b] : 0, y != nullptr, ls, &dec, tid, NT,
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
    TSTAMP(7)
    if (!want_grad || !demb) return;
    // 9. d/d e5 of the score combination
    float dotp = 0.f;
    for (int d = tid; d < D; d += NT) {
        const float psi = c_psi[d];
        const float r = psi / (psi + 1.f);
        const float iv1 = 1.f / (1.f + r), iv0 = 1.f / (psi + 1.f);
        float g = 0.f;
        for (int s = 0; s < S; ++s) {
            const float w = dsc[s];
            if (w != 0.f) g += w * (-(e5[d] - r * enr[(size_t)s * D + d]) * iv1 + e5[d] * iv0);
        }
        dv[d] = g;
        dotp += g * e4[d];
    }
    TSTAMP(8)
    // 10. through the PLDA normalisation factor (differentiable, plda.py:92-97), and the length-norm ratio
    const float dot = block_sum(dotp, red);
    for (int d = tid; d < D; d += NT) dv[d] = (fac * dv[d] - dot * (fac / q) * e4[d] / (c_psi[d] + 1.f)) * ratio;
    __syncthreads();
    TSTAMP(9)
    // 11-13. (P A)^T: P^T and LDA^T in one product -> d loss / d fc1 output
    TSTAMP(10)
    matvec_cols4(m.pa, kLdaLd, D, kEmb / 4, dv, e1, part);
    TSTAMP(11)
    for (int i = tid; i < kEmb; i += NT) demb[(size_t)b * kEmb + i] = e1[i];
    TSTAMP(12)
#undef TSTAMP
}

hipError_t launch_tail(const TailArgs& a, hipStream_t s) {
    const XvModel& x = *a.m;
    const int S = x.enroll_override ? x.S_override : x.S;  // per-call enroll_embs= (iv_plda.py:155-165)
    if (x.D > kMaxD || S > kMaxS || S < 1) return hipErrorInvalidValue;
    TailModelDev m;
    m.fc1_b = x.fc1_b; m.emb_mean = x.emb_mean; m.lda = x.lda; m.lda_t = x.lda_t; m.plda_mean = x.plda_mean;
    m.plda_p = x.plda_p; m.plda_pt = x.plda_pt; m.plda_psi = x.plda_psi; m.pa = x.pa;
    m.enroll = x.enroll_override ? x.enroll_override : x.enroll;
    m.D = x.D; m.Dp = x.Dp; m.S = S; m.threshold = x.threshold;
    m.logdet_given = x.logdet_given; m.logdet_without = x.logdet_without;
    // tuning aid: SG_TAIL_TRACE=1 prints the phase timestamps (100 MHz) of block 0 after every launch (synchronises)
    static const bool tr_on = sg_tune_env("SG_TAIL_TRACE") != nullptr;
    static PerDeviceScratch tr_buf;
    unsigned long long* tr_dev = tr_on ? static_cast<unsigned long long*>(tr_buf.get(16 * 8)) : nullptr;
    const int grid = a.B < 64 ? 64 : a.B;  // small batches: helper blocks up to one full set of 8 per XCD (see touch_matrices)
    hipLaunchKernelGGL(tail_kernel, dim3(grid), dim3(kTailThreads), 0, s, m, a.fc1_part, a.nsplit, a.B, a.y, a.loss, a.want_grad,
                       a.tdnn_emb, a.emb, a.scores, a.decisions, a.loss_out, a.demb, a.loss_trace, a.decision_trace,
                       a.success, tr_on ? tr_dev : nullptr, a.coef_rows);
    if (tr_on && tr_dev && hipStreamSynchronize(s) == hipSuccess) {
        unsigned long long h[16];
        if (hipMemcpy(h, tr_dev, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "tail phases (us):");
            for (int i = 1; i <= 12; ++i) fprintf(stderr, " %d:%.2f", i, (double)(h[i] - h[i - 1]) * 0.01);
            fprintf(stderr, "  total %.2f\n", (double)(h[12] - h[0]) * 0.01);
        }
    }
    return hipGetLastError();
}

}  // namespace sg
