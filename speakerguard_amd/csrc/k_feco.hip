// FeCo feature-level defense: per-utterance k-means over the frames of a feature
// matrix, then every cluster replaced by the mean of its frames (reference defense/feature_level.py:168-217),
// forward and backward.
//
// The reference calls a third-party k-means with a RANDOM initialisation (libKMCUDA / kmeans_pytorch), so its
// cluster ids are not reproducible even between two runs of the reference.  The clustering here follows its own
// DETERMINISM CONTRACT (restated in oracle/feco.py, checked bit for bit):
//   * k = int(F * ratio) centroids, centroid j initialised to frame floor(j * F / k) -- or, for the SEEDED form
//     (sg_feco_kmeans_seeded: what makes expectation-over-transformation against this defense meaningful, like
//     the reference's randomly initialised k-means), to the frame of rank j when the frames are ordered by
//     (Philox4x32-10(counter = (frame, 0, utterance lo, utterance hi), key = seed) word 0, frame) ascending:
//     k distinct frames, a uniformly random subset in random order, a function of (seed, GLOBAL utterance index)
//     only (oracle/philox.py feco_random_init);
//   * assignment: squared L2 distance accumulated over d = 0..D-1 in fp32 without FMA contraction, nearest
//     centroid wins, ties go to the lowest centroid index;
//   * stop when no assignment changed or after max_iter assignment steps; otherwise update: centroid j = (sum of its
//     frames in ascending frame order, fp32) / count, an empty cluster keeps its centroid;
//   * the ids of the LAST assignment step are the result.
// Given the ids, the compression is the reference's :204-216: mean of the cluster's frames, an empty cluster i
// falls back to frame i when `force` (batch > 1) and is dropped otherwise (the host compacts).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "sg_internal.h"
#include "philox.h"

#pragma clang fp contract(off)

using namespace sg;

namespace {

int feco_fail(sg_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

// tuning aid (SG_FECO_TRACE=1): phase timestamps (100 MHz) of block (0, 0): per iteration [start, after the assignment,
// after the member lists, after the update]
__device__ unsigned long long g_feco_trace[4 * 16 + 4];
__device__ int g_feco_trace_on;
#define FECO_STAMP(i) \
    if (g_feco_trace_on && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && (i) < 4 * 16 + 4) g_feco_trace[i] = __builtin_amdgcn_s_memrealtime();

constexpr int kFecoMaxD = 64;
__host__ __device__ constexpr int al4(int n) { return (n + 3) & ~3; }
typedef float float2v __attribute__((ext_vector_type(2)));
// centroid j, dimension d inside the pair-interleaved image (rows of 2 * DPAD floats)
#define CS_AT(j, d) ((size_t)((j) >> 1) * (2 * DPAD) + 2 * (size_t)(d) + ((j) & 1))

// (x.lo - c.lo, x.lo - c.hi) and (x.hi - c.lo, x.hi - c.hi): one v_pk_add_f32 each.  The operand-select bits broadcast
// one half of the x register pair to both lanes and the neg bits turn the add into an IEEE subtraction, so no
// register is spent on splats (hipcc builds them with v_mov pairs and spills at 1024 threads per block).
__device__ __forceinline__ float2v pk_sub_xlo(float2v x, float2v c) {
    float2v r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(c));
    return r;
}
__device__ __forceinline__ float2v pk_sub_xhi(float2v x, float2v c) {
    float2v r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(c));
    return r;
}

// One 1024-thread block per utterance.  Round 1 kept frames and centroids in LDS and gave every thread a frame whose
// D floats it re-read from LDS for every centroid: at D = 32 the 64 lanes of a wave hit ONE bank (stride 32 words) --
// 3.1 ms per call at 64 x 300 x 32, 80 % of a step against a FeCo-defended AudioNet.  Now:
//   * a thread keeps ITS frame in registers for the whole call (frames are read once, coalescing irrelevant), the
//     centroids sit in LDS padded to DPAD floats per row and are read as 16-byte broadcasts;
//   * the centroid range is cut into JC = 1024 / F chunks so that all 1024 threads work on the assignment (thread =
//     (frame, chunk)); chunk minima are merged in ascending centroid order with a strict <, i.e. the lowest index still
//     wins ties;
//   * the update walks per-cluster member lists (built by a stable counting pass: ascending frame order, as the
//     contract demands) instead of scanning all F ids for each of the k x D centroid entries.
//   * the assignment is VALU-bound on the one CU an utterance gets (F x k x D x {sub, mul, add}: 4.3 M lane-operations at
//     300 x 150 x 32 = 28 us per assignment step at 64 lanes per clock): centroids are kept in PAIRS, interleaved
//     per dimension ([pair][d][2]), and a thread measures its frame against both centroids of a pair with packed-fp32
//     instructions (v_pk_add_f32 / v_pk_mul_f32: two lane-operations per lane and clock) -- per centroid still
//     sub, mul, add in ascending d with separate roundings, so the distances are the same bits.
//   * what was left after that was LATENCY: the member lists were built by one thread per cluster walking all F ids
//     (one LDS round trip per id), their offsets by thread 0 walking all k counts, and the update summed its members
//     through a chain of dependent L2 loads -- together ~20 us of the ~30 us an iteration took.  Now the counts come
//     from LDS atomics (one per frame), every cluster adds up the counts below it with 16-byte broadcast reads, every
//     FRAME finds its own slot (number of earlier frames with its id: 16-byte reads again -- ascending frame order
//     inside a cluster by construction), and the frames are staged in LDS once so the update never leaves the CU
//     (utterances too long for that keep reading HBM / L2).
//   * the cluster means the reference takes next (feature_level.py:204-216) ARE the centroids of the last update --
//     same ids, same ascending sums, same division -- so the kernel can hand them out itself (out / counts, an empty
//     cluster i taking frame i): the separate compress launch walked all F ids per output element (28 us).
// Same arithmetic, same order, same ids as before (oracle/feco.py restates the contract; tests compare bit for bit).
// Dynamic LDS (every array 16-byte aligned): cs[ceil(k/2)][DPAD][2], ids[F], cnt[k], start[k + 1], members[F], pd[2048],
// pj[2048], xs[F][D] (when it fits).
template <int DPAD>
__global__ __launch_bounds__(1024) void feco_kmeans_kernel(const float* __restrict__ feats, int F, int D, int k,
                                                           int max_iter, int seeded, uint64_t seed, int64_t index_base,
                                                           int x_in_lds, int* __restrict__ assign, float* __restrict__ out,
                                                           int* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int npair = (k + 1) >> 1;
    float* cs = lds;                                             // [npair][DPAD][2], pad columns / odd partner zero
    int* ids = reinterpret_cast<int*>(cs + (size_t)npair * 2 * DPAD);   // [F]
    int* cnt = ids + al4(F);                                     // [k]
    int* start = cnt + al4(k);                                   // [k + 1]
    int* members = start + al4(k + 1);                           // [F] frames grouped by cluster, ascending inside a group
    float* pd = reinterpret_cast<float*>(members + al4(F));      // [2048] chunk minima: [chunk][frame of the pass]
    int* pj = reinterpret_cast<int*>(pd + 2048);                 // [2048]
    float* xs = reinterpret_cast<float*>(pj + 2048);             // [F][D] the frames, if x_in_lds
    __shared__ int changed;
    const int tid = threadIdx.x;
    // blockIdx.y = repeat: the same utterances clustered again from other random frames (EOT over the defense); repeat r
    // uses key seed + r * 0xC2B2AE3D27D4EB4F and writes slot r * gridDim.x + utterance of every output
    const float* x = feats + (size_t)blockIdx.x * F * D;
    const size_t slot = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    seed += (uint64_t)blockIdx.y * 0xC2B2AE3D27D4EB4Full;
    for (int i = tid; i < al4(F); i += 1024) ids[i] = -1;  // the pad entries stay -1: no cluster
    if (x_in_lds)
        for (int e = tid; e < F * D; e += 1024) xs[e] = x[e];
    const float* xu = x_in_lds ? xs : x;  // what the update reads
    if (seeded) {
        // random initialisation: rank the frames by (key, frame); `members` holds the keys, `cnt` the k chosen frames
        // (both are free until the first update)
        unsigned* keys = reinterpret_cast<unsigned*>(members);
        int* chosen = cnt;
        const int64_t utt = index_base + blockIdx.x;
        for (int i = tid; i < F; i += 1024)
            keys[i] = philox4x32_10_w0(seed, (uint32_t)i, 0u, (uint32_t)utt, (uint32_t)((uint64_t)utt >> 32));
        __syncthreads();
        for (int i = tid; i < F; i += 1024) {
            const unsigned ki = keys[i];
            int rank = 0;
            for (int g = 0; g < F; ++g) {
                const unsigned kg = keys[g];
                rank += (kg < ki) || (kg == ki && g < i);
            }
            if (rank < k) chosen[rank] = i;
        }
        __syncthreads();
    }
    for (int e = tid; e < npair * 2 * DPAD; e += 1024) {
        const int j = e / DPAD, d = e - j * DPAD;
        float v = 0.f;
        if (j < k && d < D) {
            const int f0 = seeded ? cnt[j] : (int)((long long)j * F / k);
            v = x[(size_t)f0 * D + d];
        }
        cs[CS_AT(j, d)] = v;
    }
    // thread = (frame slot li, centroid chunk jc), TWO frames per thread (li and li + slots): a centroid read (16-byte LDS
    // broadcast) serves both frames -- half the 6.5 MB of LDS reads per assignment step.  Phase trace (SG_FECO_TRACE) at 64 x
    // 300 x 32, k = 150: assignment 25.5 -> 23.5 us per step (lists 5.6, update 2.1) -- so the step is bound by its packed
    // sub / mul / add (16 us at the full packed rate: it runs at ~70 % of it), not by the LDS; the third of them that an
    // fma would save is the contract's separate roundings.  Same arithmetic per (frame, centroid); chunk minima are still
    // merged in ascending centroid order.  F <= 2048: one pass, the frames stay in registers.
    // (D > 32: one frame per thread -- two frames of 64 dimensions do not fit the 128 registers of a 1024-thread block)
    constexpr bool TWO = DPAD == 32;
    constexpr int PMAX = TWO ? 2048 : 1024;
    const int P = F <= PMAX ? F : PMAX;         // frames per pass
    const int slots = TWO ? (P + 1) / 2 : P;    // thread slots per chunk
    const int JC = max(1, min(8, 1024 / slots));  // JC * P <= 2048: capacity of pd / pj
    const int li = tid % slots, jc = tid / slots;
    const bool worker = jc < JC;
    const int plo = (int)((long long)npair * jc / JC), phi = worker ? (int)((long long)npair * (jc + 1) / JC) : 0;
    float2v xr0[DPAD / 2], xr1[TWO ? DPAD / 2 : 1];  // the two frames, dimensions (2q, 2q + 1) per register pair
    auto load_frames = [&](int f0) __attribute__((always_inline)) {
        const int i0 = f0 + li, i1 = f0 + li + slots, fend = min(F, f0 + P);
#pragma unroll
        for (int q = 0; q < DPAD / 2; ++q) {
            xr0[q].x = (worker && i0 < fend && 2 * q < D) ? x[(size_t)i0 * D + 2 * q] : 0.f;
            xr0[q].y = (worker && i0 < fend && 2 * q + 1 < D) ? x[(size_t)i0 * D + 2 * q + 1] : 0.f;
            if constexpr (TWO) {
                xr1[q].x = (worker && i1 < fend && 2 * q < D) ? x[(size_t)i1 * D + 2 * q] : 0.f;
                xr1[q].y = (worker && i1 < fend && 2 * q + 1 < D) ? x[(size_t)i1 * D + 2 * q + 1] : 0.f;
            }
        }
    };
    load_frames(0);
    __syncthreads();
    for (int it = 0; it < max_iter; ++it) {
        FECO_STAMP(4 * it)
        if (tid == 0) changed = 0;
        __syncthreads();
        for (int f0 = 0; f0 < F; f0 += P) {
            const int np = min(P, F - f0);  // frames of this pass
            if (F > P) load_frames(f0);     // long utterances (several passes)
            float best0 = INFINITY, best1 = INFINITY;
            int bj0 = 2 * plo, bj1 = 2 * plo;
            if (worker && li < np) {
                for (int pr = plo; pr < phi; ++pr) {
                    const float4* c4 = reinterpret_cast<const float4*>(cs + (size_t)pr * 2 * DPAD);
                    float2v acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};  // (distance to centroid 2 pr, to centroid 2 pr + 1) per frame
#pragma unroll
                    for (int q = 0; q < DPAD / 2; ++q) {  // d ascending; pad dims add +0 (x = c = 0), acc unchanged
                        const float4 c = c4[q];            // c[2q] of both centroids, c[2q + 1] of both
                        const float2v c0 = {c.x, c.y}, c1 = {c.z, c.w};
                        float2v df = pk_sub_xlo(xr0[q], c0);
                        acc0 = acc0 + df * df;
                        df = pk_sub_xhi(xr0[q], c1);
                        acc0 = acc0 + df * df;
                        if constexpr (TWO) {
                            df = pk_sub_xlo(xr1[q], c0);
                            acc1 = acc1 + df * df;
                            df = pk_sub_xhi(xr1[q], c1);
                            acc1 = acc1 + df * df;
                        }
                    }
                    const bool odd_ok = 2 * pr + 1 < k;
                    if (acc0.x < best0) { best0 = acc0.x; bj0 = 2 * pr; }
                    if (odd_ok && acc0.y < best0) { best0 = acc0.y; bj0 = 2 * pr + 1; }
                    if constexpr (TWO) {
                        if (acc1.x < best1) { best1 = acc1.x; bj1 = 2 * pr; }
                        if (odd_ok && acc1.y < best1) { best1 = acc1.y; bj1 = 2 * pr + 1; }
                    }
                }
            }
            if (worker && li < np) {
                pd[jc * np + li] = best0;
                pj[jc * np + li] = bj0;
                if (TWO && li + slots < np) {
                    pd[jc * np + li + slots] = best1;
                    pj[jc * np + li + slots] = bj1;
                }
            }
            __syncthreads();
            for (int r = tid; r < np; r += 1024) {  // merge the chunks in ascending centroid order: the lowest index wins ties
                float b = pd[r];
                int bb = pj[r];
                for (int c = 1; c < JC; ++c) {
                    const float v = pd[c * np + r];
                    if (v < b) {
                        b = v;
                        bb = pj[c * np + r];
                    }
                }
                if (ids[f0 + r] != bb) {
                    ids[f0 + r] = bb;
                    changed = 1;
                }
            }
            __syncthreads();
        }
        FECO_STAMP(4 * it + 1)
        if (!changed) break;
        // member lists.  Counts: one LDS atomic per frame.
        for (int j = tid; j < k; j += 1024) cnt[j] = 0;
        __syncthreads();
        for (int i = tid; i < F; i += 1024) atomicAdd(&cnt[ids[i]], 1);
        __syncthreads();
        // offsets: cluster j adds up the counts below it (16-byte broadcast reads)
        for (int j = tid; j <= k; j += 1024) {
            int run = 0;
            const int j4 = j & ~3;
            for (int q = 0; q < j4; q += 4) {
                const int4 c = *reinterpret_cast<const int4*>(cnt + q);
                run += c.x + c.y + c.z + c.w;
            }
            for (int q = j4; q < j; ++q) run += cnt[q];
            start[j] = run;
        }
        __syncthreads();
        // slots: frame i goes behind the earlier frames of its cluster -> ascending frame order inside a cluster
        for (int i = tid; i < F; i += 1024) {
            const int j = ids[i];
            int pos = 0;
            const int i4 = i & ~3;
            for (int q = 0; q < i4; q += 4) {
                const int4 v = *reinterpret_cast<const int4*>(ids + q);
                pos += (v.x == j) + (v.y == j) + (v.z == j) + (v.w == j);
            }
            for (int q = i4; q < i; ++q) pos += ids[q] == j;
            members[start[j] + pos] = i;
        }
        __syncthreads();
        FECO_STAMP(4 * it + 2)
        // update: thread (j, d) sums its cluster's frames in ascending frame order; an empty cluster keeps its centroid
        for (int e = tid; e < k * DPAD; e += 1024) {
            const int j = e / DPAD, d = e - j * DPAD;
            const int n = cnt[j];
            if (d < D && n > 0) {
                float sum = 0.f;
                const int o = start[j];
#pragma unroll 4
                for (int m = 0; m < n; ++m) sum = sum + xu[(size_t)members[o + m] * D + d];
                cs[CS_AT(j, d)] = sum / (float)n;
            }
        }
        __syncthreads();
        FECO_STAMP(4 * it + 3)
    }
    FECO_STAMP(4 * 16)
    // cnt / cs describe the final ids in both exits: "nothing changed" leaves the previous iteration's lists and means
    // valid, the max_iter exit has just rebuilt them
    if (out) {
        float* o = out + slot * k * D;
        for (int e = tid; e < k * D; e += 1024) {
            const int j = e / D, d = e - j * D;
            o[e] = cnt[j] > 0 ? cs[CS_AT(j, d)] : x[(size_t)j * D + d];  // feature_level.py:213-214 `force` fallback
        }
        for (int j = tid; j < k; j += 1024) counts[slot * k + j] = cnt[j];
    }
    for (int i = tid; i < F; i += 1024) assign[slot * F + i] = ids[i];
    FECO_STAMP(4 * 16 + 1)
}

// out[b][j][d] = mean over frames with id j (ascending order) or, for an empty cluster, feats[b][j][d]
__global__ void feco_compress_kernel(const float* __restrict__ feats, const int* __restrict__ assign, int F, int D, int k,
                                     float* __restrict__ out, int* __restrict__ counts) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= k * D) return;
    const int j = e / D, d = e - j * D;
    const float* x = feats + (size_t)b * F * D;
    const int* ids = assign + (size_t)b * F;
    float sum = 0.f;
    int cnt = 0;
    for (int i = 0; i < F; ++i)
        if (ids[i] == j) {
            sum = sum + x[(size_t)i * D + d];
            ++cnt;
        }
    out[((size_t)b * k + j) * D + d] = cnt > 0 ? sum / (float)cnt : x[(size_t)j * D + d];
    if (d == 0) counts[(size_t)b * k + j] = cnt;
}

// dfeats[b][i][d] = dout[b][id_i][d] / count[id_i]  (+ dout[b][i][d] if cluster i is empty and force)
__global__ void feco_compress_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ assign,
                                         const int* __restrict__ counts, int F, int D, int k, int force,
                                         float* __restrict__ dfeats) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= F * D) return;
    const int i = e / D, d = e - i * D;
    const int j = assign[(size_t)b * F + i];
    float g = dout[((size_t)b * k + j) * D + d] / (float)counts[(size_t)b * k + j];
    if (force && i < k && counts[(size_t)b * k + i] == 0) g = g + dout[((size_t)b * k + i) * D + d];
    dfeats[((size_t)b * F + i) * D + d] = g;
}

// The same for R repeats of the clustering of the SAME features (EOT over the defense): the compression is linear, so
// the repeats' gradients wrt the features are summed right here, in repeat order -- one log-mel / MFCC adjoint
// serves all of them.  dout (R, B, k, D), assign (R, B, F), counts (R, B, k) -> dfeats (B, F, D).
__global__ void feco_compress_bwd_reps_kernel(const float* __restrict__ dout, const int* __restrict__ assign,
                                              const int* __restrict__ counts, int B, int F, int D, int k, int force, int R,
                                              float* __restrict__ dfeats) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= F * D) return;
    const int i = e / D, d = e - i * D;
    float acc = 0.f;
    for (int r = 0; r < R; ++r) {
        const size_t u = (size_t)r * B + b;
        const int j = assign[u * F + i];
        float g = dout[(u * k + j) * D + d] / (float)counts[u * k + j];
        if (force && i < k && counts[u * k + i] == 0) g = g + dout[(u * k + i) * D + d];
        acc = r == 0 ? g : acc + g;
    }
    dfeats[((size_t)b * F + i) * D + d] = acc;
}

}  // namespace

static int feco_kmeans_impl(sg_ctx* ctx, const float* feats_dev, int32_t B, int32_t F, int32_t D, int32_t k, int32_t max_iter,
                            int seeded, uint64_t seed, int64_t index_base, int reps, int32_t* assign_dev, float* out_dev,
                            int32_t* counts_dev, void* stream) {
    if (!ctx) return SG_ERR_ARG;
    if (!feats_dev || !assign_dev || B <= 0 || F <= 0 || D <= 0 || D > kFecoMaxD || k <= 0 || k > F || max_iter <= 0 || reps < 1 ||
        reps > 65535)
        return feco_fail(ctx, SG_ERR_ARG, "sg_feco_kmeans: need 0 < k <= F, 0 < D <= %d, max_iter > 0", kFecoMaxD);
    if (hipSetDevice(ctx->device) != hipSuccess) return feco_fail(ctx, SG_ERR_HIP, "sg_feco_kmeans: hipSetDevice failed");
    // LDS holds the centroids (rows padded to 32 / 64 floats), ids, member lists and the chunk minima; the frames stay
    // in registers / HBM.  150 KB of it covers ~23 s at D <= 32, ratio 0.5 (k = 1150); beyond that the call is refused.
    constexpr size_t kLdsMax = 150 * 1024;
    const int dpad = D <= 32 ? 32 : 64;
    size_t lds = (size_t)((k + 1) / 2) * 2 * dpad * sizeof(float) +
                 ((size_t)2 * al4(F) + al4(k) + al4(k + 1)) * sizeof(int) + 2 * 2048 * sizeof(float);
    const size_t xbytes = (size_t)F * D * sizeof(float);
    const int x_in_lds = lds + xbytes <= kLdsMax;  // the frames too, so the update never leaves the CU
    if (x_in_lds) lds += xbytes;
    if (lds > kLdsMax)
        return feco_fail(ctx, SG_ERR_ARG, "sg_feco_kmeans: %d clusters x %d dims + %d frames need %zu bytes of LDS (limit %zu): "
                         "utterance too long for one block", k, D, F, lds, kLdsMax);
    const void* fn = dpad == 32 ? reinterpret_cast<const void*>(feco_kmeans_kernel<32>)
                                : reinterpret_cast<const void*>(feco_kmeans_kernel<64>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    if (e != hipSuccess) return feco_fail(ctx, SG_ERR_HIP, "sg_feco_kmeans: %s", hipGetErrorString(e));
    if (dpad == 32)
        hipLaunchKernelGGL(feco_kmeans_kernel<32>, dim3(B, reps), dim3(1024), lds, (hipStream_t)stream, feats_dev, F, D, k, max_iter,
                           seeded, seed, index_base, x_in_lds, assign_dev, out_dev, counts_dev);
    else
        hipLaunchKernelGGL(feco_kmeans_kernel<64>, dim3(B, reps), dim3(1024), lds, (hipStream_t)stream, feats_dev, F, D, k, max_iter,
                           seeded, seed, index_base, x_in_lds, assign_dev, out_dev, counts_dev);
    e = hipGetLastError();
    if (e != hipSuccess) return feco_fail(ctx, SG_ERR_HIP, "sg_feco_kmeans: %s", hipGetErrorString(e));
    static const bool tr_on = getenv("SG_FECO_TRACE") != nullptr;
    if (tr_on) {
        static bool armed = false;
        if (!armed) {
            const int one = 1;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_feco_trace_on), &one, sizeof(one));
            armed = true;
        } else if (hipStreamSynchronize((hipStream_t)stream) == hipSuccess) {
            unsigned long long h[4 * 16 + 4];
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_feco_trace), sizeof(h)) == hipSuccess) {
                fprintf(stderr, "feco k-means phases (us), block (0, 0): set-up %.2f;", 0.0);
                for (int it = 0; it < max_iter && it < 16 && h[4 * it + 1] > h[4 * it]; ++it)
                    fprintf(stderr, " it%d assign %.2f lists %.2f update %.2f;", it, (h[4 * it + 1] - h[4 * it]) * 0.01,
                            h[4 * it + 2] > h[4 * it + 1] ? (h[4 * it + 2] - h[4 * it + 1]) * 0.01 : 0.0,
                            h[4 * it + 3] > h[4 * it + 2] ? (h[4 * it + 3] - h[4 * it + 2]) * 0.01 : 0.0);
                fprintf(stderr, " out %.2f; loop + out total %.2f\n", (h[65] - h[64]) * 0.01, (h[65] - h[0]) * 0.01);
            }
        }
    }
    return SG_OK;
}

extern "C" int sg_feco_kmeans(sg_ctx* ctx, const float* feats_dev, int32_t B, int32_t F, int32_t D, int32_t k,
                              int32_t max_iter, int32_t* assign_dev, void* stream) {
    return feco_kmeans_impl(ctx, feats_dev, B, F, D, k, max_iter, 0, 0, 0, 1, assign_dev, nullptr, nullptr, stream);
}

extern "C" int sg_feco_kmeans_seeded(sg_ctx* ctx, const float* feats_dev, int32_t B, int32_t F, int32_t D, int32_t k,
                                     int32_t max_iter, uint64_t seed, int64_t index_base, int32_t* assign_dev, void* stream) {
    return feco_kmeans_impl(ctx, feats_dev, B, F, D, k, max_iter, 1, seed, index_base, 1, assign_dev, nullptr, nullptr, stream);
}

extern "C" int sg_feco_kmeans_compress(sg_ctx* ctx, const float* feats_dev, int32_t B, int32_t F, int32_t D, int32_t k,
                                       int32_t max_iter, int32_t random_init, uint64_t seed, int64_t index_base, int32_t reps,
                                       int32_t* assign_dev, float* out_dev, int32_t* counts_dev, void* stream) {
    if (!out_dev || !counts_dev) return feco_fail(ctx, SG_ERR_ARG, "sg_feco_kmeans_compress: out and counts are required");
    if (reps > 1 && !random_init)
        return feco_fail(ctx, SG_ERR_ARG, "sg_feco_kmeans_compress: repeats of the evenly started clustering coincide (reps must be 1)");
    return feco_kmeans_impl(ctx, feats_dev, B, F, D, k, max_iter, random_init != 0, seed, index_base, reps, assign_dev, out_dev,
                            counts_dev, stream);
}

extern "C" int sg_feco_compress(sg_ctx* ctx, const float* feats_dev, const int32_t* assign_dev, int32_t B, int32_t F,
                                int32_t D, int32_t k, float* out_dev, int32_t* counts_dev, void* stream) {
    if (!ctx) return SG_ERR_ARG;
    if (!feats_dev || !assign_dev || !out_dev || !counts_dev || B <= 0 || F <= 0 || D <= 0 || k <= 0 || k > F)
        return feco_fail(ctx, SG_ERR_ARG, "sg_feco_compress: bad arguments");
    if (hipSetDevice(ctx->device) != hipSuccess) return feco_fail(ctx, SG_ERR_HIP, "sg_feco_compress: hipSetDevice failed");
    hipLaunchKernelGGL(feco_compress_kernel, dim3((k * D + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, feats_dev,
                       assign_dev, F, D, k, out_dev, counts_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return feco_fail(ctx, SG_ERR_HIP, "sg_feco_compress: %s", hipGetErrorString(e));
    return SG_OK;
}

extern "C" int sg_feco_compress_backward(sg_ctx* ctx, const float* dout_dev, const int32_t* assign_dev,
                                         const int32_t* counts_dev, int32_t B, int32_t F, int32_t D, int32_t k,
                                         int32_t force, float* dfeats_dev, void* stream) {
    if (!ctx) return SG_ERR_ARG;
    if (!dout_dev || !assign_dev || !counts_dev || !dfeats_dev || B <= 0 || F <= 0 || D <= 0 || k <= 0 || k > F)
        return feco_fail(ctx, SG_ERR_ARG, "sg_feco_compress_backward: bad arguments");
    if (hipSetDevice(ctx->device) != hipSuccess)
        return feco_fail(ctx, SG_ERR_HIP, "sg_feco_compress_backward: hipSetDevice failed");
    hipLaunchKernelGGL(feco_compress_bwd_kernel, dim3((F * D + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, dout_dev,
                       assign_dev, counts_dev, F, D, k, force, dfeats_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return feco_fail(ctx, SG_ERR_HIP, "sg_feco_compress_backward: %s", hipGetErrorString(e));
    return SG_OK;
}

extern "C" int sg_feco_compress_backward_reps(sg_ctx* ctx, const float* dout_dev, const int32_t* assign_dev,
                                              const int32_t* counts_dev, int32_t B, int32_t F, int32_t D, int32_t k,
                                              int32_t force, int32_t reps, float* dfeats_dev, void* stream) {
    if (!ctx) return SG_ERR_ARG;
    if (!dout_dev || !assign_dev || !counts_dev || !dfeats_dev || B <= 0 || F <= 0 || D <= 0 || k <= 0 || k > F || reps < 1)
        return feco_fail(ctx, SG_ERR_ARG, "sg_feco_compress_backward_reps: bad arguments");
    if (hipSetDevice(ctx->device) != hipSuccess)
        return feco_fail(ctx, SG_ERR_HIP, "sg_feco_compress_backward_reps: hipSetDevice failed");
    hipLaunchKernelGGL(feco_compress_bwd_reps_kernel, dim3((F * D + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, dout_dev,
                       assign_dev, counts_dev, B, F, D, k, force, reps, dfeats_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return feco_fail(ctx, SG_ERR_HIP, "sg_feco_compress_backward_reps: %s", hipGetErrorString(e));
    return SG_OK;
}
