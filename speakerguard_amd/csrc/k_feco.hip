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

constexpr int kFecoMaxD = 64;

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
// Same arithmetic, same order, same ids as before (oracle/feco.py restates the contract; tests compare bit for bit).
// Dynamic LDS: cs[k][DPAD], ids[F], cnt[k], start[k + 1], members[F], pd[1024], pj[1024].
template <int DPAD>
__global__ __launch_bounds__(1024) void feco_kmeans_kernel(const float* __restrict__ feats, int F, int D, int k,
                                                           int max_iter, int seeded, uint64_t seed, int64_t index_base,
                                                           int* __restrict__ assign) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* cs = lds;                                             // [k][DPAD], pad columns zero
    int* ids = reinterpret_cast<int*>(cs + (size_t)k * DPAD);   // [F]
    int* cnt = ids + F;                                          // [k]
    int* start = cnt + k;                                        // [k + 1]
    int* members = start + k + 1;                                // [F] frames grouped by cluster, ascending inside a group
    float* pd = reinterpret_cast<float*>(members + F);           // [1024] chunk minima
    int* pj = reinterpret_cast<int*>(pd + 1024);                 // [1024]
    __shared__ int changed;
    const int tid = threadIdx.x;
    const float* x = feats + (size_t)blockIdx.x * F * D;
    for (int i = tid; i < F; i += 1024) ids[i] = -1;
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
    for (int e = tid; e < k * DPAD; e += 1024) {
        const int j = e / DPAD, d = e - j * DPAD;
        const int f0 = seeded ? cnt[j] : (int)((long long)j * F / k);
        cs[e] = d < D ? x[(size_t)f0 * D + d] : 0.f;
    }
    // thread = (frame slot li, centroid chunk jc); F <= 1024: one pass, the frame stays in registers
    const int fpp = F <= 1024 ? F : 1024;
    const int JC = F <= 1024 ? max(1, min(8, 1024 / F)) : 1;
    const int li = tid % fpp, jc = tid / fpp;
    const bool worker = jc < JC;
    const int jlo = (int)((long long)k * jc / JC), jhi = worker ? (int)((long long)k * (jc + 1) / JC) : 0;
    float xr[DPAD];
#pragma unroll
    for (int d = 0; d < DPAD; ++d) xr[d] = (worker && li < F && d < D) ? x[(size_t)li * D + d] : 0.f;
    __syncthreads();
    for (int it = 0; it < max_iter; ++it) {
        if (tid == 0) changed = 0;
        __syncthreads();
        for (int f0 = 0; f0 < F; f0 += fpp) {
            const int i = f0 + li;
            if (F > fpp) {  // long utterances (several passes): load the frame of this pass
#pragma unroll
                for (int d = 0; d < DPAD; ++d) xr[d] = (worker && i < F && d < D) ? x[(size_t)i * D + d] : 0.f;
            }
            float best = INFINITY;
            int bj = jlo;
            if (worker && i < F) {
                for (int j = jlo; j < jhi; ++j) {
                    const float4* c4 = reinterpret_cast<const float4*>(cs + (size_t)j * DPAD);
                    float acc = 0.f;
#pragma unroll
                    for (int q = 0; q < DPAD / 4; ++q) {  // d ascending; pad dims add +0 (x = c = 0), acc unchanged
                        const float4 c = c4[q];
                        float df = xr[4 * q] - c.x;
                        acc = acc + df * df;
                        df = xr[4 * q + 1] - c.y;
                        acc = acc + df * df;
                        df = xr[4 * q + 2] - c.z;
                        acc = acc + df * df;
                        df = xr[4 * q + 3] - c.w;
                        acc = acc + df * df;
                    }
                    if (acc < best) {
                        best = acc;
                        bj = j;
                    }
                }
            }
            pd[tid] = best;
            pj[tid] = bj;
            __syncthreads();
            if (jc == 0 && i < F) {  // merge the chunks in ascending centroid order: the lowest index wins ties
                float b = pd[li];
                int bb = pj[li];
                for (int c = 1; c < JC; ++c) {
                    const float v = pd[c * fpp + li];
                    if (v < b) {
                        b = v;
                        bb = pj[c * fpp + li];
                    }
                }
                if (ids[i] != bb) {
                    ids[i] = bb;
                    changed = 1;
                }
            }
            __syncthreads();
        }
        if (!changed) break;
        // member lists: thread j counts its frames, thread 0 scans, thread j writes them in ascending frame order
        for (int j = tid; j < k; j += 1024) {
            int n = 0;
            for (int i = 0; i < F; ++i) n += ids[i] == j;
            cnt[j] = n;
        }
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int j = 0; j < k; ++j) {
                start[j] = run;
                run += cnt[j];
            }
            start[k] = run;
        }
        __syncthreads();
        for (int j = tid; j < k; j += 1024) {
            int o = start[j];
            for (int i = 0; i < F; ++i)
                if (ids[i] == j) members[o++] = i;
        }
        __syncthreads();
        // update: thread (j, d) sums its cluster's frames in ascending frame order; an empty cluster keeps its centroid
        for (int e = tid; e < k * DPAD; e += 1024) {
            const int j = e / DPAD, d = e - j * DPAD;
            const int n = cnt[j];
            if (d < D && n > 0) {
                float sum = 0.f;
                const int o = start[j];
                for (int m = 0; m < n; ++m) sum = sum + x[(size_t)members[o + m] * D + d];
                cs[e] = sum / (float)n;
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < F; i += 1024) assign[(size_t)blockIdx.x * F + i] = ids[i];
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

}  // namespace

static int feco_kmeans_impl(sg_ctx* ctx, const float* feats_dev, int32_t B, int32_t F, int32_t D, int32_t k, int32_t max_iter,
                            int seeded, uint64_t seed, int64_t index_base, int32_t* assign_dev, void* stream) {
    if (!ctx) return SG_ERR_ARG;
    if (!feats_dev || !assign_dev || B <= 0 || F <= 0 || D <= 0 || D > kFecoMaxD || k <= 0 || k > F || max_iter <= 0)
        return feco_fail(ctx, SG_ERR_ARG, "sg_feco_kmeans: need 0 < k <= F, 0 < D <= %d, max_iter > 0", kFecoMaxD);
    if (hipSetDevice(ctx->device) != hipSuccess) return feco_fail(ctx, SG_ERR_HIP, "sg_feco_kmeans: hipSetDevice failed");
    // LDS holds the centroids (rows padded to 32 / 64 floats), ids, member lists and the chunk minima; the frames stay
    // in registers / HBM.  150 KB of it covers ~23 s at D <= 32, ratio 0.5 (k = 1150); beyond that the call is refused.
    constexpr size_t kLdsMax = 150 * 1024;
    const int dpad = D <= 32 ? 32 : 64;
    const size_t lds = (size_t)k * dpad * sizeof(float) + ((size_t)2 * F + 2 * (size_t)k + 1) * sizeof(int) + 2 * 1024 * sizeof(float);
    if (lds > kLdsMax)
        return feco_fail(ctx, SG_ERR_ARG, "sg_feco_kmeans: %d clusters x %d dims + %d frames need %zu bytes of LDS (limit %zu): "
                         "utterance too long for one block", k, D, F, lds, kLdsMax);
    const void* fn = dpad == 32 ? reinterpret_cast<const void*>(feco_kmeans_kernel<32>)
                                : reinterpret_cast<const void*>(feco_kmeans_kernel<64>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    if (e != hipSuccess) return feco_fail(ctx, SG_ERR_HIP, "sg_feco_kmeans: %s", hipGetErrorString(e));
    if (dpad == 32)
        hipLaunchKernelGGL(feco_kmeans_kernel<32>, dim3(B), dim3(1024), lds, (hipStream_t)stream, feats_dev, F, D, k, max_iter,
                           seeded, seed, index_base, assign_dev);
    else
        hipLaunchKernelGGL(feco_kmeans_kernel<64>, dim3(B), dim3(1024), lds, (hipStream_t)stream, feats_dev, F, D, k, max_iter,
                           seeded, seed, index_base, assign_dev);
    e = hipGetLastError();
    if (e != hipSuccess) return feco_fail(ctx, SG_ERR_HIP, "sg_feco_kmeans: %s", hipGetErrorString(e));
    return SG_OK;
}

extern "C" int sg_feco_kmeans(sg_ctx* ctx, const float* feats_dev, int32_t B, int32_t F, int32_t D, int32_t k,
                              int32_t max_iter, int32_t* assign_dev, void* stream) {
    return feco_kmeans_impl(ctx, feats_dev, B, F, D, k, max_iter, 0, 0, 0, assign_dev, stream);
}

extern "C" int sg_feco_kmeans_seeded(sg_ctx* ctx, const float* feats_dev, int32_t B, int32_t F, int32_t D, int32_t k,
                                     int32_t max_iter, uint64_t seed, int64_t index_base, int32_t* assign_dev, void* stream) {
    return feco_kmeans_impl(ctx, feats_dev, B, F, D, k, max_iter, 1, seed, index_base, assign_dev, stream);
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
