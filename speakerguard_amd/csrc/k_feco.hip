// FeCo feature-level defense: per-utterance k-means over the frames of a feature
// matrix, then every cluster replaced by the mean of its frames (reference defense/feature_level.py:168-217),
// forward and backward.
//
// The reference calls a third-party k-means with a RANDOM initialisation (libKMCUDA / kmeans_pytorch), so its
// cluster ids are not reproducible even between two runs of the reference.  The clustering here follows its own
// DETERMINISM CONTRACT (restated in oracle/feco.py, checked bit for bit):
//   * k = int(F * ratio) centroids, centroid j initialised to frame floor(j * F / k);
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

// one block per utterance; dynamic LDS: [x[F][D] when X_IN_LDS,] c[k][D], ids[F] (int); changed flag.
// X_IN_LDS = false is the long-utterance form: the frames stay in HBM (read-only, served by L1/L2), only the
// centroids and ids live in LDS -- same arithmetic, same order, same ids.
template <bool X_IN_LDS>
__global__ __launch_bounds__(256) void feco_kmeans_kernel(const float* __restrict__ feats, int F, int D, int k,
                                                          int max_iter, int* __restrict__ assign) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* cs = lds + (X_IN_LDS ? (size_t)F * D : 0);
    int* ids = reinterpret_cast<int*>(cs + (size_t)k * D);
    __shared__ int changed;
    const float* x = feats + (size_t)blockIdx.x * F * D;
    const float* xs = x;
    if (X_IN_LDS) {
        for (int i = threadIdx.x; i < F * D; i += blockDim.x) lds[i] = x[i];
        xs = lds;
    }
    for (int i = threadIdx.x; i < F; i += blockDim.x) ids[i] = -1;
    __syncthreads();
    for (int i = threadIdx.x; i < k * D; i += blockDim.x) {
        const int j = i / D, d = i - j * D;
        cs[i] = xs[(size_t)(int)((long long)j * F / k) * D + d];
    }
    __syncthreads();
    for (int it = 0; it < max_iter; ++it) {
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < F; i += blockDim.x) {
            float best = INFINITY;
            int bj = 0;
            for (int j = 0; j < k; ++j) {
                float acc = 0.f;
                for (int d = 0; d < D; ++d) {
                    const float df = xs[(size_t)i * D + d] - cs[(size_t)j * D + d];
                    acc = acc + df * df;
                }
                if (acc < best) {
                    best = acc;
                    bj = j;
                }
            }
            if (ids[i] != bj) {
                ids[i] = bj;
                changed = 1;
            }
        }
        __syncthreads();
        if (!changed) break;
        // update: thread (j, d) sums its cluster's frames in ascending frame order
        for (int e = threadIdx.x; e < k * D; e += blockDim.x) {
            const int j = e / D, d = e - j * D;
            float sum = 0.f;
            int cnt = 0;
            for (int i = 0; i < F; ++i)
                if (ids[i] == j) {
                    sum = sum + xs[(size_t)i * D + d];
                    ++cnt;
                }
            if (cnt > 0) cs[e] = sum / (float)cnt;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < F; i += blockDim.x) assign[(size_t)blockIdx.x * F + i] = ids[i];
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

extern "C" int sg_feco_kmeans(sg_ctx* ctx, const float* feats_dev, int32_t B, int32_t F, int32_t D, int32_t k,
                              int32_t max_iter, int32_t* assign_dev, void* stream) {
    if (!ctx) return SG_ERR_ARG;
    if (!feats_dev || !assign_dev || B <= 0 || F <= 0 || D <= 0 || D > kFecoMaxD || k <= 0 || k > F || max_iter <= 0)
        return feco_fail(ctx, SG_ERR_ARG, "sg_feco_kmeans: need 0 < k <= F, 0 < D <= %d, max_iter > 0", kFecoMaxD);
    if (hipSetDevice(ctx->device) != hipSuccess) return feco_fail(ctx, SG_ERR_HIP, "sg_feco_kmeans: hipSetDevice failed");
    // Placement: frames + centroids + ids in one block's LDS when they fit (3 s .. ~8 s utterances); for longer
    // utterances the frames stay in HBM and only centroids + ids use LDS (up to k * D * 4 + F * 4 <= 150 KB, i.e.
    // ~23 s at D = 32, ratio 0.5); beyond that the call is refused.
    constexpr size_t kLdsMax = 150 * 1024;
    const size_t lds_small = ((size_t)k * D) * sizeof(float) + (size_t)F * sizeof(int);
    const size_t lds_full = lds_small + (size_t)F * D * sizeof(float);
    if (lds_small > kLdsMax)
        return feco_fail(ctx, SG_ERR_ARG, "sg_feco_kmeans: %d clusters x %d dims + %d ids need %zu bytes of LDS (limit %zu): "
                         "utterance too long for one block", k, D, F, lds_small, kLdsMax);
    const bool in_lds = lds_full <= kLdsMax;
    const void* fn = in_lds ? reinterpret_cast<const void*>(feco_kmeans_kernel<true>)
                            : reinterpret_cast<const void*>(feco_kmeans_kernel<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    if (e != hipSuccess) return feco_fail(ctx, SG_ERR_HIP, "sg_feco_kmeans: %s", hipGetErrorString(e));
    if (in_lds)
        hipLaunchKernelGGL(feco_kmeans_kernel<true>, dim3(B), dim3(256), lds_full, (hipStream_t)stream, feats_dev, F, D, k,
                           max_iter, assign_dev);
    else
        hipLaunchKernelGGL(feco_kmeans_kernel<false>, dim3(B), dim3(256), lds_small, (hipStream_t)stream, feats_dev, F, D, k,
                           max_iter, assign_dev);
    e = hipGetLastError();
    if (e != hipSuccess) return feco_fail(ctx, SG_ERR_HIP, "sg_feco_kmeans: %s", hipGetErrorString(e));
    return SG_OK;
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
