// Post-attack data formats: int16 PCM quantisation of the adversarial
// audio with the perturbation metrics the evaluation reads, and the equal-error-rate threshold scan.
// HBM-bound byte / integer work: one pass for the range decisions, one fused pass for everything else.
#include <cmath>
#include <cstdarg>
#include <cstdio>

#include "sg_internal.h"

using namespace sg;

namespace {

int post_fail(sg_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

template <typename T, typename Op>
__device__ __forceinline__ T block_reduce(T v, T* scratch, Op op) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o, 64));
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    T r = scratch[0];
    for (int i = 1; i < nw; ++i) r = op(r, scratch[i]);
    return r;
}

// One block per utterance.
//   PCM (reference attackMain.py:154-166 save_audio): if 0.9*max <= 1 and 0.9*min >= -1 (per utterance)
//   the audio is multiplied by 2^15 in float32, then numpy's astype(int16): truncate toward zero, keep the
//   low 16 bits (1.0 -> 32768 -> -32768; verified against numpy in tests/golden/make_golden.py).
//   Metrics (reference metric/metric.py:8-42): preprocess() divides a signal by 2^15 unless
//   -1 <= max <= 1 (only the max is tested), then L2 / L0 / L1 / Linf of adver - benign and
//   SNR = 10 log10(sum benign^2 / sum noise^2), +inf when the noise power is 0.  Sums in fp64.
__global__ __launch_bounds__(1024) void wav_finalize_kernel(const float* __restrict__ benign,
                                                            const float* __restrict__ adver, int T,
                                                            int16_t* __restrict__ pcm, double* __restrict__ metrics) {
    __shared__ double sd[16];
    __shared__ float sf[16];
    __shared__ long long sl[16];
    const int b = blockIdx.x;
    const float* a = adver + (size_t)b * T;
    const float* g = benign ? benign + (size_t)b * T : nullptr;
    float amax = -INFINITY, amin = INFINITY, gmax = -INFINITY;
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        const float v = a[i];
        amax = fmaxf(amax, v);
        amin = fminf(amin, v);
        if (g) gmax = fmaxf(gmax, g[i]);
    }
    auto fmx = [](float x, float y) { return fmaxf(x, y); };
    auto fmn = [](float x, float y) { return fminf(x, y); };
    amax = block_reduce(amax, sf, fmx);
    amin = block_reduce(amin, sf, fmn);
    gmax = block_reduce(gmax, sf, fmx);
    const bool scale_pcm = 0.9f * amax <= 1.f && 0.9f * amin >= -1.f;
    const bool a_raw = -1.f <= amax && amax <= 1.f;  // metric.preprocess keeps the signal as is
    const bool g_raw = -1.f <= gmax && gmax <= 1.f;
    double s2 = 0.0, s1 = 0.0, sb = 0.0;
    float linf = 0.f;
    long long l0 = 0;
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        const float v = a[i];
        if (pcm) {
            const float q = scale_pcm ? v * 32768.f : v;
            pcm[(size_t)b * T + i] = (int16_t)(int)q;  // (int) truncates toward zero; the narrowing keeps 16 bits
        }
        if (g) {
            const float av = a_raw ? v : v / 32768.f;
            const float gv = g_raw ? g[i] : g[i] / 32768.f;
            const float d = av - gv;
            s2 += (double)d * (double)d;
            s1 += fabs((double)d);
            sb += (double)gv * (double)gv;
            linf = fmaxf(linf, fabsf(d));
            l0 += d != 0.f;
        }
    }
    if (!g || !metrics) return;
    auto dadd = [](double x, double y) { return x + y; };
    auto ladd = [](long long x, long long y) { return x + y; };
    s2 = block_reduce(s2, sd, dadd);
    s1 = block_reduce(s1, sd, dadd);
    sb = block_reduce(sb, sd, dadd);
    linf = block_reduce(linf, sf, fmx);
    l0 = block_reduce(l0, sl, ladd);
    if (threadIdx.x == 0) {
        double* m = metrics + (size_t)b * 5;
        m[0] = sqrt(s2);
        m[1] = (double)l0;
        m[2] = s1;
        m[3] = (double)linf;
        m[4] = s2 <= 0.0 ? INFINITY : 10.0 * log10(sb / s2);
    }
}

// set_threshold.py:22-47: for every candidate c in score_target (in order),
//   frr = #(target < c) * 100 / n_target, far = #(untarget >= c) * 100 / n_untarget, keep the FIRST
//   candidate with the smallest |frr - far|.  One thread per candidate, then a first-wins argmin.
__global__ void eer_count_kernel(const float* __restrict__ tgt, int nt, const float* __restrict__ unt, int nu,
                                 double* __restrict__ diff, double* __restrict__ frr, double* __restrict__ far) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nt) return;
    const float c = tgt[i];
    int lt = 0, ge = 0;
    for (int k = 0; k < nt; ++k) lt += tgt[k] < c;
    for (int k = 0; k < nu; ++k) ge += unt[k] >= c;
    const double r = (double)((long long)lt * 100) / (double)nt;
    const double a = (double)((long long)ge * 100) / (double)nu;
    frr[i] = r;
    far[i] = a;
    diff[i] = fabs(r - a);
}

__global__ __launch_bounds__(1024) void eer_pick_kernel(const float* __restrict__ tgt, int nt,
                                                        const double* __restrict__ diff, const double* __restrict__ frr,
                                                        const double* __restrict__ far, double* __restrict__ out3) {
    __shared__ double sd[16];
    __shared__ int si[16];
    double best = INFINITY;
    int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < nt; i += blockDim.x) {
        const double d = diff[i];
        if (d < best) {  // strictly smaller: the earliest index of a thread's stride wins
            best = d;
            idx = i;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ob < best || (ob == best && oi < idx)) {
            best = ob;
            idx = oi;
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) {
        sd[wid] = best;
        si[wid] = idx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (sd[w] < best || (sd[w] == best && si[w] < idx)) {
                best = sd[w];
                idx = si[w];
            }
        // the reference starts from min_difference = inf and threshold 0.0 (:31-35): nothing beats inf
        // only if every difference is NaN/inf, which cannot happen for n > 0
        out3[0] = idx < nt ? (double)tgt[idx] : 0.0;
        out3[1] = idx < nt ? frr[idx] : 0.0;
        out3[2] = idx < nt ? far[idx] : 0.0;
    }
}

}  // namespace

extern "C" int sg_wav_finalize(sg_ctx* ctx, const float* benign_dev, const float* adver_dev, int32_t B, int32_t T,
                               int16_t* pcm_dev, double* metrics_dev, void* stream) {
    if (!ctx) return SG_ERR_ARG;
    if (!adver_dev || B <= 0 || T <= 0 || (!pcm_dev && !metrics_dev) || (metrics_dev && !benign_dev))
        return post_fail(ctx, SG_ERR_ARG, "sg_wav_finalize: need adver, B > 0, T > 0, an output, and benign for metrics");
    if (hipSetDevice(ctx->device) != hipSuccess) return post_fail(ctx, SG_ERR_HIP, "sg_wav_finalize: hipSetDevice failed");
    hipLaunchKernelGGL(wav_finalize_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, metrics_dev ? benign_dev : nullptr,
                       adver_dev, T, pcm_dev, metrics_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return post_fail(ctx, SG_ERR_HIP, "sg_wav_finalize: %s", hipGetErrorString(e));
    return SG_OK;
}

extern "C" int sg_eer_threshold(sg_ctx* ctx, const float* target_dev, int32_t n_target, const float* untarget_dev,
                                int32_t n_untarget, double* out3_dev, void* stream) {
    if (!ctx) return SG_ERR_ARG;
    if (!target_dev || !untarget_dev || !out3_dev || n_target <= 0 || n_untarget <= 0)
        return post_fail(ctx, SG_ERR_ARG, "sg_eer_threshold: need non-empty target and untarget score lists");
    if (hipSetDevice(ctx->device) != hipSuccess) return post_fail(ctx, SG_ERR_HIP, "sg_eer_threshold: hipSetDevice failed");
    hipStream_t s = (hipStream_t)stream;
    double* tmp = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&tmp), (size_t)3 * n_target * sizeof(double));
    if (e != hipSuccess) return post_fail(ctx, SG_ERR_HIP, "sg_eer_threshold: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(eer_count_kernel, dim3((n_target + 255) / 256), dim3(256), 0, s, target_dev, n_target, untarget_dev,
                       n_untarget, tmp, tmp + n_target, tmp + 2 * (size_t)n_target);
    hipLaunchKernelGGL(eer_pick_kernel, dim3(1), dim3(1024), 0, s, target_dev, n_target, tmp, tmp + n_target,
                       tmp + 2 * (size_t)n_target, out3_dev);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    if (e != hipSuccess) return post_fail(ctx, SG_ERR_HIP, "sg_eer_threshold: %s", hipGetErrorString(e));
    return SG_OK;
}
