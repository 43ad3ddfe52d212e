// Small memory-bound kernels around the TDNN contractions: input range check, CMVN, statistics
// pooling (+backward), the fused PGD update.
#include "sg_internal.h"

namespace sg {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------- check_input_range (model/utils.py:7-19)
// range_type='origin' (xv_plda.py:48): if 0.9*max <= 1 and 0.9*min >= -1 the batch is in the
// [-1,1] float domain and is multiplied by 2^15, otherwise it is left alone.  Single block; the
// result stays on the device so no host sync is needed.  Two stages: 256 blocks -> partial max/min,
// one block -> the decision (a single-block version cost 0.86 ms at 64 x 3 s, 11 % of a PGD step).
constexpr int kScaleBlocks = 256;

// (round 5: 16-byte loads, four of them in flight per thread, 1024 threads per block -- the one-float-per-iteration loop of
// 256-thread blocks ran at 0.8 TB/s: 121 us for the 98 MB of 512 utterances, three times per attack.  max / min do not
// depend on the order.)
constexpr int kScaleThreads = 1024;
__global__ __launch_bounds__(kScaleThreads) void input_range_partial_kernel(const float* __restrict__ x, int64_t n,
                                                                            float* __restrict__ part) {
    __shared__ float smax[kScaleThreads / 64], smin[kScaleThreads / 64];
    float mx = -INFINITY, mn = INFINITY;
    const int64_t n4 = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? n / 4 : 0;  // whole float4s (an unaligned row: one by one)
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const int64_t stride = (int64_t)kScaleBlocks * kScaleThreads;
    int64_t i = (int64_t)blockIdx.x * kScaleThreads + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
        mx = fmaxf(fmaxf(fmaxf(mx, fmaxf(a.x, a.y)), fmaxf(fmaxf(a.z, a.w), fmaxf(b.x, b.y))),
                   fmaxf(fmaxf(fmaxf(b.z, b.w), fmaxf(c.x, c.y)), fmaxf(fmaxf(c.z, c.w), fmaxf(fmaxf(d.x, d.y), fmaxf(d.z, d.w)))));
        mn = fminf(fminf(fminf(mn, fminf(a.x, a.y)), fminf(fminf(a.z, a.w), fminf(b.x, b.y))),
                   fminf(fminf(fminf(b.z, b.w), fminf(c.x, c.y)), fminf(fminf(c.z, c.w), fminf(fminf(d.x, d.y), fminf(d.z, d.w)))));
    }
    for (; i < n4; i += stride) {
        const float4 a = x4[i];
        mx = fmaxf(mx, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
        mn = fminf(mn, fminf(fminf(a.x, a.y), fminf(a.z, a.w)));
    }
    for (int64_t j = 4 * n4 + (int64_t)blockIdx.x * kScaleThreads + threadIdx.x; j < n; j += stride) {
        const float v = x[j];
        mx = fmaxf(mx, v);
        mn = fminf(mn, v);
    }
    mx = wave_max_f(mx);
    mn = wave_min_f(mn);
    if ((threadIdx.x & 63) == 0) {
        smax[threadIdx.x >> 6] = mx;
        smin[threadIdx.x >> 6] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kScaleThreads / 64; ++w) {
            mx = fmaxf(mx, smax[w]);
            mn = fminf(mn, smin[w]);
        }
        part[blockIdx.x] = mx;
        part[kScaleBlocks + blockIdx.x] = mn;
    }
}

__global__ __launch_bounds__(256) void input_range_final_kernel(const float* __restrict__ part, float* scale, int mode) {
    __shared__ float smax[4], smin[4];
    float mx = wave_max_f(part[threadIdx.x]);
    float mn = wave_min_f(part[kScaleBlocks + threadIdx.x]);
    if ((threadIdx.x & 63) == 0) {
        smax[threadIdx.x >> 6] = mx;
        smin[threadIdx.x >> 6] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
        mn = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        const bool unit = 0.9f * mx <= 1.f && 0.9f * mn >= -1.f;  // ori_type == 'scale'
        *scale = mode == 0 ? (unit ? 32768.f : 1.f) : (unit ? 1.f : 1.f / 32768.f);
    }
}

hipError_t launch_input_scale(const float* x, int64_t n, float* scratch, float* scale, int mode, hipStream_t s) {
    hipLaunchKernelGGL(input_range_partial_kernel, dim3(kScaleBlocks), dim3(kScaleThreads), 0, s, x, n, scratch);
    hipLaunchKernelGGL(input_range_final_kernel, dim3(1), dim3(256), 0, s, scratch, scale, mode);
    return hipGetLastError();
}

// ---------------------------------------------------------------- PGD update (attack/FGSM.py:65,68)
__global__ __launch_bounds__(256) void pgd_update_kernel(float* __restrict__ x, const float* __restrict__ g,
                                                         const float* __restrict__ lo, const float* __restrict__ hi,
                                                         int64_t n, float step, int grad_sign) {
    const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 3 < n) {
        float4 xv = *reinterpret_cast<float4*>(x + i4);
        const float4 gv = *reinterpret_cast<const float4*>(g + i4);
        const float4 lv = *reinterpret_cast<const float4*>(lo + i4);
        const float4 hv = *reinterpret_cast<const float4*>(hi + i4);
        const float s = step * (float)grad_sign;
#define SG_UPD(c) xv.c = fminf(fmaxf(xv.c + s * (gv.c > 0.f ? 1.f : (gv.c < 0.f ? -1.f : 0.f)), lv.c), hv.c)
        SG_UPD(x); SG_UPD(y); SG_UPD(z); SG_UPD(w);
#undef SG_UPD
        *reinterpret_cast<float4*>(x + i4) = xv;
    } else {
        for (int64_t i = i4; i < n; ++i) {
            const float gv = g[i];
            const float sg = gv > 0.f ? 1.f : (gv < 0.f ? -1.f : 0.f);
            x[i] = fminf(fmaxf(x[i] + step * (float)grad_sign * sg, lo[i]), hi[i]);
        }
    }
}

hipError_t launch_pgd_update(float* x, const float* g, const float* lo, const float* hi, int64_t n, float step,
                             int grad_sign, hipStream_t s) {
    const int64_t blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(pgd_update_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, g, lo, hi, n, step, grad_sign);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out,
                                                        int ld_out, int64_t rows, int ncol) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ld_out) return;
    const int64_t r = i / ld_out;
    const int c = (int)(i - r * ld_out);
    out[i] = c < ncol ? in[r * ld_in + c] : 0.f;
}

hipError_t launch_copy_cols(const float* in, int ld_in, float* out, int ld_out, int64_t rows, int ncol, hipStream_t s) {
    const int64_t n = rows * ld_out;
    hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, ld_in, out, ld_out, rows,
                       ncol);
    return hipGetLastError();
}

// out[r][0..ncol) = sum_z in[z][r][0..ncol)  (split-K slabs, fixed order)
__global__ __launch_bounds__(256) void sum_cols_kernel(const float* __restrict__ in, int ld_in, int nsplit,
                                                       long long slab_stride, float* __restrict__ out, int ld_out,
                                                       int64_t rows, int ncol) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ncol) return;
    const int64_t r = i / ncol;
    const int c = (int)(i - r * ncol);
    float v = 0.f;
    for (int z = 0; z < nsplit; ++z) v += in[(size_t)z * slab_stride + r * ld_in + c];
    out[r * ld_out + c] = v;
}

hipError_t launch_sum_cols(const float* in, int ld_in, int nsplit, long long slab_stride, float* out, int ld_out,
                           int64_t rows, int ncol, hipStream_t s) {
    const int64_t n = rows * ncol;
    hipLaunchKernelGGL(sum_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, ld_in, nsplit, slab_stride,
                       out, ld_out, rows, ncol);
    return hipGetLastError();
}

// ---------------------------------------------------------------- CMVN (model/iv_plda.py:296-377)
// Centred sliding window of 300 frames, mean only.  window(t) = [start, end) as at :321-336.
__device__ __forceinline__ void cmvn_window(int t, int F, int& start, int& end) {
    start = t - kCmnWindow / 2;
    end = start + kCmnWindow;
    if (start < 0) {
        end -= start;
        start = 0;
    }
    if (end > F) {
        start -= end - F;
        end = F;
        if (start < 0) start = 0;
    }
}

// grid (B), block 1024.  out row stride ld_out >= 30; columns 30..ld_out-1 are zeroed (GEMM K pad).
// One block per utterance is all the parallelism there is (64 blocks at B = 64), so the kernels are latency
// bound: 1024 threads and 32 row-strided partial sums per column keep 4x more loads in flight than the first
// version (256 threads, 8 partials: 23 us per launch for 2.3 MB).
constexpr int kCmvnThreads = 1024;
constexpr int kCmvnParts = kCmvnThreads / 32;
__global__ __launch_bounds__(kCmvnThreads) void cmvn_fwd_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out,
                                                       int ld_out, int F) {
    const int b = blockIdx.x;
    const float* x = in + (size_t)b * F * ld_in;
    float* y = out + (size_t)b * F * ld_out;
    __shared__ double total[kCep];
    __shared__ double part[kCmvnParts][32];
    if (F <= kCmnWindow) {
        // every window is the whole utterance: one column sum per cepstrum, 32 row-strided partial sums per column
        // combined in a fixed order.  Thread (column d, row class r) keeps its <= 10 values (rows r, r + 32, ...) in
        // registers -- loaded in one batch of independent, clamped loads -- and writes the same elements back: one global
        // round trip and two barriers, where the first version read the utterance twice (9.1 -> see profiles/).
        constexpr int kRows = (kCmnWindow + kCmvnParts - 1) / kCmvnParts;
        const int d = threadIdx.x & 31, r = threadIdx.x >> 5;
        float v[kRows];
#pragma unroll
        for (int j = 0; j < kRows; ++j) {
            const int t = min(r + j * kCmvnParts, F - 1);
            v[j] = d < kCep ? x[(size_t)t * ld_in + d] : 0.f;
        }
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < kRows; ++j)
            if (r + j * kCmvnParts < F) acc += (double)v[j];
        part[r][d] = acc;
        __syncthreads();
        if (threadIdx.x < kCep) {
            double a2 = 0.0;
            for (int rr = 0; rr < kCmvnParts; ++rr) a2 += part[rr][threadIdx.x];
            total[threadIdx.x] = a2;
        }
        __syncthreads();
        const float mean = d < kCep ? (float)(total[d] / (double)F) : 0.f;
        if (d < ld_out) {
#pragma unroll
            for (int j = 0; j < kRows; ++j) {
                const int t = r + j * kCmvnParts;
                if (t < F) y[(size_t)t * ld_out + d] = d < kCep ? v[j] - mean : 0.f;
            }
        }
        for (int i = threadIdx.x; i < F * (ld_out - 32); i += kCmvnThreads) {  // (row stride above 32: the rest of the zero padding)
            const int t = i / (ld_out - 32), dd = 32 + i - t * (ld_out - 32);
            y[(size_t)t * ld_out + dd] = 0.f;
        }
    } else {
        for (int i = threadIdx.x; i < F * ld_out; i += kCmvnThreads) {
            const int t = i / ld_out, d = i - t * ld_out;
            float v = 0.f;
            if (d < kCep) {
                int s, e;
                cmvn_window(t, F, s, e);
                double acc = 0.0;
                for (int u = s; u < e; ++u) acc += (double)x[(size_t)u * ld_in + d];
                v = x[(size_t)t * ld_in + d] - (float)(acc / (double)(e - s));
            }
            y[i] = v;
        }
    }
}

// d_in[u] = d_out[u] - sum_{t : u in window(t)} d_out[t] / |window(t)|
// d_out arrives as `nsplit` split-K slabs of the tdnn1 data-gradient contraction (summed in order).
// NS > 0: the slab count as a compile-time constant (the attack loop's kL1BwdSplitK): the slab loads of a row are then one
// batch of independent loads instead of two dependent ones.
template <int NS>
__global__ __launch_bounds__(kCmvnThreads) void cmvn_bwd_kernel(const float* __restrict__ dout, int ld_dout, int nsplit_rt,
                                                       long long slab_stride, float* __restrict__ din, int ld_din,
                                                       int F) {
    const int b = blockIdx.x;
    const float* g = dout + (size_t)b * F * ld_dout;
    float* y = din + (size_t)b * F * ld_din;
    const int nsplit = NS > 0 ? NS : nsplit_rt;
    auto gsum = [&](int t, int d) __attribute__((always_inline)) {
        const float* src = g + (size_t)t * ld_dout + d;
        if (NS > 0) {
            float part[NS > 0 ? NS : 1];
#pragma unroll
            for (int z = 0; z < NS; ++z) part[z] = src[(size_t)z * slab_stride];
            float v = 0.f;
#pragma unroll
            for (int z = 0; z < NS; ++z) v += part[z];
            return v;
        }
        float v = 0.f;
#pragma unroll 5
        for (int z = 0; z < nsplit; ++z) v += src[(size_t)z * slab_stride];
        return v;
    };
    __shared__ double total[kCep];
    __shared__ double part[kCmvnParts][32];
    if (F <= kCmnWindow) {
        // thread (column d, row class r): the slab sums of its <= 10 rows stay in registers (kRows x nsplit independent
        // loads, slabs added in order), the column means come from the same row-strided partial sums as before
        constexpr int kRows = (kCmnWindow + kCmvnParts - 1) / kCmvnParts;
        const int d = threadIdx.x & 31, r = threadIdx.x >> 5;
        float v[kRows];
#pragma unroll
        for (int j = 0; j < kRows; ++j) {
            const int t = min(r + j * kCmvnParts, F - 1);
            v[j] = d < kCep ? gsum(t, d) : 0.f;
        }
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < kRows; ++j)
            if (r + j * kCmvnParts < F) acc += (double)v[j];
        part[r][d] = acc;
        __syncthreads();
        if (threadIdx.x < kCep) {
            double a2 = 0.0;
            for (int rr = 0; rr < kCmvnParts; ++rr) a2 += part[rr][threadIdx.x];
            total[threadIdx.x] = a2 / (double)F;
        }
        __syncthreads();
        if (d < kCep) {
            const float mean = (float)total[d];
#pragma unroll
            for (int j = 0; j < kRows; ++j) {
                const int t = r + j * kCmvnParts;
                if (t < F) y[(size_t)t * ld_din + d] = v[j] - mean;
            }
        }
    } else {
        for (int i = threadIdx.x; i < F * kCep; i += kCmvnThreads) {
            const int u = i / kCep, d = i - u * kCep;
            double acc = 0.0;
            for (int t = 0; t < F; ++t) {
                int s, e;
                cmvn_window(t, F, s, e);
                if (u >= s && u < e) acc += (double)gsum(t, d) / (double)(e - s);
            }
            y[(size_t)u * ld_din + d] = gsum(u, d) - (float)acc;
        }
    }
}

hipError_t launch_cmvn_fwd(const float* in, int ld_in, float* out, int ld_out, int B, int F, hipStream_t s) {
    hipLaunchKernelGGL(cmvn_fwd_kernel, dim3(B), dim3(kCmvnThreads), 0, s, in, ld_in, out, ld_out, F);
    return hipGetLastError();
}
hipError_t launch_cmvn_bwd(const float* dout, int ld_dout, int nsplit, long long slab_stride, float* din, int ld_din,
                           int B, int F, hipStream_t s) {
    if (nsplit == kL1BwdSplitK)
        hipLaunchKernelGGL(cmvn_bwd_kernel<kL1BwdSplitK>, dim3(B), dim3(kCmvnThreads), 0, s, dout, ld_dout, nsplit, slab_stride, din, ld_din, F);
    else
        hipLaunchKernelGGL(cmvn_bwd_kernel<0>, dim3(B), dim3(kCmvnThreads), 0, s, dout, ld_dout, nsplit, slab_stride, din, ld_din, F);
    return hipGetLastError();
}

// ---------------------------------------------------------------- statistics pooling (xvecTDNN.py:62)
// stats = cat(mean_t, std_t (unbiased)) over the tdnn5 relu output (BatchNorm folded into fc1).
// grid (kPoolC/64, B), block 256: lane = channel, the 4 waves split the frames.
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ act, int Tc, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int b = blockIdx.y;
    const float* a = act + (size_t)b * Tc * kPoolC + c;
    __shared__ float red[4][64];
    // two-pass variance (like torch.std); the <= 96 frames a wave owns stay in registers between the
    // passes so the activation tensor is read once (longer utterances re-read from L2)
    constexpr int kKeep = 96;
    float keep[kKeep];
    const bool fits = Tc <= 4 * kKeep;
    float s = 0.f;
    if (fits) {
#pragma unroll
        for (int i = 0; i < kKeep; ++i) {
            const int t = wid + 4 * i;
            keep[i] = t < Tc ? a[(size_t)t * kPoolC] : 0.f;
            s += keep[i];
        }
    } else {
        for (int t = wid; t < Tc; t += 4) s += a[(size_t)t * kPoolC];
    }
    red[wid][lane] = s;
    __syncthreads();
    const float mean = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / (float)Tc;
    __syncthreads();
    float q = 0.f;
    if (fits) {
#pragma unroll
        for (int i = 0; i < kKeep; ++i) {
            const float d = keep[i] - mean;
            q += (wid + 4 * i < Tc) ? d * d : 0.f;
        }
    } else {
        for (int t = wid; t < Tc; t += 4) {
            const float d = a[(size_t)t * kPoolC] - mean;
            q += d * d;
        }
    }
    red[wid][lane] = q;
    __syncthreads();
    if (wid == 0) {
        const float var = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / (float)(Tc - 1);
        stats[(size_t)b * kStats + c] = mean;
        stats[(size_t)b * kStats + kPoolC + c] = sqrtf(var);
    }
}

// d act[t][c] = relu'(act) * ( dmean[c]/T + dstd[c] * (act - mean) / ((T-1) * std) ), std == 0 -> no std term
// (torch std_backward masks result == 0).  dstats arrives as `nsplit` split-K partials.
// NS > 0: the slab count at compile time (kFc1BwdSplitK): all 2 NS loads of the slab sums in flight at once.
template <int NS>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ act, const float* __restrict__ stats,
                                                       const float* __restrict__ dpart, int nsplit, int B, int Tc,
                                                       float* __restrict__ dact) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int b = blockIdx.y;
    float dmean = 0.f, dstd = 0.f;
    if (NS > 0) {
        float pm[NS > 0 ? NS : 1], ps[NS > 0 ? NS : 1];
#pragma unroll
        for (int z = 0; z < NS; ++z) {
            pm[z] = dpart[((size_t)z * B + b) * kStats + c];
            ps[z] = dpart[((size_t)z * B + b) * kStats + kPoolC + c];
        }
#pragma unroll
        for (int z = 0; z < NS; ++z) {
            dmean += pm[z];
            dstd += ps[z];
        }
    } else {
        for (int z = 0; z < nsplit; ++z) {
            dmean += dpart[((size_t)z * B + b) * kStats + c];
            dstd += dpart[((size_t)z * B + b) * kStats + kPoolC + c];
        }
    }
    const float mean = stats[(size_t)b * kStats + c];
    const float sd = stats[(size_t)b * kStats + kPoolC + c];
    const float beta = sd > 0.f ? dstd / ((float)(Tc - 1) * sd) : 0.f;
    const float alpha = dmean / (float)Tc;
    const float* a = act + (size_t)b * Tc * kPoolC + c;
    float* d = dact + (size_t)b * Tc * kPoolC + c;
    // elementwise from here on: the frames are dealt over the blocks of grid.z (small batches: 24 x 8 blocks of a 68-step
    // loop left most CUs idle, 15 us at 8 utterances) -- any split gives the same bits
    const int per = (Tc + (int)gridDim.z - 1) / (int)gridDim.z;
    const int t0 = (int)blockIdx.z * per, t1 = min(Tc, t0 + per);
    for (int t = t0 + wid; t < t1; t += 4) {
        const float v = a[(size_t)t * kPoolC];
        d[(size_t)t * kPoolC] = v > 0.f ? alpha + beta * (v - mean) : 0.f;
    }
}

hipError_t launch_pool_fwd(const float* act5, int B, int Tc, float* stats, hipStream_t s) {
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(kPoolC / 64, B), dim3(256), 0, s, act5, Tc, stats);
    return hipGetLastError();
}
hipError_t launch_pool_bwd(const float* act5, const float* stats, const float* dstats_part, int nsplit, int B, int Tc,
                           float* dact5, hipStream_t s) {
    const int z = B >= 64 ? 1 : (1536 + (kPoolC / 64) * B - 1) / ((kPoolC / 64) * B);  // >= ~1500 blocks in all
    if (nsplit == kFc1BwdSplitK)
        hipLaunchKernelGGL(pool_bwd_kernel<kFc1BwdSplitK>, dim3(kPoolC / 64, B, z), dim3(256), 0, s, act5, stats, dstats_part, nsplit, B,
                           Tc, dact5);
    else
        hipLaunchKernelGGL(pool_bwd_kernel<0>, dim3(kPoolC / 64, B, z), dim3(256), 0, s, act5, stats, dstats_part, nsplit, B, Tc, dact5);
    return hipGetLastError();
}

// ---------------------------------------------------------------- per-step records of a step with R EOT repeats
// rows r * B + b -> loss_out[b] = mean over r (repeat order), dec_out[b] = the most frequent decision, the first seen
// winning a tie (attack/FGSM.py:50-58 divides the summed per-batch means; attack/utils.py:118-125 Counter.most_common)
__global__ void eot_trace_reduce_kernel(const float* __restrict__ loss_rows, const int64_t* __restrict__ dec_rows, int R, int B,
                                        float* __restrict__ loss_out, int64_t* __restrict__ dec_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (loss_out) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += loss_rows[(size_t)r * B + b];
        loss_out[b] = s / (float)R;
    }
    if (dec_out) {
        int64_t best = dec_rows[b];
        int best_n = 0;
        for (int r = 0; r < R; ++r) {
            const int64_t d = dec_rows[(size_t)r * B + b];
            bool seen = false;
            for (int q = 0; q < r; ++q) seen |= dec_rows[(size_t)q * B + b] == d;
            if (seen) continue;
            int n = 0;
            for (int q = r; q < R; ++q) n += dec_rows[(size_t)q * B + b] == d;
            if (n > best_n) { best_n = n; best = d; }
        }
        dec_out[b] = best;
    }
}

hipError_t launch_eot_trace_reduce(const float* loss_rows, const int64_t* dec_rows, int R, int B, float* loss_out,
                                   int64_t* dec_out, hipStream_t s) {
    hipLaunchKernelGGL(eot_trace_reduce_kernel, dim3((B + 127) / 128), dim3(128), 0, s, loss_rows, dec_rows, R, B, loss_out, dec_out);
    return hipGetLastError();
}

}  // namespace sg
