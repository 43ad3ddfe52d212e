// Latency of a DEPENDENT chain of f32 MFMAs (one accumulator, one wave per SIMD): what bounds the small-batch
// contractions.  Prints cycles per instruction for v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32, with 1, 2 and 4
// independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(64) void chain16(float* out, long long* cyc, int iters, float a, float b) {
    f32x4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[k], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += c[i][0];
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; out[0] = s; }
}
template <int NACC>
__global__ __launch_bounds__(64) void chain32(float* out, long long* cyc, int iters, float a, float b) {
    f32x16 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = f32x16{0};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[k], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += c[i][0];
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; out[0] = s; }
}
// four waves per block, one accumulator each: do they land on four different SIMDs (44 cycles per MFMA) or share (more)?
__global__ __launch_bounds__(256) void chain16_block(float* out, long long* cyc, int iters, float a, float b) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) { cyc[threadIdx.x >> 6] = t1 - t0; out[threadIdx.x >> 6] = c[0]; }
}
int main() {
    {
        float* d; long long* c; (void)hipMalloc(&d, 64); (void)hipMalloc(&c, 64);
        long long h[4];
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int blocks : {1, 136, 272}) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(chain16_block, dim3(blocks), dim3(256), 0, 0, d, c, 20000, 0.5f, 0.25f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  (kernel %.3f ms by HIP events: with the cycle count below the shader clock of this short launch is ~%.2f GHz)\n", ms,
                   20000.0 * (blocks > 256 ? 64.0 : 44.0) / (ms * 1e6));
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
            printf("4 waves per block, %3d blocks: %.1f %.1f %.1f %.1f cycles per dependent 16x16x4 MFMA (waves 0..3 of the last block to write)\n",
                   blocks, h[0] / 20000.0, h[1] / 20000.0, h[2] / 20000.0, h[3] / 20000.0);
        }
    }
    float* d; long long* c; (void)hipMalloc(&d, 4); (void)hipMalloc(&c, 8);
    const int iters = 20000;
    long long h;
#define RUN(K, N, NAME) \
    hipLaunchKernelGGL((K<N>), dim3(1), dim3(64), 0, 0, d, c, iters, 0.5f, 0.25f); (void)hipDeviceSynchronize(); \
    hipLaunchKernelGGL((K<N>), dim3(1), dim3(64), 0, 0, d, c, iters, 0.5f, 0.25f); (void)hipDeviceSynchronize(); \
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); \
    printf("%-22s %d accumulator(s): %.1f clock-counter ticks per MFMA\n", NAME, N, (double)h / ((double)iters * N));
    RUN(chain16, 1, "v_mfma_f32_16x16x4_f32") RUN(chain16, 2, "v_mfma_f32_16x16x4_f32") RUN(chain16, 4, "v_mfma_f32_16x16x4_f32")
    RUN(chain32, 1, "v_mfma_f32_32x32x2_f32") RUN(chain32, 2, "v_mfma_f32_32x32x2_f32")
    // s_memtime / readcyclecounter runs at a fixed 100 MHz on this part: convert with the 2.4 GHz shader clock
    printf("(the counter ticks at 100 MHz: multiply by 24 for 2.4 GHz shader cycles)\n");
    return 0;
}
