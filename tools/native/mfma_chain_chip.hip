// Whole-chip version of mfma_chain.hip: every SIMD runs ONE wave with a dependent chain of v_mfma_f32_32x32x2_f32
// (NACC independent accumulators), the situation of the 32-row / 64-row stream-K tiles at 8 / 16 utterances per GPU.
// Prints TFLOP/s and the implied cycles per MFMA per SIMD at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 tools/native/mfma_chain_chip.hip -o tools/native/mfma_chain_chip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int LDS_KB>
__global__ __launch_bounds__(256) void chain32(float* out, int iters, float a, float b) {
    __shared__ float pad[LDS_KB * 256];
    f32x16 c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = f32x16{0};
    if (iters < 0) pad[threadIdx.x] = a;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[k], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += c[i][0];
    if (s == 12345.f) out[0] = s + pad[0];
}
template <int NACC>
__global__ __launch_bounds__(256) void chain16(float* out, int iters, float a, float b) {
    __shared__ float pad[84 * 256];
    f32x4 c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (iters < 0) pad[threadIdx.x] = a;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[k], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += c[i][0];
    if (s == 12345.f) out[0] = s + pad[0];
}
template <typename K>
static void run(const char* name, K kern, int blocks, int nacc, double flop_per_mfma, float* d) {
    const int iters = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f, 0.25f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f, 0.25f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_wave = (double)iters * nacc;
    const double tf = mfma_per_wave * 4 * blocks * flop_per_mfma / (ms * 1e-3) / 1e12;
    printf("%-34s blocks %4d acc %d: %8.3f ms  %7.1f TFLOP/s  %6.1f cycles per MFMA and wave at 2.4 GHz\n", name, blocks, nacc, ms, tf,
           ms * 1e-3 * 2.4e9 / mfma_per_wave);
}
int main() {
    float* d;
    (void)hipMalloc(&d, 64);
    for (int blocks : {1, 64, 256, 512}) {
        run("32x32x2 (1 block/CU: 84 KB LDS)", chain32<1, 84>, blocks, 1, 4096, d);
        run("32x32x2 (1 block/CU: 84 KB LDS)", chain32<2, 84>, blocks, 2, 4096, d);
        run("32x32x2 (1 block/CU: 84 KB LDS)", chain32<4, 84>, blocks, 4, 4096, d);
    }
    for (int blocks : {256, 512, 1024}) {
        run("32x32x2 (small LDS)", chain32<1, 1>, blocks, 1, 4096, d);
        run("32x32x2 (small LDS)", chain32<2, 1>, blocks, 2, 4096, d);
    }
    for (int blocks : {256}) {
        run("16x16x4 (84 KB LDS)", chain16<1>, blocks, 1, 2048, d);
        run("16x16x4 (84 KB LDS)", chain16<2>, blocks, 2, 2048, d);
        run("16x16x4 (84 KB LDS)", chain16<4>, blocks, 4, 2048, d);
        run("16x16x4 (84 KB LDS)", chain16<8>, blocks, 8, 2048, d);
    }
    return 0;
}
