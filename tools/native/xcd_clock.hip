// Per-XCD speed under a whole-chip f32 MFMA load: every CU runs one block of 8 waves (two per SIMD, four accumulators
// each), every wave stamps the 100 MHz wall clock (s_memrealtime) and the shader clock (s_memtime) around its loop.
// Prints, per XCD, the median wall time of its waves and the shader-clock frequency that implies.
// Two operand sets: launches 0-2 multiply the same two constants for ever (the quietest the matrix pipes can be), launches
// 3-5 multiply per-lane pseudo-random operands that change with every instruction (what a real contraction feeds them).
//   hipcc --offload-arch=gfx950 -O3 tools/native/xcd_clock.hip -o tools/native/xcd_clock
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float rnd(unsigned& st) {
    st = st * 1664525u + 1013904223u;
    return (float)(int)(st >> 8) * (1.0f / 8388608.0f) - 1.0f;  // [-1, 1)
}
__global__ __launch_bounds__(512) void load_random(unsigned long long* rec, float* out, int iters) {
    f32x16 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = f32x16{0};
    unsigned st = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = rnd(st);
        b[i] = rnd(st);
    }
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i += 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + (k >> 2)) & 7], c[k & 3], 0, 0, 0);
    }
    float s = c[0][0] + c[1][0] + c[2][0] + c[3][0];
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), t1 = __builtin_readcyclecounter();
    if (s == 12345.f) out[0] = s;
    if ((threadIdx.x & 63) == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const size_t w = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
        rec[w * 3 + 0] = r1 - r0;
        rec[w * 3 + 1] = t1 - t0;
        rec[w * 3 + 2] = xcc & 0xF;
    }
}
__global__ __launch_bounds__(512) void load(unsigned long long* rec, float* out, int iters, float a, float b) {
    __shared__ float pad[84 * 256];
    f32x16 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = f32x16{0};
    if (iters < 0) pad[threadIdx.x] = a;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[k], 0, 0, 0);
    }
    float s = c[0][0] + c[1][0] + c[2][0] + c[3][0];
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), t1 = __builtin_readcyclecounter();
    if (s == 12345.f) out[0] = s + pad[0];
    if ((threadIdx.x & 63) == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const size_t w = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
        rec[w * 3 + 0] = r1 - r0;
        rec[w * 3 + 1] = t1 - t0;
        rec[w * 3 + 2] = xcc & 0xF;
    }
}
int main() {
    const int blocks = 256, waves = blocks * 8, iters = 100000;
    unsigned long long* rec;
    float* out;
    (void)hipMalloc(&rec, (size_t)waves * 3 * 8);
    (void)hipMalloc(&out, 64);
    for (int rep = 0; rep < 6; ++rep) {
        if (rep < 3) hipLaunchKernelGGL(load, dim3(blocks), dim3(512), 0, 0, rec, out, iters, 0.5f, 0.25f);
        else hipLaunchKernelGGL(load_random, dim3(blocks), dim3(512), 0, 0, rec, out, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)waves * 3);
        (void)hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost);
        printf("launch %d (%d MFMAs per wave, 2 waves per SIMD, %s operands):\n", rep, iters * 4, rep < 3 ? "constant" : "per-lane random");
        for (int x = 0; x < 8; ++x) {
            std::vector<double> us, cyc;
            for (int w = 0; w < waves; ++w)
                if ((int)h[(size_t)w * 3 + 2] == x) {
                    us.push_back(h[(size_t)w * 3] * 0.01);
                    cyc.push_back((double)h[(size_t)w * 3 + 1]);
                }
            if (us.empty()) continue;
            std::sort(us.begin(), us.end());
            std::sort(cyc.begin(), cyc.end());
            const double mu = us[us.size() / 2], mc = cyc[cyc.size() / 2];
            printf("  XCD %d: %4zu waves  median %9.1f us (min %9.1f max %9.1f)  s_memtime ticks %12.0f = %7.1f MHz;  %5.1f wall-ns per MFMA pair-slot -> %6.1f TFLOP/s if all XCDs ran like this\n",
                   x, us.size(), mu, us.front(), us.back(), mc, mc / mu, mu * 1e3 / (iters * 4.0 * 2),
                   (double)iters * 4 * 4096 * waves / (mu * 1e-6) / 1e12);
        }
    }
    return 0;
}
