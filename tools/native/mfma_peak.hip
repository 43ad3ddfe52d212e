// Measures the sustained f32 MFMA rate of the whole chip with nothing else in the loop: the practical
// ceiling the conv GEMMs are compared with in DESIGN.md.  Operand data matters (switching power
// lowers the sustained clock), so both constant and pseudo-random operands are timed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void spin(float* out, int iters, int mode) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        s = s * 1664525u + 1013904223u; a[i] = mode ? ((int)(s >> 8) - (1 << 23)) * (1.f / (1 << 23)) : 1.f;
        s = s * 1664525u + 1013904223u; b[i] = mode ? ((int)(s >> 8) - (1 << 23)) * (1.f / (1 << 23)) : 2.f;
    }
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c3, 0, 0, 0);
        if (mode == 2) {  // new operands every step (cheap VALU in the MFMA shadow)
            a[0] = -a[1]; a[1] = -a[2]; a[2] = -a[3]; a[3] = -b[0]; b[0] = -b[1]; b[1] = -b[2]; b[2] = -b[3]; b[3] = a[0] * 0.999f;
        }
    }
    c0 += c1 + c2 + c3;
    if (c0[0] == 12345.f) out[0] = c0[1];
}
__global__ __launch_bounds__(256) void spin0(float* out, int iters, float a, float b) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    c0 += c1 + c2 + c3;
    if (c0[0] == 12345.f) out[0] = c0[1];
}
int main() {
    float* d; (void)hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int cfg[][3] = {{512, 20000, 1}, {1024, 20000, 1}, {2048, 20000, 1}, {512, 100000, 1}, {1024, 100000, 1},
                          {1024, 2000, 1}, {1024, 2000, 0}, {1024, 100000, 0}, {1024, 100000, 2}};
    for (int rep = 0; rep < 6; ++rep) {
        const int blocks = 1024, iters = rep < 3 ? 20000 : 100000;
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin0, dim3(blocks), dim3(256), 0, 0, d, iters, rep % 3 == 2 ? 0.7312f : 1.f, rep % 3 == 2 ? -1.3371f : 2.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 4 * (2.0 * 32 * 32 * 2);
        printf("spin0 blocks %4d iters %6d: %8.3f ms  %.1f TFLOP/s\n", blocks, iters, ms, flop / ms * 1e-9);
    }
    for (auto& c : cfg) {
        for (int rep = 0; rep < 2; ++rep) {
            const int blocks = c[0], iters = c[1], mode = c[2];
            hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, d, iters, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * 4 * iters * 4 * (2.0 * 32 * 32 * 2);
            printf("blocks %4d iters %6d operands %-16s: %8.3f ms  %.1f TFLOP/s\n", blocks, iters,
                   mode == 0 ? "constant" : mode == 1 ? "random, fixed" : "random, changing", ms, flop / ms * 1e-9);
        }
    }
    return 0;
}
