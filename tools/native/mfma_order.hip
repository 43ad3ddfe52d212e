// Do v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 accumulate the k products in the same order and with the same
// roundings?  If a chain of K/4 16x16x4 instructions gives the bits of a chain of K/2 32x32x2 instructions, the
// latency-bound small-batch contractions (one dependent chain of K/2 x 64 cycles per output tile, 48 us for tdnn3) could
// use the 16x16x4 form (K/4 x 32 cycles: a quarter of the chain latency) without giving up the bit-equality of results
// across batch sizes that shard invariance rests on.  Also compared: a sequential fmaf chain and a mul-then-add chain.
//   hipcc --offload-arch=gfx950 -O2 tools/native/mfma_order.hip -o /tmp/mfma_order && /tmp/mfma_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void run(const float* A, const float* B, int K, float* C32, float* C16) {
    const int l = threadIdx.x;
    // 32x32x2: lane l feeds A[row l%32][k0 + l/32] and B[k0 + l/32][col l%32]
    f32x16 c = {0};
    for (int k0 = 0; k0 < K; k0 += 2)
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * K + k0 + (l >> 5)], B[(k0 + (l >> 5)) * 32 + (l & 31)], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C32[((r >> 2) * 8 + (l >> 5) * 4 + (r & 3)) * 32 + (l & 31)] = c[r];
    // 16x16x4 on each 16x16 sub-block: lane l feeds A[row l%16][k0 + l/16], B[k0 + l/16][col l%16]
    for (int bi = 0; bi < 2; ++bi)
        for (int bj = 0; bj < 2; ++bj) {
            f32x4 d = {0};
            for (int k0 = 0; k0 < K; k0 += 4)
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(bi * 16 + (l & 15)) * K + k0 + (l >> 4)],
                                                         B[(k0 + (l >> 4)) * 32 + bj * 16 + (l & 15)], d, 0, 0, 0);
            for (int r = 0; r < 4; ++r) C16[(bi * 16 + 4 * (l >> 4) + r) * 32 + bj * 16 + (l & 15)] = d[r];
        }
}

int main() {
    int total_diff_mfma = 0, total_diff_fma = 0, total_diff_muladd = 0, total = 0;
    for (int trial = 0; trial < 20; ++trial) {
        const int K = trial < 10 ? 64 : 3584;
        std::vector<float> A(32 * K), B(K * 32), C32(1024), C16(1024);
        unsigned s = 12345u + trial * 977u;
        auto rnd = [&]() {
            s = s * 1664525u + 1013904223u;
            const float m = ((int)(s >> 8) - (1 << 23)) * (1.f / (1 << 23));
            s = s * 1664525u + 1013904223u;
            return m * ldexpf(1.f, (int)(s >> 28) - 8);  // magnitudes spread over 16 binades: roundings matter
        };
        for (auto& v : A) v = rnd();
        for (auto& v : B) v = rnd();
        float *dA, *dB, *d32, *d16;
        (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&d32, 4096); (void)hipMalloc(&d16, 4096);
        (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(run, dim3(1), dim3(64), 0, 0, dA, dB, K, d32, d16);
        (void)hipMemcpy(C32.data(), d32, 4096, hipMemcpyDeviceToHost);
        (void)hipMemcpy(C16.data(), d16, 4096, hipMemcpyDeviceToHost);
        int dm = 0, df = 0, da = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                float f = 0.f, m = 0.f;
                for (int k = 0; k < K; ++k) {
                    f = fmaf(A[i * K + k], B[k * 32 + j], f);
                    volatile float p = A[i * K + k] * B[k * 32 + j];
                    m = m + p;
                }
                dm += memcmp(&C32[i * 32 + j], &C16[i * 32 + j], 4) != 0;
                df += memcmp(&C32[i * 32 + j], &f, 4) != 0;
                da += memcmp(&C32[i * 32 + j], &m, 4) != 0;
            }
        printf("K %4d trial %2d: 32x32x2 vs 16x16x4 differ in %4d / 1024 outputs; 32x32x2 vs sequential fmaf %4d; vs mul-then-add %4d\n",
               K, trial, dm, df, da);
        total_diff_mfma += dm; total_diff_fma += df; total_diff_muladd += da; total += 1024;
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(d32); (void)hipFree(d16);
    }
    printf("TOTAL: 32x32x2 vs 16x16x4 %d / %d differ; vs sequential fmaf chain %d; vs mul-then-add chain %d\n", total_diff_mfma, total,
           total_diff_fma, total_diff_muladd);
    return 0;
}
