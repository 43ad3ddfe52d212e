// Philox4x32-10 counter-based generator (Salmon, Moraes, Dror, Shaw, SC'11; Random123 philox4x32, 10 rounds).
// Restated on the CPU in oracle/philox.py, which is checked against the Random123 known-answer vectors.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace sg {

__device__ __forceinline__ uint32_t philox4x32_10_w0(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

}  // namespace sg
