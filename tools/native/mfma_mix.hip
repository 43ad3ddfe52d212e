// Which companions slow an f32 MFMA stream on gfx950?  Each kernel issues 4 MFMAs per loop step on 4
// accumulators (4 waves per SIMD) together with one kind of companion work; the printed rate is the MFMA
// rate.  Used to choose the operand-feeding scheme of the conv GEMM ("MFMA feed experiments").
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

template <int MODE, int NV>
__global__ __launch_bounds__(256) void mix(float* out, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i % 13) * 0.25f - 1.f;
    __syncthreads();
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float a0 = 0.5f + lane, a1 = 0.25f - lane, b0 = 1.5f * lane, b1 = 3.f - lane;
    float x0 = lane, x1 = 1.f, x2 = 2.f, x3 = 3.f;
    const float* p = lds + wid * 64 + lane;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 1) {  // independent VALU work
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(x1));
        }
        if (MODE == 2) {  // LDS reads feeding the NEXT step's operands (register double buffer)
            float na0, na1, nb0, nb1;
            const int o = (i & 15) * 256;
            na0 = p[o]; na1 = p[o + 2048]; nb0 = p[o + 4096]; nb1 = p[o + 6144];
            MFMA(a0, b0, c0); MFMA(a1, b0, c1); MFMA(a0, b1, c2); MFMA(a1, b1, c3);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            continue;
        }
        if (MODE == 3) {  // LDS reads, results not used by MFMAs
            const int o = (i & 15) * 256;
            x0 += p[o]; x1 += p[o + 2048]; x2 += p[o + 4096]; x3 += p[o + 6144];
        }
        if (MODE == 4) {  // W8 shape: two accumulators, 3 reads per 2 MFMAs, two k-steps per loop step
            float na0, na1, nb0;
            const int o = (i & 7) * 512;
            na0 = p[o]; na1 = p[o + 2048]; nb0 = p[o + 4096];
            MFMA(a0, b0, c0); MFMA(a1, b0, c1);
            a0 = p[o + 256]; a1 = p[o + 2304]; b0 = p[o + 4352];
            MFMA(na0, nb0, c0); MFMA(na1, nb0, c1);
            continue;
        }
        MFMA(a0, b0, c0); MFMA(a1, b0, c1); MFMA(a0, b1, c2); MFMA(a1, b1, c3);
    }
    c0 += c1 + c2 + c3;
    if (c0[0] == 12345.f) out[0] = c0[1] + x0 + x1 + x2 + x3;
}

// b128-fed variants: the lane's operands of 4 consecutive k-steps come from one ds_read_b128.
// TILE 0: 64x64 wave tile (4 reads / 16 MFMA); 1: 64x32 (3 reads / 8 MFMA).  GLDS: also stage with
// 4 global_load_lds_dwordx4 per wave and one barrier per 32 MFMAs, like one K-chunk of the GEMM.
template <int TILE, int GLDS, int BAR>
__global__ __launch_bounds__(1024) void mixq(float* out, const float* __restrict__ g, int iters) {
    __shared__ float4 lds[2][2048];  // 2 stages x 32 KB
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) (&lds[0][0])[i] = make_float4(0.25f, -0.5f, 0.125f, 1.f);
    __syncthreads();
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const int lane = threadIdx.x & 63, wid = (threadIdx.x >> 6) & 7;
    const float* gp = g + (size_t)((blockIdx.x * 8 + wid) & 4095) * 4096 + lane * 4;
    for (int i = 0; i < iters; ++i) {
        const float4* p = &lds[i & 1][0] + wid * 64 + lane;
        if (GLDS > 0) {
#pragma unroll
            for (int q = 0; q < GLDS; ++q) {
                // asm, not the builtin: after the builtin hipcc waits vmcnt(0) before the next ds_read
                const unsigned dst = __builtin_amdgcn_readfirstlane(
                    (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&lds[(i + 1) & 1][((wid * 4 + q) & 31) * 64]));
                const float* src = gp + (q & 3) * 256 + (i & 3) * 1024;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            }
        }
        float4 st[8];
        if (GLDS < 0) {
#pragma unroll
            for (int q = 0; q < -GLDS; ++q) st[q] = *reinterpret_cast<const float4*>(gp + (q & 3) * 256 + (i & 3) * 1024);
        }
        float4 a0 = p[0], a1 = p[512], b0 = p[1024], b1 = p[1536];
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {  // 4 k-groups of 8 = one 32-deep chunk
            float4 na0, na1, nb0, nb1;
            if (kg < 3) { na0 = p[(kg + 1) * 8]; na1 = p[512 + (kg + 1) * 8]; nb0 = p[1024 + (kg + 1) * 8]; if (TILE == 0) nb1 = p[1536 + (kg + 1) * 8]; }
            MFMA(a0.x, b0.x, c0); MFMA(a1.x, b0.x, c1); if (TILE == 0) { MFMA(a0.x, b1.x, c2); MFMA(a1.x, b1.x, c3); }
            MFMA(a0.y, b0.y, c0); MFMA(a1.y, b0.y, c1); if (TILE == 0) { MFMA(a0.y, b1.y, c2); MFMA(a1.y, b1.y, c3); }
            MFMA(a0.z, b0.z, c0); MFMA(a1.z, b0.z, c1); if (TILE == 0) { MFMA(a0.z, b1.z, c2); MFMA(a1.z, b1.z, c3); }
            MFMA(a0.w, b0.w, c0); MFMA(a1.w, b0.w, c1); if (TILE == 0) { MFMA(a0.w, b1.w, c2); MFMA(a1.w, b1.w, c3); }
            if (kg < 3) { a0 = na0; a1 = na1; b0 = nb0; if (TILE == 0) b1 = nb1; }
            if (GLDS < 0 && kg == 2) {
#pragma unroll
                for (int q = 0; q < -GLDS; ++q) lds[(i + 1) & 1][((wid * 4 + q) & 31) * 64 + lane] = st[q];
            }
        }
        if (GLDS > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (BAR) __syncthreads();
    }
    c0 += c1 + c2 + c3;
    if (c0[0] == 12345.f) out[0] = c0[1];
}

template <int TILE, int GLDS, int BAR>
void runq(const char* name, float* d, const float* g, int blocks, int threads) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    const int iters = 4000;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mixq<TILE, GLDS, BAR>), dim3(blocks), dim3(threads), 0, 0, d, g, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flop = (double)blocks * (threads / 64) * iters * (TILE == 0 ? 64 : 32) * (2.0 * 32 * 32 * 2);
    printf("%-58s blocks %4d x %4d thr (waves/SIMD %d): %7.3f ms  %6.1f TFLOP/s\n", name, blocks, threads, blocks * (threads / 64) / 1024, best, flop / best * 1e-9);
}

template <int MODE, int NV>
void run(const char* name, float* d, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    const int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mix<MODE, NV>), dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flop = (double)blocks * 4 * iters * 4 * (2.0 * 32 * 32 * 2);
    printf("%-58s waves/SIMD %d: %7.3f ms  %6.1f TFLOP/s\n", name, blocks / 256, best, flop / best * 1e-9);
}

int main() {
    float* d; (void)hipMalloc(&d, 4);
    for (int blocks : {512, 1024}) {
        run<0, 0>("MFMA only, distinct operand registers", d, blocks);
        run<1, 4>("+ 4 independent VALU / 4 MFMA", d, blocks);
        run<1, 8>("+ 8 independent VALU / 4 MFMA", d, blocks);
        run<1, 16>("+ 16 independent VALU / 4 MFMA", d, blocks);
        run<1, 32>("+ 32 independent VALU / 4 MFMA", d, blocks);
        run<2, 0>("+ 4 ds_read_b32 feeding next step (64x64 wave tile)", d, blocks);
        run<3, 0>("+ 4 ds_read_b32 + 4 VALU, not feeding MFMA", d, blocks);
        run<4, 0>("W8 shape: 3 ds_read_b32 per 2 MFMA (64x32 wave tile)", d, blocks);
    }
    float* g; (void)hipMalloc(&g, (size_t)512 * 8 * 4096 * 4); (void)hipMemset(g, 0, (size_t)512 * 8 * 4096 * 4);
    for (int threads : {512, 1024}) {
        const int blocks = threads == 512 ? 512 : 256;
        runq<0, 0, 1>("64x64 wave tile, barrier only", d, g, blocks, threads);
        runq<0, 8, 0>("64x64 wave tile, 8 glds, no barrier", d, g, blocks, threads);
        runq<0, 8, 1>("64x64 wave tile, 8 glds + barrier", d, g, blocks, threads);
        runq<0, 4, 1>("64x64 wave tile, 4 glds + barrier", d, g, blocks, threads);
        runq<1, 0, 1>("64x32 wave tile, barrier only", d, g, blocks, threads);
        runq<1, 4, 0>("64x32 wave tile, 4 glds, no barrier", d, g, blocks, threads);
        runq<1, 4, 1>("64x32 wave tile, 4 glds + barrier", d, g, blocks, threads);
        runq<0, -8, 1>("64x64 wave tile, 8 reg-staged loads + barrier", d, g, blocks, threads);
        runq<0, -4, 1>("64x64 wave tile, 4 reg-staged loads + barrier", d, g, blocks, threads);
        runq<1, -4, 1>("64x32 wave tile, 4 reg-staged loads + barrier", d, g, blocks, threads);
    }
    return 0;
}
