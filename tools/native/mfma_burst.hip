// What does one wave per SIMD lose per chunk of 16 dependent v_mfma_f32_32x32x2_f32 when the chunk also carries the
// staging work of the 32-row stream-K tile?  Variants add one ingredient at a time (all 1024 SIMDs busy, one wave each):
//   0 MFMAs only   1 + 8 ds_read_b128 feeding the NEXT chunk   2 + s_barrier   3 + 5 ds_write_b128   4 + 5 buffer loads
//   hipcc --offload-arch=gfx950 -O3 tools/native/mfma_burst.hip -o tools/native/mfma_burst
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int V>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 10240; i += 256) smem[i] = 0.001f * (float)(i & 15);
    __syncthreads();
    f32x16 acc = f32x16{0};
    float4 a[2][4], b[2][4];
    const int l31 = tid & 31, lhi = (tid >> 5) & 1, swz = (l31 >> 1) & 7;
    const float* ap = smem + l31 * 32;  // the kernel's A image: row l31, 16-byte slot (2 kg + lhi) ^ swz: conflict-free
#define AOFF(kg) ((((2 * (kg) + lhi) ^ swz) << 2))
    const float* bp = smem + 4096 + tid * 4;
    float* wp = smem + 5120 + tid * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 24, 0x00020000);
    i32x4 r[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) r[i] = i32x4{0, 0, 0, 0};
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) { a[0][kg] = *(const float4*)(ap + AOFF(kg)); b[0][kg] = *(const float4*)(bp + kg * 1024); }
    unsigned voff = (unsigned)(blockIdx.x * 4096 + tid * 16);
    auto chunk = [&](auto ptag) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value;
        __builtin_amdgcn_sched_barrier(0);
        if ((V & 2) && !(V & 16)) __syncthreads();
        if (V & 1) {
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) { a[1 - P][kg] = *(const float4*)(ap + AOFF(kg) + (1 - P) * 1024); b[1 - P][kg] = *(const float4*)(bp + kg * 1024 + (1 - P) * 16); }
        }
        if (V & 4) {
#pragma unroll
            for (int i = 0; i < 5; ++i) *(i32x4*)(wp + i * 1024) = r[i];
        }
        if (V & 8) {
#pragma unroll
            for (int i = 0; i < 5; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + i * 65536, 0, 0);
            voff = (voff + 1024) & 0xFFFFF;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            if ((V & 16) && kg == 2) {
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
            const float4 x = a[(V & 1) ? P : 0][kg], y = b[(V & 1) ? P : 0][kg];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, y.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, y.w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int i = 0; i < iters; ++i) {
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[e];
    if (s == 12345.f) out[0] = s;
}
template <typename K>
static void run(const char* name, K kern, float* d, float* src) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 84 * 1024, 0, d, src, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 84 * 1024, 0, d, src, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  %7.1f cycles per chunk of 16 MFMAs at 2.4 GHz (1024 = the MFMAs alone)\n", name, ms, ms * 1e-3 * 2.4e9 / (2.0 * iters));
}

// Wave-specialised block of 8 waves: waves 0-3 (one per SIMD) only read operands and multiply, waves 4-7 (the second wave of
// each SIMD) only stage: 5 buffer loads + 5 ds_write_b128 per thread and chunk.  One s_barrier per chunk for all.
template <int PRIO, int MID>
__global__ __launch_bounds__(512, 2) void ks(float* out, const float* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 10240; i += 512) smem[i] = 0.001f * (float)(i & 15);
    __syncthreads();
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wid >= 4) {  // producers
        const int t = tid - 256;
        float* wp = smem + 5120 + t * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 24, 0x00020000);
        i32x4 r[2][5];
        unsigned voff = (unsigned)(blockIdx.x * 4096 + t * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 5; ++i) r[j][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + i * 65536, 0, 0);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int i = 0; i < 5; ++i) *(i32x4*)(wp + i * 1024) = r[j][i];
#pragma unroll
                for (int i = 0; i < 5; ++i) r[j][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + i * 65536, 0, 0);
                voff = (voff + 1024) & 0xFFFFF;
                __syncthreads();
            }
        }
        return;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    f32x16 acc = f32x16{0};
    float4 a[2][4], b[2][4];
    const int l31 = tid & 31, lhi = (tid >> 5) & 1, swz = (l31 >> 1) & 7;
    const float* ap = smem + l31 * 32;
    const float* bp = smem + 4096 + tid * 4;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) { a[0][kg] = *(const float4*)(ap + AOFF(kg)); b[0][kg] = *(const float4*)(bp + kg * 1024); }
    auto chunk = [&](auto ptag) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value;
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (!MID) {
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) { a[1 - P][kg] = *(const float4*)(ap + AOFF(kg) + (1 - P) * 1024); b[1 - P][kg] = *(const float4*)(bp + kg * 1024 + (1 - P) * 16); }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            if (MID && kg == MID) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; ++g) { a[1 - P][g] = *(const float4*)(ap + AOFF(g) + (1 - P) * 1024); b[1 - P][g] = *(const float4*)(bp + g * 1024 + (1 - P) * 16); }
                __builtin_amdgcn_sched_barrier(0);
            }
            const float4 x = a[P][kg], y = b[P][kg];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, y.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, y.w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int i = 0; i < iters; ++i) {
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[e];
    if (s == 12345.f) out[0] = s;
}
template <typename K>
static void run512(const char* name, K kern, float* d, float* src) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 84 * 1024, 0, d, src, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 84 * 1024, 0, d, src, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  %7.1f cycles per chunk of 16 MFMAs at 2.4 GHz (1024 = the MFMAs alone)\n", name, ms, ms * 1e-3 * 2.4e9 / (2.0 * iters));
}
int main() {
    float *d, *src;
    (void)hipMalloc(&d, 64);
    (void)hipMalloc(&src, 1 << 24);
    (void)hipMemset(src, 0, 1 << 24);
    printf("features: 1 = 8 ds_read_b128 for the next chunk, 2 = s_barrier, 4 = 5 ds_write_b128, 8 = 5 buffer_load_dwordx4\n");
    run("0", k<0>, d, src); run("1 reads", k<1>, d, src); run("2 barrier", k<2>, d, src); run("3 reads+barrier", k<3>, d, src);
    run("4 writes", k<4>, d, src); run("5 reads+writes", k<5>, d, src); run("6 barrier+writes", k<6>, d, src);
    run("7 reads+barrier+writes", k<7>, d, src); run("8 loads", k<8>, d, src); run("9 reads+loads", k<9>, d, src);
    run("12 writes+loads", k<12>, d, src); run("13 reads+writes+loads (wave-private pipeline)", k<13>, d, src);
    run("14 barrier+writes+loads", k<14>, d, src); run("15 all", k<15>, d, src);
    run("19 reads + barrier after 8 MFMAs", k<19>, d, src); run("23 reads+writes + barrier after 8 MFMAs", k<23>, d, src);
    run("31 all, barrier after 8 MFMAs", k<31>, d, src);
    run512("wave-specialised: 4 MFMA waves + 4 staging waves", ks<0, 0>, d, src);
    run512("  same, s_setprio 3 on the MFMA waves", ks<1, 0>, d, src);
    run512("  operand reads after 4 of the 16 MFMAs", ks<0, 1>, d, src);
    run512("  operand reads after 8 of the 16 MFMAs", ks<0, 2>, d, src);
    run512("  operand reads after 12 of the 16 MFMAs", ks<0, 3>, d, src);
    return 0;
}
