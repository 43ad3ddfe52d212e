// Probe: LDS-DMA (buffer_load ... lds, 16 B per lane) on gfx950 -- destination above 64 KB, out-of-range lanes, vmcnt.
//   hipcc --offload-arch=gfx950 -O3 tools/native/glds_probe.hip -o tools/native/glds_probe && tools/native/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void probe(const float* src, int src_bytes, float* out, int lds_byte_off) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 36 * 1024; i += 256) smem[i] = -7.f;  // 144 KB sentinel
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
    // lane l of wave wv loads float4 number (wv * 64 + (l ^ 1)) -- a source-side permutation; every 5th lane is out of range
    unsigned voff = (unsigned)((wv * 64 + (lane ^ 1)) * 16);
    if (lane % 5 == 4) voff = 0x80000000u;
    float* dst = smem + lds_byte_off / 4 + wv * 256;  // wave-uniform
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = tid; i < 1024; i += 256) out[i] = smem[lds_byte_off / 4 + i];
}
int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *src, *out;
    (void)hipMalloc(&src, 4096); (void)hipMalloc(&out, 4096);
    (void)hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    for (int off : {0, 65536 + 4096, 131072 + 8192}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 147456, 0, src, 4096, out, off);
        std::vector<float> r(1024);
        (void)hipMemcpy(r.data(), out, 4096, hipMemcpyDeviceToHost);
        int ok = 0, zero = 0, sentinel = 0, bad = 0;
        for (int q = 0; q < 256; ++q) {  // float4 slot q was written by lane q % 64 of wave q / 64
            const int l = q & 63, srcq = (q & ~63) + (l ^ 1);
            for (int e = 0; e < 4; ++e) {
                const float v = r[q * 4 + e];
                if (l % 5 == 4) { if (v == 0.f) ++zero; else if (v == -7.f) ++sentinel; else ++bad; }
                else if (v == (float)(srcq * 4 + e)) ++ok; else ++bad;
            }
        }
        printf("lds offset %6d: in-range ok %d / %d; out-of-range lanes: zero %d, untouched %d; wrong %d  (err %s)\n", off, ok, 4 * (256 - 52), zero, sentinel, bad, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
