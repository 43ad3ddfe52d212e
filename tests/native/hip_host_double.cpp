// Host-memory double of the few HIP runtime entry points the C-ABI library's HOST half uses -- test infrastructure
// for `make asan` ("-fsanitize=address host build of the C-ABI shim"), never part of the product.
//
// The library's .hip sources are compiled with `hipcc --cuda-host-only -fsanitize=address,undefined` (host code only:
// argument validation, table builders, BatchNorm folding, k4 packing, workspace sizing, launch-strategy selection,
// error-string lifetime) and linked against THIS file instead of libamdhip64: "device" memory is host memory from
// calloc (so AddressSanitizer sees every upload / download the host half performs with its true extent), events are
// counters, and a kernel launch is counted and otherwise ignored -- no kernel arithmetic runs here, GPU results are
// the business of `pytest -m gpu`.  Runs in the GPU-less build container.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

static long g_launches = 0, g_allocs = 0, g_frees = 0, g_bytes = 0;
static dim3 g_grid, g_block;
static size_t g_shmem = 0;
static hipStream_t g_stream = nullptr;

extern "C" {

long hipdouble_launches() { return g_launches; }
long hipdouble_live_allocs() { return g_allocs - g_frees; }

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) {
    std::memset(p, 0, sizeof(*p));
    p->multiProcessorCount = 256;
    p->sharedMemPerBlock = 64 * 1024;
    p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
    std::snprintf(p->name, sizeof(p->name), "host double of gfx950");
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n) {
    *p = std::calloc(1, n ? n : 1);
    ++g_allocs;
    g_bytes += (long)n;
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
    if (p) ++g_frees;
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
// device globals: the host-only objects keep a host shadow of every __device__ variable; copy to / from that
hipError_t hipMemcpyToSymbol(const void* sym, const void* src, size_t n, size_t off, hipMemcpyKind) {
    std::memmove(const_cast<char*>(static_cast<const char*>(sym)) + off, src, n);
    return hipSuccess;
}
hipError_t hipMemcpyFromSymbol(void* dst, const void* sym, size_t n, size_t off, hipMemcpyKind) {
    std::memmove(dst, static_cast<const char*>(sym) + off, n);
    return hipSuccess;
}
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "error (host double)"; }

struct FakeEvent { long stamp; };
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(std::calloc(1, sizeof(FakeEvent))); ++g_allocs; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { ++g_frees; std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { reinterpret_cast<FakeEvent*>(e)->stamp = g_launches; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = 0.001f * (float)(reinterpret_cast<FakeEvent*>(b)->stamp - reinterpret_cast<FakeEvent*>(a)->stamp);
    return hipSuccess;
}

hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 2; return hipSuccess; }

hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t s) {
    g_grid = grid; g_block = block; g_shmem = shmem; g_stream = s;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* s) {
    *grid = g_grid; *block = g_block; *shmem = g_shmem; *s = g_stream;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void*, dim3 grid, dim3 block, void**, size_t, hipStream_t) {
    // what a launch the hardware would refuse looks like: empty or oversized grids / blocks are host-side bugs
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x * block.y * block.z == 0 || block.x * block.y * block.z > 1024) {
        std::fprintf(stderr, "hip double: invalid launch configuration grid (%u,%u,%u) block (%u,%u,%u)\n", grid.x, grid.y, grid.z,
                     block.x, block.y, block.z);
        std::abort();
    }
    ++g_launches;
    return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) { static void* handle[2]; return handle; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}

}  // extern "C"
