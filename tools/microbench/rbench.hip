// scratch micro-benchmark: what a pure read stream reaches on this chip (launch shape, bytes per lane, loads in flight, cache hints)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int U, int FLAV>   // U 16-byte loads in flight per lane; FLAV 0 plain, 1 nt, 2 sc1, 3 sc0 sc1
__global__ __launch_bounds__(1024) void rk(const u32x4* src, size_t n16, uint32_t* sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * stride;
            const u32x4* p = src + (j < n16 ? j : 0);
            if (FLAV == 0) v[u] = *p;
            if (FLAV == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v[u]) : "v"(p));
            if (FLAV == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(p));
            if (FLAV == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[u]) : "v"(p));
        }
        if (FLAV != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; ++u) { asm volatile("" : "+v"(v[u])); acc ^= v[u]; }
    }
    if (acc.x == 0x12345 && acc.y == 7) sink[0] = acc.z + acc.w;
}
// contiguous chunk per workgroup (the tagger's / pass A's shape: a workgroup walks its own part of the array)
template <int U>
__global__ __launch_bounds__(1024) void rk_chunk(const u32x4* src, size_t n16, uint32_t* sink) {
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x, a = per * blockIdx.x, b = a + per < n16 ? a + per : n16;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = a + threadIdx.x; i < b; i += (size_t)blockDim.x * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const size_t j = i + (size_t)u * blockDim.x; v[u] = src[j < b ? j : a]; }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if (acc.x == 0x12345 && acc.y == 7) sink[0] = acc.z + acc.w;
}

int main() {
    const size_t bytes = (size_t)24 << 30;
    u32x4* src; uint32_t* sink;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 1, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t n16 = bytes / 16;
#define RUN(KERN, GRID, BLOCK, NAME) { \
    for (int rep = 0; rep < 2; ++rep) { CK(hipEventRecord(a)); hipLaunchKernelGGL(KERN, dim3(GRID), dim3(BLOCK), 0, 0, src, n16, sink); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); } \
    float ms; CK(hipEventElapsedTime(&ms, a, b)); printf("%-52s %7.3f ms  %.2f TB/s\n", NAME, ms, bytes / ms / 1e9); }
    RUN((rk<1, 0>), 256 * 2, 1024, "grid-stride 512 x 1024, 1 load in flight")
    RUN((rk<4, 0>), 256 * 2, 1024, "grid-stride 512 x 1024, 4 in flight")
    RUN((rk<8, 0>), 256 * 2, 1024, "grid-stride 512 x 1024, 8 in flight")
    RUN((rk<4, 0>), 256, 1024, "grid-stride 256 x 1024, 4 in flight")
    RUN((rk<8, 0>), 256, 1024, "grid-stride 256 x 1024, 8 in flight")
    RUN((rk<4, 0>), 256 * 8, 256, "grid-stride 2048 x 256, 4 in flight")
    RUN((rk<4, 0>), 256 * 32, 256, "grid-stride 8192 x 256, 4 in flight")
    RUN((rk<4, 1>), 256 * 2, 1024, "grid-stride 512 x 1024, 4 in flight, nt")
    RUN((rk<4, 2>), 256 * 2, 1024, "grid-stride 512 x 1024, 4 in flight, sc1")
    RUN((rk<4, 3>), 256 * 2, 1024, "grid-stride 512 x 1024, 4 in flight, sc0 sc1")
    RUN((rk<8, 1>), 256, 1024, "grid-stride 256 x 1024, 8 in flight, nt")
    RUN((rk_chunk<4>), 256, 1024, "chunk per workgroup 256 x 1024, 4 in flight")
    RUN((rk_chunk<8>), 256, 1024, "chunk per workgroup 256 x 1024, 8 in flight")
    RUN((rk_chunk<4>), 256 * 16, 1024, "chunk per workgroup 4096 x 1024, 4 in flight")
    RUN((rk_chunk<4>), 256 * 64, 256, "chunk per workgroup 16384 x 256, 4 in flight")
    return 0;
}
