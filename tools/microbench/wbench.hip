// scratch micro-benchmark: how the memory system takes pass A's write pattern (65 536 parts, each filled front to back, one chunk per
// part and step) beside a read stream.  modes: chunk dwords / alignment / store flavour.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int CH, int FLAV, bool READ>   // CH dwords per chunk (lanes of a wave are split into 64 / min(CH,64)... see below)
__global__ __launch_bounds__(1024) void wk(uint32_t* parts, size_t part_words, uint32_t steps, uint32_t shift, const uint4* rd, size_t rd_per_wg_step, uint32_t* sink) {
    const uint32_t tid = threadIdx.x, wg = blockIdx.x;
    uint4 acc = {0, 0, 0, 0};
    for (uint32_t t = 0; t < steps; ++t) {
        if (READ) {
            const uint4* src = rd + ((size_t)t * gridDim.x + wg) * rd_per_wg_step;
            for (uint32_t i = tid; i < rd_per_wg_step; i += 1024) { uint4 v = src[i]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
        }
        // 256 parts per workgroup, CH dwords each: 256 * CH dwords per step
        for (uint32_t i = tid; i < 256 * CH; i += 1024) {
            const uint32_t p = i / CH, j = i % CH;
            uint32_t* dst = parts + ((size_t)wg * 256 + p) * part_words + shift + (size_t)t * CH + j;
            const uint32_t e = i ^ t;
            if (FLAV == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(e) : "memory");
            if (FLAV == 1) asm volatile("global_store_dword %0, %1, off nt" ::"v"(dst), "v"(e) : "memory");
            if (FLAV == 2) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(dst), "v"(e) : "memory");
            if (FLAV == 3) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(dst), "v"(e) : "memory");
            if (FLAV == 4) asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(e) : "memory");
        }
    }
    if (acc.x == 0x12345 && acc.y == 7) sink[0] = acc.z + acc.w;
}

int main(int argc, char** argv) {
    const uint32_t steps = 400;
    const int n_wg = 256;
    const size_t part_words = 32 * steps + 64;   // room for 32-dword chunks
    uint32_t* parts; uint4* rd; uint32_t* sink;
    const size_t rd_per = 78 * 1024 / 16;   // 78 KiB per workgroup and step
    CK(hipMalloc(&parts, (size_t)n_wg * 256 * part_words * 4 + 4096));
    CK(hipMalloc(&rd, (size_t)steps * n_wg * rd_per * 16));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(rd, 1, (size_t)steps * n_wg * rd_per * 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
#define RUN(CH, FLAV, READ, SHIFT, NAME) { \
    for (int rep = 0; rep < 2; ++rep) { CK(hipEventRecord(a)); hipLaunchKernelGGL((wk<CH, FLAV, READ>), dim3(n_wg), dim3(1024), 0, 0, parts, part_words, steps, SHIFT, rd, rd_per, sink); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); } \
    float ms; CK(hipEventElapsedTime(&ms, a, b)); \
    const double wb = (double)steps * n_wg * 256 * CH * 4, rb = READ ? (double)steps * n_wg * rd_per * 16 : 0; \
    printf("%-44s %7.3f ms  write %.2f GB  read %.2f GB  -> %.2f TB/s\n", NAME, ms, wb / 1e9, rb / 1e9, (wb + rb) / ms / 1e9); }
    RUN(24, 0, false, 0, "w24 plain, no read")
    RUN(16, 0, false, 0, "w16 plain, no read")
    RUN(32, 0, false, 0, "w32 plain, no read")
    RUN(24, 0, true, 0, "w24 plain + read")
    RUN(24, 0, true, 5, "w24 shifted + read")
    RUN(16, 0, true, 0, "w16 aligned + read")
    RUN(32, 0, true, 0, "w32 aligned + read")
    RUN(24, 1, true, 0, "w24 nt + read")
    RUN(16, 1, true, 0, "w16 nt + read")
    RUN(32, 1, true, 0, "w32 nt + read")
    RUN(24, 2, true, 0, "w24 sc0sc1 + read")
    RUN(16, 2, true, 0, "w16 sc0sc1 + read")
    RUN(32, 2, true, 0, "w32 sc0sc1 + read")
    RUN(16, 3, true, 0, "w16 sc1 + read")
    RUN(16, 4, true, 0, "w16 sc0sc1nt + read")
    RUN(32, 4, true, 0, "w32 sc0sc1nt + read")
    RUN(1, 0, true, 0, "w1 (read only, almost)")
    return 0;
}
