// Micro-benchmark: what does the memory system sustain for level 1's store pattern?  256 workgroups x 1024 lanes,
// every lane owns a stream of 1-KiB chunks taken from a shared pool (like the partition pool) and appends RUN bytes
// (16-byte stores) per iteration.  Prints GB/s for several run lengths, chunks handed out in workgroup-private
// ranges (as level 1 does) or every lane streaming through its own contiguous region.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o scatter_runs scatter_runs.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
template <int UNITS, bool CONTIG, int WORK>
__global__ __launch_bounds__(1024) void k(uint4 *pool, u64 *next, u64 pool_chunks, int iters) {
    __shared__ u64 s_next;
    const int tid = threadIdx.x;
    if (tid == 0) s_next = atomicAdd(next, (u64)(iters * UNITS / 64 + 2) * 1024);
    __syncthreads();
    u64 cur = CONTIG ? ((u64)blockIdx.x * 1024 + tid) * (u64)(iters * UNITS / 64 + 2) : 0;
    uint32_t fill = CONTIG ? 0 : 64;  // units in the current chunk (64 x 16 B = 1 KiB)
    uint4 x = make_uint4(tid, blockIdx.x, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int w = 0; w < WORK; ++w) x.z = x.z * 1664525u + 1013904223u;  // stand-in for the compute between flushes
        for (int u = 0; u < UNITS; ++u) {
            if (fill == 64) {
                cur = CONTIG ? cur + 1 : atomicAdd(&s_next, 1ull);
                fill = 0;
            }
            x.w = it;
            pool[cur * 64 + fill] = x;
            ++fill;
        }
        __syncthreads();
    }
}
template <typename F>
static float run(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    const u64 bytes = 16ull << 30;
    uint4 *pool; u64 *next;
    hipMalloc(&pool, bytes + (64ull << 20)); hipMalloc(&next, 8);
    hipMemset(pool, 0, bytes);
#define CASE(UNITS, CONTIG, WORK) { \
        const int iters = (int)(bytes / (256ull * 1024 * 16 * UNITS)) * 3 / 4; \
        hipMemset(next, 0, 8); \
        run([&] { hipLaunchKernelGGL((k<UNITS, CONTIG, WORK>), dim3(256), dim3(1024), 0, 0, pool, next, bytes / 1024, 8); }); \
        hipMemset(next, 0, 8); \
        const float ms = run([&] { hipLaunchKernelGGL((k<UNITS, CONTIG, WORK>), dim3(256), dim3(1024), 0, 0, pool, next, bytes / 1024, iters); }); \
        printf("run %3d B per lane per iteration, %s, work %4d: %7.2f ms for %5.1f GB = %6.0f GB/s\n", 16 * UNITS, \
               CONTIG ? "own contiguous region" : "1-KiB chunks from WG range", WORK, ms, 256.0 * 1024 * 16 * UNITS * iters / 1e9, \
               256.0 * 1024 * 16 * UNITS * iters / 1e6 / ms); }
    CASE(1, false, 0) CASE(2, false, 0) CASE(3, false, 0) CASE(4, false, 0) CASE(8, false, 0) CASE(16, false, 0)
    CASE(4, true, 0) CASE(8, true, 0)
    CASE(4, false, 2000) CASE(4, false, 6000)
    return 0;
}
