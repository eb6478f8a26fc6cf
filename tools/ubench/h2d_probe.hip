// h2d_probe.hip -- how fast does pinned host memory reach HBM, and by which engine?  (round 5: the command line moves a 31.6 GB
// file at ~40 GB/s where the link takes ~57.)  Cases: hipMemcpyAsync on 1 / 2 / 4 streams (halves / quarters of one buffer),
// a copy KERNEL reading the pinned buffer over the link (zero-copy loads), and SDMA + kernel side by side.
//   hipcc --offload-arch=gfx950 -O2 -o h2d_probe h2d_probe.hip && ./h2d_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                \
    do {                                                                     \
        hipError_t e_ = (x);                                                 \
        if (e_ != hipSuccess) {                                              \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));          \
            exit(2);                                                         \
        }                                                                    \
    } while (0)

__global__ void copy_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const uint64_t N = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 1024ull) << 20;  // MiB
    CK(hipSetDevice(0));
    uint8_t *h = nullptr, *d = nullptr;
    CK(hipHostMalloc((void **)&h, N, hipHostMallocDefault));
    memset(h, 0x5a, N);
    CK(hipMalloc((void **)&d, N));
    hipStream_t st[8];
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto run = [&](const char *name, int reps, auto &&body) {
        body();  // warm
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int r = 0; r < reps; ++r) body();
        CK(hipDeviceSynchronize());
        const double dt = (now() - t0) / reps;
        printf("%-58s %7.2f ms  %6.1f GB/s\n", name, dt * 1e3, (double)N / dt / 1e9);
    };
    for (int ns : {1, 2, 4, 8}) {
        char name[96];
        snprintf(name, sizeof(name), "hipMemcpyAsync, %d stream(s), %llu MiB in equal parts", ns, (unsigned long long)(N >> 20));
        run(name, 5, [&] {
            const uint64_t part = (N / ns) & ~4095ull;
            for (int i = 0; i < ns; ++i) {
                const uint64_t o = part * i, len = i == ns - 1 ? N - o : part;
                CK(hipMemcpyAsync(d + o, h + o, len, hipMemcpyHostToDevice, st[i]));
            }
            for (int i = 0; i < ns; ++i) CK(hipStreamSynchronize(st[i]));
        });
    }
    for (int wg : {256, 1024, 4096}) {
        char name[96];
        snprintf(name, sizeof(name), "copy kernel (16-byte loads of pinned memory), %d workgroups", wg);
        run(name, 5, [&] {
            hipLaunchKernelGGL(copy_kernel, dim3(wg), dim3(256), 0, st[0], (uint4 *)d, (const uint4 *)h, N / 16);
            CK(hipStreamSynchronize(st[0]));
        });
    }
    for (int pct : {30, 40, 50}) {
        char name[96];
        snprintf(name, sizeof(name), "SDMA (1 stream) + copy kernel side by side, kernel takes %d %%", pct);
        run(name, 5, [&] {
            const uint64_t ksz = (N * pct / 100) & ~4095ull;
            CK(hipMemcpyAsync(d + ksz, h + ksz, N - ksz, hipMemcpyHostToDevice, st[0]));
            hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, st[1], (uint4 *)d, (const uint4 *)h, ksz / 16);
            CK(hipStreamSynchronize(st[0]));
            CK(hipStreamSynchronize(st[1]));
        });
    }
    // pinning: what it costs to register memory the caller already holds (kh_host_register), per GiB
    {
        const uint64_t M = 1ull << 30;
        uint8_t *p = (uint8_t *)aligned_alloc(4096, M);
        memset(p, 1, M);
        const double t0 = now();
        CK(hipHostRegister(p, M, hipHostRegisterDefault));
        const double t1 = now();
        CK(hipHostUnregister(p));
        const double t2 = now();
        printf("hipHostRegister of 1 GiB of touched pageable memory: %.1f ms (%.1f GB/s), unregister %.1f ms\n", (t1 - t0) * 1e3, 1.0737 / (t1 - t0), (t2 - t1) * 1e3);
        // ... and the same in 16 MiB slices
        const double t3 = now();
        for (uint64_t o = 0; o < M; o += 16u << 20) CK(hipHostRegister(p + o, 16u << 20, hipHostRegisterDefault));
        const double t4 = now();
        for (uint64_t o = 0; o < M; o += 16u << 20) CK(hipHostUnregister(p + o));
        printf("the same in 64 slices of 16 MiB: %.1f ms (%.1f GB/s)\n", (t4 - t3) * 1e3, 1.0737 / (t4 - t3));
        // pageable hipMemcpy (the runtime's own staging)
        const double t5 = now();
        CK(hipMemcpy(d, p, M, hipMemcpyHostToDevice));
        const double t6 = now();
        printf("hipMemcpy of 1 GiB of pageable memory (the runtime's staging): %.1f ms (%.1f GB/s)\n", (t6 - t5) * 1e3, 1.0737 / (t6 - t5));
        free(p);
    }
    // ---- the command line's loop in miniature: 128 MiB pinned chunks, pushed one after the other (each waited for), while
    // helper threads fill the OTHER chunk buffer from ordinary memory (what the pread()s do) ----
    {
        const uint64_t CH = 128ull << 20;
        const int NCH = 48;
        uint8_t *pb[2], *src = (uint8_t *)aligned_alloc(4096, CH);
        memset(src, 3, CH);
        for (auto &p : pb) {
            CK(hipHostMalloc((void **)&p, CH, hipHostMallocPortable));
            memset(p, 7, CH);
        }
        for (int fillers : {0, 4, 12}) {
            for (int split : {1, 2}) {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                for (int i = 0; i < NCH; ++i) {
                    std::vector<std::thread> th;
                    uint8_t *other = pb[(i + 1) & 1];
                    for (int t = 0; t < fillers; ++t)
                        th.emplace_back([=] { memcpy(other + (CH / fillers) * t, src + (CH / fillers) * t, CH / fillers); });
                    uint8_t *cur = pb[i & 1];
                    if (split == 1) {
                        CK(hipMemcpyAsync(d, cur, CH, hipMemcpyHostToDevice, st[0]));
                    } else {
                        CK(hipMemcpyAsync(d, cur, CH / 2, hipMemcpyHostToDevice, st[0]));
                        CK(hipMemcpyAsync(d + CH / 2, cur + CH / 2, CH / 2, hipMemcpyHostToDevice, st[1]));
                        CK(hipStreamSynchronize(st[1]));
                    }
                    CK(hipStreamSynchronize(st[0]));
                    for (auto &t : th) t.join();
                }
                const double dt = now() - t0;
                printf("chunk loop: %d x 128 MiB, %d stream(s) per chunk, %2d filler threads on the other buffer: %6.1f GB/s\n", NCH, split, fillers,
                       (double)CH * NCH / dt / 1e9);
            }
        }
    }
    // ---- pinning the caller's pageable memory in place, slices registered by several threads at once ----
    {
        const uint64_t M = 4ull << 30, SL = 64ull << 20;
        uint8_t *p = (uint8_t *)aligned_alloc(1 << 21, M);
        memset(p, 1, M);
        for (int nt : {1, 2, 4, 8}) {
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([=] {
                    (void)hipSetDevice(0);
                    for (uint64_t o = SL * t; o < M; o += SL * nt)
                        if (hipHostRegister(p + o, SL, hipHostRegisterDefault) != hipSuccess) fprintf(stderr, "register failed\n");
                });
            for (auto &t : th) t.join();
            const double t1 = now();
            // one DMA out of the registered range, to see that it is usable
            CK(hipMemcpyAsync(d, p, std::min<uint64_t>(N, M), hipMemcpyHostToDevice, st[0]));
            CK(hipStreamSynchronize(st[0]));
            const double t2 = now();
            for (uint64_t o = 0; o < M; o += SL) CK(hipHostUnregister(p + o));
            const double t3 = now();
            printf("hipHostRegister of 4 GiB in 64 MiB slices by %d thread(s): %7.1f ms (%5.1f GB/s); DMA of the first GiB %5.1f GB/s; unregister %6.1f ms\n", nt,
                   (t1 - t0) * 1e3, (double)M / (t1 - t0) / 1e9, (double)std::min<uint64_t>(N, M) / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
        }
        free(p);
    }
    return 0;
}
