// stage_probe.hip -- what bounds kh_push of PAGEABLE host memory (round 6)?
// The path: caller's pageable buffer -> memcpy by T threads into a pinned chunk -> DMA to the device, chunk i + 1's memcpy beside
// chunk i's DMA.  Cases, each over 4 GiB of touched pageable memory in 64 MiB chunks:
//   (1) the memcpy alone, T = 1 .. 16 threads: glibc memcpy, and a loop of non-temporal 32-byte stores (no read-for-ownership of
//       the destination lines);
//   (2) the whole pipeline (memcpy + DMA overlapped through two / three pinned chunks) for the same T;
//   (3) no memcpy: hipHostRegister of the caller's pages slice by slice (T threads registering disjoint slices) + DMA from there.
//   hipcc --offload-arch=gfx950 -O2 -mavx2 -pthread -o stage_probe stage_probe.hip && ./stage_probe
#include <hip/hip_runtime.h>
#include <immintrin.h>

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

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void copy_nt(void *dst, const void *src, size_t n) {  // n a multiple of 64, both 32-byte aligned
    const __m256i *s = (const __m256i *)src;
    __m256i *d = (__m256i *)dst;
    for (size_t i = 0; i < n / 32; i += 2) {
        const __m256i a = _mm256_loadu_si256(s + i), b = _mm256_loadu_si256(s + i + 1);
        _mm256_stream_si256(d + i, a);
        _mm256_stream_si256(d + i + 1, b);
    }
    _mm_sfence();
}

template <typename F>
static void par_copy(void *dst, const void *src, size_t n, unsigned T, F f) {
    if (T <= 1) {
        f(dst, src, n);
        return;
    }
    const size_t per = ((n + T - 1) / T + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (unsigned i = 1; i < T; ++i) {
        const size_t off = (size_t)i * per;
        if (off >= n) break;
        const size_t len = std::min(per, n - off);
        th.emplace_back([=] { f((char *)dst + off, (const char *)src + off, len); });
    }
    f(dst, src, std::min(per, n));
    for (auto &t : th) t.join();
}

int main() {
    CK(hipSetDevice(0));
    const size_t TOTAL = 4ull << 30, CH = 64ull << 20;
    uint8_t *src = (uint8_t *)aligned_alloc(4096, TOTAL);
    for (size_t i = 0; i < TOTAL; i += 4096) src[i] = (uint8_t)i;  // touched
    uint8_t *pin[3], *dev = nullptr;
    for (auto &p : pin) CK(hipHostMalloc((void **)&p, CH, hipHostMallocDefault));
    CK(hipMalloc((void **)&dev, TOTAL));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t ev[3];
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    printf("host threads: %u\n", std::thread::hardware_concurrency());
    auto glibc = [](void *d, const void *s, size_t n) { memcpy(d, s, n); };
    auto nt = [](void *d, const void *s, size_t n) { copy_nt(d, s, n); };
    for (unsigned T : {1u, 2u, 4u, 6u, 8u, 12u, 16u}) {
        for (int kind = 0; kind < 2; ++kind) {
            double t0 = now();
            for (size_t o = 0; o < TOTAL; o += CH) {
                if (kind == 0) par_copy(pin[0], src + o, CH, T, glibc);
                else par_copy(pin[0], src + o, CH, T, nt);
            }
            const double a = now() - t0;
            // the pipeline: memcpy of chunk i + 1 beside the DMA of chunk i, NB pinned chunks
            for (int NB : {2, 3}) {
                CK(hipDeviceSynchronize());
                t0 = now();
                size_t i = 0;
                for (size_t o = 0; o < TOTAL; o += CH, ++i) {
                    const int b = (int)(i % NB);
                    if (i >= (size_t)NB) CK(hipEventSynchronize(ev[b]));
                    if (kind == 0) par_copy(pin[b], src + o, CH, T, glibc);
                    else par_copy(pin[b], src + o, CH, T, nt);
                    CK(hipMemcpyAsync(dev + o, pin[b], CH, hipMemcpyHostToDevice, st));
                    CK(hipEventRecord(ev[b], st));
                }
                CK(hipStreamSynchronize(st));
                const double p = now() - t0;
                if (NB == 2) printf("T = %2u %-22s memcpy alone %6.1f GB/s | pipeline with 2 chunks %6.1f GB/s", T, kind == 0 ? "glibc memcpy" : "non-temporal stores", TOTAL / a / 1e9, TOTAL / p / 1e9);
                else printf(" | with 3 chunks %6.1f GB/s\n", TOTAL / p / 1e9);
            }
        }
    }
    // (3) register the caller's pages in place, slice by slice, T threads at a time; DMA straight from them
    for (unsigned T : {1u, 2u, 4u, 8u}) {
        const size_t SL = 64ull << 20;
        CK(hipDeviceSynchronize());
        const double t0 = now();
        double treg = 0;
        for (size_t o = 0; o < TOTAL; o += SL * T) {
            const double r0 = now();
            std::vector<std::thread> th;
            for (unsigned i = 0; i < T && o + i * SL < TOTAL; ++i)
                th.emplace_back([=] { (void)hipSetDevice(0); CK(hipHostRegister(src + o + i * SL, SL, hipHostRegisterDefault)); });
            for (auto &t : th) t.join();
            treg += now() - r0;
            for (unsigned i = 0; i < T && o + i * SL < TOTAL; ++i) CK(hipMemcpyAsync(dev + o + i * SL, src + o + i * SL, SL, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            for (unsigned i = 0; i < T && o + i * SL < TOTAL; ++i) CK(hipHostUnregister(src + o + i * SL));
        }
        const double all = now() - t0;
        printf("register in place, %u slice(s) of 64 MiB at a time: registering %6.1f GB/s, register + DMA + unregister (serial) %6.1f GB/s\n", T, TOTAL / treg / 1e9, TOTAL / all / 1e9);
    }
    return 0;
}
