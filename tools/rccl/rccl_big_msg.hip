// rccl_big_msg.hip -- does RCCL deliver a >= 2^30-byte point-to-point message whole?  No library of this repository involved.
//
// Round 4 (tools/world1_merge_probe.py, through the whole library): a rank's ncclSend / ncclRecv to ITSELF in a world of one
// lost the upper half of a 1.09 GB message.  This is the stand-alone form: one communicator of one rank, one ncclSend +
// ncclRecv to self inside a group, for sizes on both sides of 2^30 and 2^31 bytes; every byte of the destination is compared
// on the device with the pattern the source was filled with.  Also the same bytes as uint32 / uint64 elements (a count below
// 2^30 / 2^29 for the same message) and as a split into <= 256 MiB messages (what exchange.hip does).
//
//   hipcc --offload-arch=gfx950 -O2 -o rccl_big_msg rccl_big_msg.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
//   ./rccl_big_msg            (prints one line per case and a verdict; exit 0 = every case whole)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIPCHECK(x)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                          \
            exit(2);                                                                         \
        }                                                                                    \
    } while (0)
#define NCCLCHECK(x)                                                                         \
    do {                                                                                     \
        ncclResult_t r_ = (x);                                                               \
        if (r_ != ncclSuccess) {                                                             \
            fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_));                         \
            exit(2);                                                                         \
        }                                                                                    \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
    return x;
}
__global__ void fill_kernel(uint64_t *p, uint64_t nwords, uint64_t salt) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) p[i] = mix(i ^ salt);
}
// bad[0] = words that differ, bad[1] = index of the first one (min)
__global__ void check_kernel(const uint64_t *p, uint64_t nwords, uint64_t salt, unsigned long long *bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long n = 0, first = ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride)
        if (p[i] != mix(i ^ salt)) {
            ++n;
            if (i < first) first = i;
        }
    if (n) {
        atomicAdd(&bad[0], n);
        atomicMin(&bad[1], first);
    }
}

int main() {
    HIPCHECK(hipSetDevice(0));
    int ver = 0;
    NCCLCHECK(ncclGetVersion(&ver));
    ncclUniqueId id;
    NCCLCHECK(ncclGetUniqueId(&id));
    ncclComm_t comm;
    NCCLCHECK(ncclCommInitRank(&comm, 1, id, 0));
    int cnt = 0;
    NCCLCHECK(ncclCommCount(comm, &cnt));
    hipStream_t st;
    HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    printf("RCCL version code %d, world %d (ncclCommCount), one rank sending to itself\n", ver, cnt);

    const uint64_t MAXB = 3ull << 30;
    uint64_t *src = nullptr, *dst = nullptr;
    unsigned long long *bad = nullptr;
    HIPCHECK(hipMalloc((void **)&src, MAXB));
    HIPCHECK(hipMalloc((void **)&dst, MAXB));
    HIPCHECK(hipMalloc((void **)&bad, 16));

    struct Case { const char *name; uint64_t bytes; int elem; uint64_t split; };
    const uint64_t G = 1ull << 30;
    std::vector<Case> cases = {
        {"2^29 bytes, uint8", G / 2, 1, 0},
        {"2^30 - 64 bytes, uint8", G - 64, 1, 0},
        {"2^30 bytes, uint8", G, 1, 0},
        {"2^30 + 64 bytes, uint8", G + 64, 1, 0},
        {"2^31 bytes, uint8", 2 * G, 1, 0},
        {"3 x 2^30 bytes, uint8", 3 * G, 1, 0},
        {"2^30 bytes as uint32 (2^28 elements)", G, 4, 0},
        {"2^31 bytes as uint32 (2^29 elements)", 2 * G, 4, 0},
        {"3 x 2^30 bytes as uint64", 3 * G, 8, 0},
        {"3 x 2^30 bytes, uint8, in 256 MiB messages", 3 * G, 1, 256ull << 20},
        {"3 x 2^30 bytes, uint8, in 1 GiB - 64 B messages", 3 * G, 1, G - 64},
    };
    int failures = 0;
    uint64_t salt = 0x1234;
    for (const Case &cs : cases) {
        ++salt;
        const uint64_t nw = cs.bytes / 8;
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, st, src, nw, salt);
        HIPCHECK(hipMemsetAsync(dst, 0xA5, cs.bytes, st));
        const unsigned long long init[2] = {0ull, ~0ull};
        HIPCHECK(hipMemcpyAsync(bad, init, 16, hipMemcpyHostToDevice, st));
        const ncclDataType_t dt = cs.elem == 1 ? ncclUint8 : cs.elem == 4 ? ncclUint32 : ncclUint64;
        const uint64_t step = cs.split ? cs.split : cs.bytes;
        NCCLCHECK(ncclGroupStart());
        for (uint64_t o = 0; o < cs.bytes; o += step) {
            const uint64_t b = cs.bytes - o < step ? cs.bytes - o : step;
            NCCLCHECK(ncclSend((const char *)src + o, (size_t)(b / cs.elem), dt, 0, comm, st));
        }
        for (uint64_t o = 0; o < cs.bytes; o += step) {
            const uint64_t b = cs.bytes - o < step ? cs.bytes - o : step;
            NCCLCHECK(ncclRecv((char *)dst + o, (size_t)(b / cs.elem), dt, 0, comm, st));
        }
        NCCLCHECK(ncclGroupEnd());
        hipLaunchKernelGGL(check_kernel, dim3(2048), dim3(256), 0, st, dst, nw, salt, bad);
        unsigned long long h[2];
        HIPCHECK(hipMemcpyAsync(h, bad, 16, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        if (h[0]) {
            ++failures;
            printf("LOST   %-52s %llu of %llu words differ, the first at byte %llu (%.4f of the message)\n", cs.name, h[0], (unsigned long long)nw,
                   h[1] * 8ull, (double)h[1] * 8.0 / (double)cs.bytes);
        } else {
            printf("whole  %-52s\n", cs.name);
        }
    }
    printf(failures ? "VERDICT: %d case(s) lost data -- the transport, not this repository's offset arithmetic\n"
                    : "VERDICT: every message arrived whole (%d lost) -- a loss seen through the library is in the library\n",
           failures);
    NCCLCHECK(ncclCommDestroy(comm));
    return failures ? 1 : 0;
}
