// level1_64.hip -- level 1 with 8-byte payloads (k >= 22, or KMERHIP_PAYLOAD=64): part1_bins64_kernel, one instance per
// k = 22..32 with the written-out window (window.hip.h), the C++ window for every other case.
#define KH_HELPERS_ONLY 1
#include "level1_api.h"
#include "level1.hip.h"

namespace kh {

namespace {
#define KH_L1_ARGS l.abase, l.qbase, l.qaligned, l.vbeg, l.vend, l.wlo, l.tile0, l.ntiles, l.tiles_per_block, l.k, l.thr, l.g, \
                   (u64 *)l.pool, l.chunk_part, l.fill8, l.pool_next, l.pool_chunks, l.ctr
// bins of 16 payloads are flushed every 2 windows per lane; with -Q (fewer windows survive) every 4
template <int MODE, int KW>
void launch_bins(const L1Launch &l) {
    if (l.use_qual) hipLaunchKernelGGL((part1_bins64_kernel<true, MODE, KW, 4>), dim3(l.grid), dim3(PART_NT), 0, l.stream, KH_L1_ARGS);
    else hipLaunchKernelGGL((part1_bins64_kernel<false, MODE, KW, 2>), dim3(l.grid), dim3(PART_NT), 0, l.stream, KH_L1_ARGS);
}
#if KH_TESTING  // round 1's tile-sorting kernel: reachable through a test-build switch only, compiled into the test build only
template <int MODE>
void launch_legacy(const L1Launch &l) {
    if (l.use_qual) hipLaunchKernelGGL((part1_scatter_chunked_kernel<true, MODE, false, u64, 0>), dim3(l.grid), dim3(PART_NT), 0, l.stream, KH_L1_ARGS);
    else hipLaunchKernelGGL((part1_scatter_chunked_kernel<false, MODE, false, u64, 0>), dim3(l.grid), dim3(PART_NT), 0, l.stream, KH_L1_ARGS);
}
#endif
}  // namespace

void launch_level1_64(const L1Launch &l, const char **kernel) {
    const bool m24 = kh_k_uses_mul24(l.k);
#if KH_TESTING
    if (l.legacy) {
        if (kernel) *kernel = "part1_scatter_chunked_kernel";
        if (m24) launch_legacy<KH_MUL_24>(l);
        else launch_legacy<KH_MUL_32>(l);
        return;
    }
#endif
    if (kernel) *kernel = "part1_bins64_kernel";
    if (!l.generic_k && l.g.shard_shift == 0 && l.g.p1_bits == 10 && l.k >= 22) {
        switch (l.k) {
        case 22: launch_bins<KH_MUL_24, 22>(l); return;
        case 23: launch_bins<KH_MUL_24, 23>(l); return;
        case 24: launch_bins<KH_MUL_24, 24>(l); return;
        case 25: launch_bins<KH_MUL_32, 25>(l); return;
        case 26: launch_bins<KH_MUL_32, 26>(l); return;
        case 27: launch_bins<KH_MUL_32, 27>(l); return;
        case 28: launch_bins<KH_MUL_32, 28>(l); return;
        case 29: launch_bins<KH_MUL_32, 29>(l); return;
        case 30: launch_bins<KH_MUL_32, 30>(l); return;
        case 31: launch_bins<KH_MUL_32, 31>(l); return;
        default: launch_bins<KH_MUL_32, 32>(l); return;
        }
    }
    if (m24) launch_bins<KH_MUL_24, 0>(l);
    else launch_bins<KH_MUL_32, 0>(l);
}

}  // namespace kh
