// level1_32.hip -- level 1 with 4-byte payloads (k <= 21 at >= 1024 table regions): part1_bins_kernel, one instance per
// k = 11..21 with the written-out window (window.hip.h), the C++ window for every other geometry.
#define KH_HELPERS_ONLY 1
#include "level1_api.h"
#include "level1.hip.h"

namespace kh {

namespace {
#define KH_L1_ARGS l.abase, l.qbase, l.qaligned, l.vbeg, l.vend, l.wlo, l.tile0, l.ntiles, l.tiles_per_block, l.k, l.thr, l.g, \
                   (uint32_t *)l.pool, l.chunk_part, l.fill8, l.pool_next, l.pool_chunks, l.ctr
template <int KW>
void launch_written(const L1Launch &l) {
    constexpr int MODE = (KW >= 16 && KW <= 24) ? KH_MUL_24 : KH_MUL_32;
    if (l.use_qual) hipLaunchKernelGGL((part1_bins_kernel<true, MODE, true, KW>), dim3(l.grid), dim3(PART_NT), 0, l.stream, KH_L1_ARGS);
    else hipLaunchKernelGGL((part1_bins_kernel<false, MODE, true, KW>), dim3(l.grid), dim3(PART_NT), 0, l.stream, KH_L1_ARGS);
}
template <int MODE, bool FAST>
void launch_cpp(const L1Launch &l) {
    if (l.use_qual) hipLaunchKernelGGL((part1_bins_kernel<true, MODE, FAST, 0>), dim3(l.grid), dim3(PART_NT), 0, l.stream, KH_L1_ARGS);
    else hipLaunchKernelGGL((part1_bins_kernel<false, MODE, FAST, 0>), dim3(l.grid), dim3(PART_NT), 0, l.stream, KH_L1_ARGS);
}
#if KH_TESTING  // round 1's tile-sorting kernel: reachable through a test-build switch only, compiled into the test build only
template <int MODE, bool FAST>
void launch_legacy(const L1Launch &l) {
    if (l.use_qual) hipLaunchKernelGGL((part1_scatter_chunked_kernel<true, MODE, FAST, uint32_t, 0>), dim3(l.grid), dim3(PART_NT), 0, l.stream, KH_L1_ARGS);
    else hipLaunchKernelGGL((part1_scatter_chunked_kernel<false, MODE, FAST, uint32_t, 0>), dim3(l.grid), dim3(PART_NT), 0, l.stream, KH_L1_ARGS);
}
#endif
}  // namespace

void launch_level1_32(const L1Launch &l, const char **kernel) {
    const bool m24 = kh_k_uses_mul24(l.k);  // the Feistel multiplier is a compile-time choice in the hot kernels
    const bool fast = p1_fast_ok(l.g);
#if KH_TESTING
    if (l.legacy) {
        if (kernel) *kernel = "part1_scatter_chunked_kernel";
        if (m24 && fast) launch_legacy<KH_MUL_24, true>(l);
        else if (m24) launch_legacy<KH_MUL_24, false>(l);
        else if (fast) launch_legacy<KH_MUL_32, true>(l);
        else launch_legacy<KH_MUL_32, false>(l);
        return;
    }
#endif
    if (kernel) *kernel = "part1_bins_kernel";
    // the written-out window: 1024 partitions, no shard shift (what every table of more than 1024 regions gets)
    if (!l.generic_k && fast && l.g.p1_bits == 10 && l.k >= 11 && l.k <= 21) {
        switch (l.k) {
        case 11: launch_written<11>(l); return;
        case 12: launch_written<12>(l); return;
        case 13: launch_written<13>(l); return;
        case 14: launch_written<14>(l); return;
        case 15: launch_written<15>(l); return;
        case 16: launch_written<16>(l); return;
        case 17: launch_written<17>(l); return;
        case 18: launch_written<18>(l); return;
        case 19: launch_written<19>(l); return;
        case 20: launch_written<20>(l); return;
        default: launch_written<21>(l); return;
        }
    }
    if (m24 && fast) launch_cpp<KH_MUL_24, true>(l);
    else if (m24) launch_cpp<KH_MUL_24, false>(l);
    else if (fast) launch_cpp<KH_MUL_32, true>(l);
    else launch_cpp<KH_MUL_32, false>(l);
}

}  // namespace kh
