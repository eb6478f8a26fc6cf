// level1_64.hip -- level 1 with 8-byte payloads (k >= 22, or KMERHIP_PAYLOAD=64): part1_bins64_kernel, one instance per
// k = 22..32 with the written-out window (window.hip.h), the C++ window for every other case.
#define KH_HELPERS_ONLY 1
#include "level1_api.h"
#include "level1.hip.h"

namespace kh {

namespace {
#define KH_L1_ARGS l.abase, l.qbase, l.qaligned, l.vbeg, l.vend, l.wlo, l.tile0, l.ntiles, l.tiles_per_block, l.k, l.thr, l.g, \
                   (u64 *)l.pool, l.chunk_part, l.fill8, l.pool_next, l.pool_chunks, l.ctr
// HOW OFTEN THE BINS ARE FLUSHED (round 6).  A bin holds 16 payloads, a flush keeps up to 7 back, so ~9 may arrive per partition
// between two flushes before the slow path's extra barrier and re-roll are taken; a flush itself is a barrier pair and ~80
// instructions in every one of the sixteen waves, whatever it finds.  Rounds 3-5 flushed every 2 windows per lane, every 4 with -Q.
// Measured on S100M (level 1, ms), windows per lane between flushes 2 / 4 / 8 / 16:  k = 31 unmasked 48.6 / 41.0 / 47.2 / --;
// k = 25 unmasked 49.5 / 42.6 / 48.9 / --;  k = 31 -Q 20 (0.44 of the windows survive) -- / 30.3 / 25.8 / 31.2;  k = 22 -Q 20 (0.6)
// -- / 31.0 / 30.2 / 36.6: best where windows x survival is 3.5-4 arrivals per partition and flush.  So: every 8 windows where
// less than 0.55 of the windows are expected to survive (the host's sample: RangeArgs::survive), else every 4.
#ifndef KH_L1_FW_LOW
#define KH_L1_FW_LOW 8   // (A/B builds)
#endif
#ifndef KH_L1_FW_HIGH
#define KH_L1_FW_HIGH 4
#endif
template <int MODE, int KW>
void launch_bins(const L1Launch &l) {
    const bool low = l.survive < 0.55;
    if (l.use_qual && low) hipLaunchKernelGGL((part1_bins64_kernel<true, MODE, KW, KH_L1_FW_LOW>), dim3(l.grid), dim3(PART_NT), 0, l.stream, KH_L1_ARGS);
    else if (l.use_qual) hipLaunchKernelGGL((part1_bins64_kernel<true, MODE, KW, KH_L1_FW_HIGH>), dim3(l.grid), dim3(PART_NT), 0, l.stream, KH_L1_ARGS);
    else if (low) hipLaunchKernelGGL((part1_bins64_kernel<false, MODE, KW, KH_L1_FW_LOW>), dim3(l.grid), dim3(PART_NT), 0, l.stream, KH_L1_ARGS);
    else hipLaunchKernelGGL((part1_bins64_kernel<false, MODE, KW, KH_L1_FW_HIGH>), dim3(l.grid), dim3(PART_NT), 0, l.stream, KH_L1_ARGS);
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
