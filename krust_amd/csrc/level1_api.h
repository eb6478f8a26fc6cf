// level1_api.h -- how kmerhip.hip reaches level 1 of the partitioned path.  The level-1 kernels are compiled in
// translation units of their own (level1_32.hip: 4-byte payloads, level1_64.hip: 8-byte payloads) -- one kernel per k
// for the written-out window (window.hip.h) is 44 instantiations of a 1024-lane kernel, minutes of compile time that
// run beside the main file's instead of after it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "part_common.hip.h"

#ifndef KH_TESTING
#define KH_TESTING 0  // (ctx.hip.h: the test build compiles kernel variants, geometry switches and failure injection in)
#endif

namespace kh {

struct L1Launch {
    hipStream_t stream;
    unsigned grid;            // workgroups (PART_G1)
    const uint8_t *abase, *qbase;
    int qaligned;
    bool use_qual;
    u64 vbeg, vend, wlo, tile0, ntiles;
    uint32_t tiles_per_block, k, thr;
    PartGeom g;
    void *pool;               // payload pool (uint32_t or u64)
    uint16_t *chunk_part;
    uint8_t *fill8;
    u64 *pool_next;
    u64 pool_chunks;
    Counters *ctr;
    double survive;           // share of the windows expected to survive masking (1 = unknown / all): level1_64.hip's flush cadence
    bool generic_k;           // KMERHIP_GENERIC_K=1: the C++ window even where a written-out one exists (A/B)
    bool legacy;              // KMERHIP_P1_BINS=0: the tile-sorting kernel of round 1 (A/B)
};

// Enqueues the level-1 kernel for 4-byte / 8-byte payloads on l.stream.  *kernel (optional) receives the name of what ran.
void launch_level1_32(const L1Launch &l, const char **kernel);
void launch_level1_64(const L1Launch &l, const char **kernel);

}  // namespace kh
