// rawparse.hip.h -- device-side FASTA / FASTQ record scanning.
//
// The step in front of the hot path: the reference parses records on the host
// (src/reader.rs:58-79, src/streaming.rs:858-893: rust-bio readers, one allocation per record) and
// that, not counting, bounds its end-to-end time.  Here the raw file text goes to HBM as it is and
// these kernels turn it into the flat layout the count kernels take (records separated by a byte
// outside ACGTacgt, optional parallel quality buffer):
//   1. count newlines per 4 KiB tile, scan -> line index of every byte
//   2. scatter the line starts LS[]
//   3. FASTQ (4-line records, validated: '@' / '+' markers, |seq| == |qual|): sequence lines are
//      copied, everything else becomes '\n'; qualities are gathered to the positions of their bases
//      FASTA: header lines collapse to one separator, line breaks inside a record are REMOVED
//      (stream compaction: k-mers span the line breaks of a wrapped record)
// Anything these kernels do not accept (wrapped FASTQ, missing markers) is reported, and the host
// falls back to its line parser -- never a silent difference.
#pragma once
#include "kernels.hip.h"

namespace kh {

constexpr int RAW_TILE = BLOCK * 16;  // 4096 bytes per workgroup iteration

// 0x80 in every byte of w equal to c
__device__ __forceinline__ uint32_t swar_eq_bytes(uint32_t w, uint32_t c) { return swar_zero_bytes(w ^ (c * 0x01010101u)); }

__device__ __forceinline__ uint32_t count_nl16(const uint4 &v) {
    return (uint32_t)(__builtin_popcount(swar_eq_bytes(v.x, '\n')) + __builtin_popcount(swar_eq_bytes(v.y, '\n')) +
                      __builtin_popcount(swar_eq_bytes(v.z, '\n')) + __builtin_popcount(swar_eq_bytes(v.w, '\n')));
}

// exclusive prefix of `v` over the 256 lanes of a workgroup; *total = sum.  One barrier pair.
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t *s_w, uint32_t *total) {
    const int tid = threadIdx.x;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t n = __shfl_up(incl, off, 64);
        if ((tid & 63) >= off) incl += n;
    }
    __syncthreads();  // s_w may still be read from a previous call
    if ((tid & 63) == 63) s_w[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += s_w[w];
    *total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    return base + incl - v;
}

__device__ __forceinline__ uint4 load16_guard(const uint8_t *__restrict__ raw, u64 pos, u64 n) {
    // raw is 16-byte aligned; the last, partial group is read byte by byte (a caller's device buffer
    // need not be padded) and reads as '\0' past n
    if (pos >= n) return make_uint4(0, 0, 0, 0);
    if (pos + 16 <= n) return *reinterpret_cast<const uint4 *>(raw + pos);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int j = 0; j < 16 && pos + j < n; ++j) w[j >> 2] |= (uint32_t)raw[pos + j] << (8 * (j & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// tile_nl[t] = number of '\n' in tile t
KH_GLOBAL __launch_bounds__(BLOCK) void raw_nl_count_kernel(const uint8_t *__restrict__ raw, u64 n, u64 ntiles,
                                                             uint32_t *__restrict__ tile_nl) {
    __shared__ uint32_t s_w[4];
    for (u64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint4 v = load16_guard(raw, t * RAW_TILE + (u64)threadIdx.x * 16, n);
        uint32_t total;
        (void)block_exclusive_scan_256(count_nl16(v), s_w, &total);
        if (threadIdx.x == 0) tile_nl[t] = total;
    }
}

// LS[i + 1] = position after the i-th newline (LS[0] = 0 is written by the host)
KH_GLOBAL __launch_bounds__(BLOCK) void raw_line_starts_kernel(const uint8_t *__restrict__ raw, u64 n, u64 ntiles,
                                                                const u64 *__restrict__ tile_base, u64 *__restrict__ LS) {
    __shared__ uint32_t s_w[4];
    for (u64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const u64 p0 = t * RAW_TILE + (u64)threadIdx.x * 16;
        const uint4 v = load16_guard(raw, p0, n);
        uint32_t total;
        u64 idx = tile_base[t] + block_exclusive_scan_256(count_nl16(v), s_w, &total);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (byte_of(v, j) == '\n') LS[++idx] = p0 + j + 1;
    }
}

// ---- FASTQ ---------------------------------------------------------------------------------------
// One lane per record r (lines 4r .. 4r+3).  err[0] |= 1 on a layout these kernels do not take.
KH_GLOBAL __launch_bounds__(BLOCK) void fastq_validate_kernel(const uint8_t *__restrict__ raw, const u64 *__restrict__ LS,
                                                               u64 nrecords, uint32_t *__restrict__ err) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    bool bad = false;
    for (u64 r = (u64)blockIdx.x * BLOCK + threadIdx.x; r < nrecords; r += stride) {
        const u64 a = LS[4 * r], b = LS[4 * r + 1], c = LS[4 * r + 2], d = LS[4 * r + 3], e = LS[4 * r + 4];
        u64 seq_len = c - b - 1, qual_len = e - d - 1;  // without the '\n'
        if (seq_len && raw[c - 2] == '\r') --seq_len;
        if (qual_len && raw[e - 2] == '\r') --qual_len;
        bad |= raw[a] != '@' || raw[c] != '+' || seq_len != qual_len;
    }
    if (kh_any(bad) && lane_id() == 0) atomicOr(err, 1u);
}

// bases_out[pos] = raw[pos] on sequence lines (line index % 4 == 1, excluding CR / LF), '\n' elsewhere;
// qual_out[pos] (optional) = the quality byte of that base.
template <bool QUAL>
__global__ __launch_bounds__(BLOCK) void fastq_mark_kernel(const uint8_t *__restrict__ raw, u64 n, u64 ntiles,
                                                           const u64 *__restrict__ tile_base, const u64 *__restrict__ LS,
                                                           uint8_t *__restrict__ bases_out, uint8_t *__restrict__ qual_out) {
    __shared__ uint32_t s_w[4];
    for (u64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const u64 p0 = t * RAW_TILE + (u64)threadIdx.x * 16;
        const uint4 v = load16_guard(raw, p0, n);
        uint32_t total;
        u64 line = tile_base[t] + block_exclusive_scan_256(count_nl16(v), s_w, &total);
        uint32_t ob[4] = {0, 0, 0, 0}, oq[4] = {0, 0, 0, 0};
        u64 delta = 0;        // qual position - base position, for the current sequence line
        bool have_delta = false;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t b = byte_of(v, j);
            uint32_t outb = '\n', outq = '\n';
            if ((line & 3) == 1 && b != '\n' && b != '\r' && p0 + j < n) {
                outb = b;
                if (QUAL) {
                    if (!have_delta) {
                        delta = LS[line + 2] - LS[line];
                        have_delta = true;
                    }
                    outq = raw[p0 + j + delta];
                }
            }
            ob[j >> 2] |= outb << (8 * (j & 3));
            oq[j >> 2] |= outq << (8 * (j & 3));
            if (b == '\n') {
                ++line;
                have_delta = false;
            }
        }
        if (p0 < n) {  // output buffers are padded to a multiple of 16
            *reinterpret_cast<uint4 *>(bases_out + p0) = make_uint4(ob[0], ob[1], ob[2], ob[3]);
            if (QUAL) *reinterpret_cast<uint4 *>(qual_out + p0) = make_uint4(oq[0], oq[1], oq[2], oq[3]);
        }
    }
}

// ---- FASTA ---------------------------------------------------------------------------------------
// hdr[L] = line L starts with '>'
KH_GLOBAL __launch_bounds__(BLOCK) void fasta_headers_kernel(const uint8_t *__restrict__ raw, u64 n, const u64 *__restrict__ LS,
                                                              u64 nlines, uint8_t *__restrict__ hdr) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    for (u64 L = (u64)blockIdx.x * BLOCK + threadIdx.x; L < nlines; L += stride) {
        const u64 s = LS[L];
        hdr[L] = (s < n && raw[s] == '>') ? 1 : 0;
    }
}

// keep(byte): inside a record every byte but CR / LF; of a header line only its '\n' (as the separator)
__device__ __forceinline__ bool fasta_keep(uint32_t b, bool header) {
    return header ? (b == '\n') : (b != '\n' && b != '\r');
}

// PASS 0: tile_keep[t] = bytes kept in tile t.   PASS 1: write them at out[tile_out[t] + ...].
// PASS 0 also raises err for a blank (' ' / TAB) right before a line end inside a record: the line
// parsers strip those before joining wrapped lines (rust-bio trims line ends), the byte-wise rule
// here would keep them as a separator in the middle of the record; and for a CR that is not directly
// followed by the line's '\n' (the parsers keep it as an invalid base, this rule would drop it).
template <int PASS>
__global__ __launch_bounds__(BLOCK) void fasta_compact_kernel(const uint8_t *__restrict__ raw, u64 n, u64 ntiles,
                                                              const u64 *__restrict__ tile_base, const uint8_t *__restrict__ hdr,
                                                              uint32_t *__restrict__ tile_keep, const u64 *__restrict__ tile_out,
                                                              uint8_t *__restrict__ out, uint32_t *__restrict__ err) {
    __shared__ uint32_t s_w[4];
    // PASS 1: the tile's kept bytes are gathered in LDS -- at the offset their destination has inside its 16-byte word -- and
    // leave as whole aligned 16-byte stores (round 5; one byte per store instruction took 4.5 ms for a 3.1 GB text)
    __shared__ __attribute__((aligned(16))) uint8_t s_out[PASS == 1 ? RAW_TILE + 32 : 16];
    for (u64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const u64 p0 = t * RAW_TILE + (u64)threadIdx.x * 16;
        const uint4 v = load16_guard(raw, p0, n);
        uint32_t total;
        const u64 line0 = tile_base[t] + block_exclusive_scan_256(count_nl16(v), s_w, &total);
        bool header = p0 < n ? hdr[line0] != 0 : false;  // (the line this lane's first byte is in)
        uint32_t keep = 0;  // bit j: byte j is kept
        bool bad = false;
        const uint32_t after = (p0 + 16 < n) ? raw[p0 + 16] : '\n';  // the byte after this lane's 16
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t b = byte_of(v, j);
            if (p0 + j < n && fasta_keep(b, header)) keep |= 1u << j;
            if (PASS == 0 && !header && (b == ' ' || b == '\t')) {
                const uint32_t nx = j < 15 ? byte_of(v, (j + 1) & 15) : after;
                bad |= (nx == '\n' || nx == '\r' || p0 + j + 1 >= n);
            }
            // A CR is a line-end byte only right before '\n' (or at the very end of the text): rust-bio's
            // trim_end() and the host line parser strip nothing else.  A bare CR in the middle of a line
            // stays an invalid base there and breaks windows; dropping it here would join "AC\rGT" to ACGT.
            if (PASS == 0 && !header && b == '\r' && p0 + j < n) {
                const uint32_t nx = j < 15 ? byte_of(v, (j + 1) & 15) : after;
                bad |= !(nx == '\n' || p0 + j + 1 >= n);
            }
            if (b == '\n') {
                // the next line is a header iff it starts with '>': hdr[] says the same (fasta_headers_kernel), but a load
                // from it HERE is a dependent global load in the middle of a sixteen-step loop -- and with 61-byte lines some
                // lane of the wave is at a line end in nearly every step (round 4: 6.1 ms per pass over a 3.1 GB text)
                const uint32_t nx = j < 15 ? byte_of(v, (j + 1) & 15) : after;
                header = (p0 + j + 1 < n) && nx == '>';
            }
        }
        uint32_t ktotal;
        const uint32_t kpre = block_exclusive_scan_256((uint32_t)__builtin_popcount(keep), s_w, &ktotal);
        if (PASS == 0) {
            if (threadIdx.x == 0) tile_keep[t] = ktotal;
            if (kh_any(bad) && lane_id() == 0) atomicOr(err, 2u);
        } else {
            const u64 obase = tile_out[t];
            const uint32_t a = (uint32_t)(reinterpret_cast<uintptr_t>(out + obase) & 15u);  // where the tile's first byte sits in its word
            uint32_t o = a + kpre;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (keep & (1u << j)) s_out[o++] = (uint8_t)byte_of(v, j);
            __syncthreads();
            const uint32_t end = a + ktotal;          // the tile's bytes are s_out[a, end)
            uint8_t *const gb = out + obase - a;      // 16-byte aligned
            for (uint32_t lo = threadIdx.x * 16u; lo < end; lo += BLOCK * 16u) {
                if (lo >= a && lo + 16u <= end) {
                    *reinterpret_cast<uint4 *>(gb + lo) = *reinterpret_cast<const uint4 *>(s_out + lo);
                } else {  // the first and the last word of the tile are shared with its neighbours: byte by byte
                    const uint32_t b0 = lo < a ? a : lo, b1 = lo + 16u < end ? lo + 16u : end;
                    for (uint32_t b = b0; b < b1; ++b) gb[b] = s_out[b];
                }
            }
            __syncthreads();  // (the next tile writes s_out)
        }
    }
}

}  // namespace kh
