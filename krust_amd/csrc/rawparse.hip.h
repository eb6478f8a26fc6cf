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
// Round 5: the FASTA passes no longer take line starts (newline counts, their scan, LS[], one header flag per line: 2.9 ms of
// kernels for a 3.1 GB text) nor walk their sixteen bytes one at a time (5.0 + 3.6 ms: with 64 lanes on a 16-wide SIMD a
// wave instruction is four cycles, and ~40 instructions per byte step made both passes ALU-bound at a tenth of the HBM rate).
// Whether a byte lies in a header line is a property of the last line end in front of it, so
//   1. fasta_line_state_kernel: st[u] per 1 KiB unit (one wave's 64 x 16 bytes): 0 = no line end in it, 1 = the line behind
//      its last line end is a record line, 2 = a header line;
//   2. both compaction passes: a wave's state at its first byte = the nearest non-zero st[] in front of it (one coalesced
//      64-byte look-back in all but pathological texts; a 100 Mbp single-line record costs the waves inside it ~200 steps),
//      the lanes' states from two ballots, the bytes' states from ONE ADD: with p = "state carries over to byte j" (no line
//      start at j) and g = "a header line starts at j", the carry chain of (p|g) + g + state_in is the header state of every
//      byte; kept bytes, blank-before-line-end and bare-CR errors are 16-bit mask expressions of the '\n', '\r', '>', ' ',
//      TAB byte masks.

// bit j: byte j of v equals c
__device__ __forceinline__ uint32_t eq_mask16(const uint4 &v, uint32_t c) {
    // swar_eq_bytes leaves 0x80 per matching byte; the multiply gathers bits 0, 8, 16, 24 of (m >> 7) into bits 21..24
    auto nib = [c](uint32_t w) { return (((swar_eq_bytes(w, c) >> 7) * 0x00204081u) >> 21) & 15u; };
    return nib(v.x) | nib(v.y) << 4 | nib(v.z) << 8 | nib(v.w) << 12;
}

constexpr int FASTA_UNIT = 64 * 16;  // bytes one wave takes per step: the granule of st[]

// the byte behind this lane's sixteen ('\n' past the end of the text)
__device__ __forceinline__ uint32_t byte_after16(const uint8_t *__restrict__ raw, u64 p0, u64 n, const uint4 &v) {
    uint32_t nx = (uint32_t)__shfl_down((int)(v.x & 255u), 1, 64);
    if ((threadIdx.x & 63u) == 63u) nx = (p0 + 16 < n) ? raw[p0 + 16] : 0u;
    return (p0 + 16 < n) ? nx : (uint32_t)'\n';
}

// 0: no line end among this lane's bytes; else the state behind its last one (1 record line, 2 header)
__device__ __forceinline__ uint32_t lane_exit_state(uint32_t NL, uint32_t GT, uint32_t after) {
    if (!NL) return 0u;
    const uint32_t k = 32u - (uint32_t)__builtin_clz(NL);  // index of the byte behind the last '\n': 1..16
    const bool gt = k < 16 ? ((GT >> k) & 1u) != 0 : after == '>';  // (bytes past the text's end read as 0 / '\n': never '>')
    return gt ? 2u : 1u;
}

KH_GLOBAL __launch_bounds__(BLOCK) void fasta_line_state_kernel(const uint8_t *__restrict__ raw, u64 n, u64 nunits,
                                                                 uint8_t *__restrict__ st) {
    const u64 nw = (u64)gridDim.x * (BLOCK / 64);
    for (u64 u = (u64)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); u < nunits; u += nw) {
        const u64 p0 = u * FASTA_UNIT + (u64)(threadIdx.x & 63u) * 16;
        const uint4 v = load16_guard(raw, p0, n);
        const uint32_t after = byte_after16(raw, p0, n, v);
        const uint32_t my = lane_exit_state(eq_mask16(v, '\n'), eq_mask16(v, '>'), after);
        const u64 md = __ballot(my != 0), mh = __ballot(my == 2u);
        if ((threadIdx.x & 63u) == 0) st[u] = md ? (((mh >> (63 - __builtin_clzll(md))) & 1) ? 2u : 1u) : 0u;
    }
}

// the state at the first byte of unit u: that of the nearest unit in front of it that holds a line end.  x = st[u - 1 - lane]
// (0 for lane >= u) is loaded by the caller, ahead of time; first_state = that of the text's first line.
__device__ __forceinline__ uint32_t fasta_unit_entry_state(const uint8_t *__restrict__ st, u64 u, uint32_t x, uint32_t first_state) {
    const uint32_t lane = threadIdx.x & 63u;
    u64 hi = u;  // units [hi - 64, hi) are in x
    for (;;) {
        const u64 m = __ballot(x != 0);
        if (m) return (uint32_t)__shfl((int)x, __builtin_ctzll(m), 64);
        if (hi <= 64) return first_state;
        hi -= 64;
        x = lane < hi ? st[hi - 1 - lane] : 0u;
    }
}

// PASS 0: tile_keep[t] = bytes kept in tile t.   PASS 1: write them at out[tile_out[t] + ...].
// Kept: inside a record every byte but CR / LF; of a header line only its '\n' (as the separator).
// PASS 0 also raises err for a blank (' ' / TAB) right before a line end inside a record: the line
// parsers strip those before joining wrapped lines (rust-bio trims line ends), the byte-wise rule
// here would keep them as a separator in the middle of the record; and for a CR that is not directly
// followed by the line's '\n' (the parsers keep it as an invalid base, this rule would drop it:
// a CR is a line-end byte only right before '\n' or at the very end of the text -- rust-bio's trim_end()
// and the host line parser strip nothing else; dropping a bare CR here would join "AC\rGT" to ACGT).
template <int PASS>
__global__ __launch_bounds__(BLOCK) void fasta_compact_kernel(const uint8_t *__restrict__ raw, u64 n, u64 ntiles,
                                                              const uint8_t *__restrict__ st, uint32_t *__restrict__ tile_keep,
                                                              const u64 *__restrict__ tile_out, uint8_t *__restrict__ out,
                                                              uint32_t *__restrict__ err) {
    __shared__ uint32_t s_w[4];
    // PASS 1: the tile's kept bytes are gathered in LDS -- at the offset their destination has inside its 16-byte word -- and
    // leave as whole aligned 16-byte stores (round 5; one byte per store instruction took 4.5 ms for a 3.1 GB text)
    __shared__ __attribute__((aligned(16))) uint8_t s_out[PASS == 1 ? RAW_TILE + 32 : 16];
    static_assert(RAW_TILE == 4 * FASTA_UNIT, "a workgroup's four waves take one unit each");
    u64 t = blockIdx.x;
    if (t >= ntiles) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t first_state = raw[0] == '>' ? 2u : 1u;
    uint4 v = load16_guard(raw, t * RAW_TILE + (u64)threadIdx.x * 16, n);
    uint32_t sx = lane < 4 * t + wave ? st[4 * t + wave - 1 - lane] : 0u;
    for (;;) {
        const u64 p0 = t * RAW_TILE + (u64)threadIdx.x * 16;
        const u64 tn = t + gridDim.x;
        uint4 vn = make_uint4(0, 0, 0, 0);
        uint32_t sxn = 0;
        if (tn < ntiles) {  // the next tile's loads are in flight while this one is worked on
            vn = load16_guard(raw, tn * RAW_TILE + (u64)threadIdx.x * 16, n);
            sxn = lane < 4 * tn + wave ? st[4 * tn + wave - 1 - lane] : 0u;
        }
        const uint32_t after = byte_after16(raw, p0, n, v);
        const uint32_t NL = eq_mask16(v, '\n'), CR = eq_mask16(v, '\r'), GT = eq_mask16(v, '>');
        // the state at this lane's first byte: behind the last line end of an earlier lane of the wave, else the wave's
        const uint32_t my = lane_exit_state(NL, GT, after);
        const u64 md = __ballot(my != 0), mh = __ballot(my == 2u);
        const u64 below = md & ((1ull << lane) - 1);
        const uint32_t wave_in = fasta_unit_entry_state(st, 4 * t + wave, sx, first_state);
        const uint32_t in = below ? (uint32_t)((mh >> (63 - __builtin_clzll(below))) & 1) : (wave_in == 2u ? 1u : 0u);
        // H bit j: byte j lies in a header line.  A line starts at j where byte j - 1 is '\n' (byte 0's line start belongs
        // to `in`); it is a header iff byte j is '>'.  H[j] = g[j] | (p[j] & H[j-1]) is the carry out of bit j of the sum.
        const uint32_t LS = (NL << 1) & 0xFFFFu, g = LS & GT, p = ~LS & 0xFFFFu;
        const uint32_t H = ((((p | g) + g + in) ^ p) >> 1) & 0xFFFFu;
        const u64 left = p0 < n ? n - p0 : 0;  // bytes of the text from p0 on
        const uint32_t VAL = left >= 16 ? 0xFFFFu : (1u << (uint32_t)left) - 1u;
        const uint32_t keep = VAL & ((H & NL) | (~H & ~(NL | CR)));  // bit j: byte j is kept
        bool bad = false;
        if (PASS == 0) {
            const uint32_t BL = eq_mask16(v, ' ') | eq_mask16(v, '\t');
            const uint32_t nxNL = (NL >> 1) | (after == '\n' ? 0x8000u : 0u), nxCR = (CR >> 1) | (after == '\r' ? 0x8000u : 0u);
            const uint32_t END = left > 16 ? 0u : ~(VAL >> 1) & 0xFFFFu;  // byte j is the text's last, or behind it
            bad = ((BL & ~H & (nxNL | nxCR | END)) | (CR & ~H & VAL & ~(nxNL | END))) != 0;
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
        if (tn >= ntiles) break;
        t = tn;
        v = vn;
        sx = sxn;
    }
}

}  // namespace kh
