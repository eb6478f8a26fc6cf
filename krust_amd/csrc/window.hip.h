// window.hip.h -- the level-1 window written out for EVERY k from 11 to 32 (gfx950).
//
// Round 2 wrote the per-window code of level 1 out by hand for k = 21 only (105 instead of ~215 issue cycles per
// window, 33.0 -> 27.0 ms on S100M); every other k took the C++ Roller window.  The reference serves k = 1..32
// uniformly (src/kmer.rs:100-110), so a headline kernel tuned to the benchmark's k alone is benchmark-shaped.
// Everything in that window is a compile-time function of (k, J): this header generates it.
//
// What a lane holds after staging a tile: its own 16 bases (J = 0..15) and the 32 in front of them as three 2-bit
// code words w2:w1:w0 (first base in the top bits of w2, base J of the lane's own chunk at bits 31-2J..30-2J of w0),
// and their reverse complements c2:c1:c0 = 2-bit groups reversed and inverted (c0 = rev2_complement(w2) ... ), so
// that in the 96-bit string C = c2:c1:c0 the complement of base m (m = 0 the oldest of the 48) sits at bits 2m..2m+1.
//   forward k-mer of the window ENDING at own base J:   (W >> F)  & mask(2k),   F  = 2 (15 - J)          in 0..30
//   its reverse complement:                              (C >> SR) & mask(2k),   SR = 2 (33 + J - k)      in 2..74
// Both are at most two words; which instructions extract them (v_alignbit, v_bfe, v_and, v_mov, none) depends only
// on where the field lies relative to the word boundaries -- WinPlan says, win_fields() does it with the builtins
// that ARE those instructions on the device and with plain C++ on the host, so that tests/test_window_plan.py can
// check every (k, J) against kh_revcomp / the packing of src/kmer.rs:467-471 on the CPU, without a GPU.
// The rest of the window -- canonical choice into an SGPR pair, the four Feistel rounds, the addresses the LDS
// instructions need -- is one asm statement per window (win_hash32 / win_hash64); its immediates come from the
// same constants.  kmer_bits.h (kh_hash_n) is the definition it must equal; the parity tests run every k.
#pragma once
#include <stdint.h>

#include "kmer_bits.h"

namespace kh {

// ---- the three device instructions the extraction is made of, with host twins --------------------------------------
KH_HD uint32_t w_alignbit(uint32_t hi, uint32_t lo, uint32_t s) {  // (hi:lo >> s), low 32 bits; s in 0..31
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> s);
#endif
}
KH_HD uint32_t w_bfe(uint32_t x, uint32_t off, uint32_t width) {  // width in 1..32, off + width <= 32
#if defined(__HIP_DEVICE_COMPILE__)
    return width == 32 ? x : __builtin_amdgcn_ubfe(x, off, width);
#else
    return width == 32 ? x : ((x >> off) & ((1u << width) - 1u));
#endif
}

template <int K, int J>
struct WinPlan {
    static_assert(K >= 11 && K <= 32 && J >= 0 && J <= 15, "the written-out window covers k = 11..32");
    static constexpr int KB = 2 * K;                   // bits of a packed k-mer
    static constexpr int LB = KB > 32 ? 32 : KB;       // ... in the low word
    static constexpr int HB = KB > 32 ? KB - 32 : 0;   // ... in the high word
    static constexpr int F = 2 * (15 - J);             // forward field: bit offset in w2:w1:w0
    static constexpr int SR = 2 * (33 + J - K);        // reverse complement: bit offset in c2:c1:c0
    static constexpr int B = SR / 32, S = SR % 32;     // ... as word index and shift
    static_assert(SR >= 0 && SR + KB <= 96 && F + KB <= 96, "field outside the lane's 48 bases");
};

// field [off, off + KB) of the 96-bit string x2:x1:x0 as (lo, hi), hi = 0 for k <= 16
template <int KB, int OFF>
KH_HD void win_field(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t &lo, uint32_t &hi) {
    constexpr int B = OFF / 32, S = OFF % 32;
    constexpr int LB = KB > 32 ? 32 : KB, HB = KB > 32 ? KB - 32 : 0;
    const uint32_t a = B == 0 ? x0 : B == 1 ? x1 : x2;          // the word the field starts in
    const uint32_t b = B == 0 ? x1 : B == 1 ? x2 : 0u;          // the next one (never needed beyond x2: static_assert above)
    const uint32_t c = B == 0 ? x2 : 0u;
    if constexpr (S + LB <= 32) lo = w_bfe(a, S, LB);                       // inside one word: one v_bfe (or nothing)
    else if constexpr (LB == 32) lo = w_alignbit(b, a, S);                  // a whole word across a boundary: one v_alignbit
    else lo = w_alignbit(b, a, S) & ((1u << LB) - 1u);
    if constexpr (HB == 0) hi = 0u;
    else if constexpr (S == 0) hi = w_bfe(b, 0, HB);
    else if constexpr (S + HB <= 32) hi = w_bfe(b, S, HB);
    else if constexpr (HB == 32) hi = w_alignbit(c, b, S);
    else hi = w_alignbit(c, b, S) & ((1u << HB) - 1u);
}

// forward k-mer and reverse complement of the window ending at own base J
template <int K, int J>
KH_HD void win_fields(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t &flo, uint32_t &fhi,
                      uint32_t &rlo, uint32_t &rhi) {
    using P = WinPlan<K, J>;
    win_field<P::KB, P::F>(w0, w1, w2, flo, fhi);
    win_field<P::KB, P::SR>(c0, c1, c2, rlo, rhi);
}

// 2-bit groups of x reversed and complemented (host twin of partition.hip.h rev2_complement)
KH_HD uint32_t w_rev2_complement(uint32_t x) {
    uint32_t y = ~x;
    y = ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);  // full bit reversal ...
    y = ((y >> 2) & 0x33333333u) | ((y & 0x33333333u) << 2);
    y = ((y >> 4) & 0x0F0F0F0Fu) | ((y & 0x0F0F0F0Fu) << 4);
    y = ((y >> 8) & 0x00FF00FFu) | ((y & 0x00FF00FFu) << 8);
    y = (y >> 16) | (y << 16);
    return ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);  // ... and the bits inside each group swapped back
}

// ---- what the asm below computes, in C++ (host and device): the reference the tests hold it to ---------------------
// level-1 digit (1024 partitions), 32-bit payload (the 2k - 10 hash bits below the digit, left-aligned) and the
// addresses level 1's LDS instructions take: counter word p1 * 4, bin p1 * 128
template <int K>
KH_HD void win_outputs_ref(uint64_t key, uint32_t &p1, uint32_t &pay32) {
    const uint64_t h = kh_hash_n<KH_MUL_AUTO>(key, K);  // 2k bits
    p1 = (uint32_t)(h >> (2 * K - 10));
    pay32 = 2 * K - 10 <= 32 ? (uint32_t)((h << (64 - (2 * K - 10))) >> 32) : 0u;
}

#if defined(__HIPCC__)
// The Feistel rounds on the halves held in two registers (round 6: TWO instructions per round).  A = v122 is the hash's upper half
// LEFT-aligned (bits 32-K .. 31), B = v123 the lower half right-aligned; both stay where they are (no swaps, no copies):
//   rounds 1, 3:  A = (A ^ (B x C)) & top-K-bits      v_mul_u32_u24 / v_mul_lo_u32 + v_bitop3 0x28  ((S0 ^ S1) & S2)
//   rounds 2, 4:  B ^= mulhi(A, C) & low-K-bits       v_mul_hi_u32 + v_bitop3 0x78                  (S0 ^ (S1 & S2))
// (kmer_bits.h: kh_feistel_f / kh_feistel_g -- the top K bits of the product's low word, and bits K .. 2K-1 of the full product:
// mulhi of the left-aligned half is exactly that.)  The & of round 1 also clears what the split leaves below A's K bits (the
// key's next bits: one v_alignbit makes A, no mask).  The two masks come in VGPRs (%[tm], %[kmv]: a v_bitop3 with an SGPR
// operand costs 4.7 cycles instead of 2.9, tools/ubench/valu_rates.hip); 16 <= K <= 24: the 24-bit multiplier with the constant's
// low 24 bits in rounds 1 and 3; otherwise v_mul_lo_u32, which is VOP3 and takes no literal: the constants come in SGPRs.
#define KH_WSTR2(x) #x
#define KH_WSTR(x) KH_WSTR2(x)
#define KH_WIN_RA24(c24) "v_mul_u32_u24 v124, " c24 ", v123\n v_bitop3_b32 v122, v122, v124, %[tm] bitop3:0x28\n"
#define KH_WIN_RA32(fc) "v_mul_lo_u32 v124, v123, " fc "\n v_bitop3_b32 v122, v122, v124, %[tm] bitop3:0x28\n"
#define KH_WIN_RB(fc) "v_mul_hi_u32 v124, v122, " fc "\n v_bitop3_b32 v123, v123, v124, %[kmv] bitop3:0x78\n"
#if KH_ABL_ROUNDS3  /* timing experiment only (wrong hash): three Feistel rounds */
#define KH_WIN_ROUNDS24 KH_WIN_RA24("0x3779b1") KH_WIN_RB("%[f1]") KH_WIN_RA24("0xb2ae3d")
#else
#define KH_WIN_ROUNDS24 KH_WIN_RA24("0x3779b1") KH_WIN_RB("%[f1]") KH_WIN_RA24("0xb2ae3d") KH_WIN_RB("%[f3]")
#endif
#define KH_WIN_ROUNDS32 KH_WIN_RA32("%[f0]") KH_WIN_RB("%[f1]") KH_WIN_RA32("%[f2]") KH_WIN_RB("%[f3]")
// 32-bit payloads (round 4): THREE rounds.  A -- the level-1 digit, the addresses -- is final after the third; the
// fourth, which only changes B, is left to level 2 (part_common.hip.h hash_p1_pay32 / pay32_finish: level 1 is bound by its
// instruction stream, level 2 is not -- or so it seemed).  KH_L1_DEFER_ROUND=1 builds it; the default is the four-round window.
#ifndef KH_L1_DEFER_ROUND
#define KH_L1_DEFER_ROUND 0  // (measured: level 1 -1.0 ms, level 2 +2.0 ms -- see part_common.hip.h)
#endif
#if KH_L1_DEFER_ROUND
#define KH_WIN_ROUNDS24_P32 KH_WIN_RA24("0x3779b1") KH_WIN_RB("%[f1]") KH_WIN_RA24("0xb2ae3d")
#define KH_WIN_ROUNDS32_P32 KH_WIN_RA32("%[f0]") KH_WIN_RB("%[f1]") KH_WIN_RA32("%[f2]")
#else
#define KH_WIN_ROUNDS24_P32 KH_WIN_ROUNDS24
#define KH_WIN_ROUNDS32_P32 KH_WIN_ROUNDS32
#endif
// After the rounds A = v122 (K bits, left-aligned: the level-1 digit on top), B = v123.
// Counter address and bin offset from A, the same shifts for every K: (A >> 20) & 0xffc and (A >> 15) & 0x1ff80.  A window
// without a key takes the lane's waste counter: sign-extended validity bit + v_bitop3.
#define KH_WIN_ADDR                                                               \
    "v_lshrrev_b32 v124, 20, v122\n v_and_b32 v124, 0xffc, v124\n"                \
    "v_bfe_i32 v120, %[good], %[gb], 1\n"                                         \
    "v_bitop3_b32 %[cnta], v120, v124, %[waste] bitop3:0xca\n"                    \
    "v_lshrrev_b32 v124, 15, v122\n v_and_b32 %[binb], 0x1ff80, v124\n"           \
    "v_lshrrev_b32 v124, 3, %[binb]\n v_and_or_b32 %[binb], v124, %[rot], %[binb]\n"  /* | (p1 & 7) << 4: the bin's unit permutation (level1.hip.h (6)) */

// 32-bit payloads, K = 11..21.  In: forward / reverse complement (hi words are zero for K <= 16), the lane's validity
// word and waste counter address.  Out: payload, byte address of the partition's counter (or the waste counter), byte
// offset of the partition's bin.  The two strands come in twice, as 64-bit pairs (for v_cmp_lt_u64) and as their
// halves (for the selects): the same registers, the compiler builds the pair around the halves.
template <int K, int J>
__device__ __forceinline__ void win_hash32(uint32_t flo, uint32_t fhi, uint32_t rlo, uint32_t rhi, uint32_t good, uint32_t waste,
                                           uint32_t rot, uint32_t &pay, uint32_t &cnta, uint32_t &binb) {
    static_assert(K >= 11 && K <= 21, "32-bit payloads: 2k - 10 <= 32");
    constexpr uint32_t KM = (1u << K) - 1u;
    constexpr int PR = 42 - 2 * K;                       // payload = A << 10 | B << PR (A's digit falls off the top)
    constexpr int AS = K >= 17 ? 2 * K - 32 : 32 - 2 * K;  // A = key >> AS (two words) or key << AS (one word): the key's top 32 bits
    const uint32_t f0 = KH_FC0, f1 = KH_FC1, f2 = KH_FC2, f3 = KH_FC3;
    const uint32_t tm = ~0u << (32 - K), kmv = KM;
    const uint64_t f64 = ((uint64_t)fhi << 32) | flo, r64 = ((uint64_t)rhi << 32) | rlo;
#define KH_W32_OPERANDS                                                                                                   \
    : [pay] "=&v"(pay), [cnta] "=&v"(cnta), [binb] "=&v"(binb)                                                            \
    : [flo] "v"(flo), [fhi] "v"(fhi), [rlo] "v"(rlo), [rhi] "v"(rhi), [f] "v"(f64), [r] "v"(r64), [good] "v"(good),      \
      [waste] "v"(waste), [tm] "v"(tm), [kmv] "v"(kmv), [f0] "s"(f0), [f1] "s"(f1), [f2] "s"(f2), [f3] "s"(f3),          \
      [rot] "s"(rot), [km] "n"(KM), [as] "n"(AS), [pr] "n"(PR), [gb] "n"(15 - J)                                         \
    : "v120", "v121", "v122", "v123", "v124", "s98", "s99"
#define KH_W32_PAY "v_lshlrev_b32 v124, %[pr], v123\n v_lshl_or_b32 %[pay], v122, 10, v124\n"
#define KH_W32_PAY0 "v_lshl_or_b32 %[pay], v122, 10, v123\n"   /* K = 21: B needs no shift */
    // canonical = min(forward, reverse complement) as integers (== the reference's lexicographic choice,
    // src/kmer.rs:348-365): k >= 17 compare into an SGPR pair, two selects on it; k <= 16 one v_min_u32
#define KH_W32_CANON64                                              \
    "v_cmp_lt_u64_e64 s[98:99], %[f], %[r]\n"                       \
    "v_cndmask_b32_e64 v120, %[rlo], %[flo], s[98:99]\n"            \
    "v_cndmask_b32_e64 v121, %[rhi], %[fhi], s[98:99]\n"            \
    "v_alignbit_b32 v122, v121, v120, %[as]\n"     /* A (+ the key's next bits below: round 1 clears them) */ \
    "v_and_b32 v123, %[km], v120\n"                /* B */
#define KH_W32_CANON32 \
    "v_min_u32 v120, %[flo], %[rlo]\n v_lshlrev_b32 v122, %[as], v120\n v_and_b32 v123, %[km], v120\n"
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass only needs the declaration)
    if constexpr (K == 21) {
        asm(KH_W32_CANON64 KH_WIN_ROUNDS24_P32 KH_W32_PAY0 KH_WIN_ADDR KH_W32_OPERANDS);
    } else if constexpr (K >= 17) {
        asm(KH_W32_CANON64 KH_WIN_ROUNDS24_P32 KH_W32_PAY KH_WIN_ADDR KH_W32_OPERANDS);
    } else if constexpr (K == 16) {
        asm(KH_W32_CANON32 KH_WIN_ROUNDS24_P32 KH_W32_PAY KH_WIN_ADDR KH_W32_OPERANDS);
    } else {
        asm(KH_W32_CANON32 KH_WIN_ROUNDS32_P32 KH_W32_PAY KH_WIN_ADDR KH_W32_OPERANDS);
    }
#else
    (void)f64; (void)r64; (void)rot; (void)f0; (void)f1; (void)f2; (void)f3; (void)KM; (void)PR; (void)AS; (void)tm; (void)kmv;
    pay = cnta = binb = 0;
#endif
#undef KH_W32_CANON32
#undef KH_W32_CANON64
#undef KH_W32_PAY0
#undef KH_W32_PAY
#undef KH_W32_OPERANDS
}

// 64-bit payloads, K = 22..32.  Out: the payload (klo, khi) -- round 6: the 2k - 10 hash bits below the level-1 digit,
// left-aligned in 64 bits (part_common.hip.h Pay<u64>; rounds 1-5: the canonical key) --, the counter's byte address (or the
// waste counter's), the bin's byte offset (16 payloads of 8 bytes = 128 bytes per bin: the same (L >> (K - 17)) & 0x1ff80).
// With h = L << K | R the payload is h << (74 - 2K): low word R << S, high word L << (42 - K) | R >> (32 - S), S = 74 - 2K.
template <int K, int J>
__device__ __forceinline__ void win_hash64(uint32_t flo, uint32_t fhi, uint32_t rlo, uint32_t rhi, uint32_t good, uint32_t waste,
                                           uint32_t rot, uint32_t &klo, uint32_t &khi, uint32_t &cnta, uint32_t &binb) {
    static_assert(K >= 22 && K <= 32, "64-bit payloads");
    constexpr uint32_t KM = K < 32 ? (1u << (K & 31)) - 1u : 0xFFFFFFFFu;
    constexpr int AS = (2 * K - 32) & 31;                         // A = key >> (2K - 32): the key's top 32 bits (K = 32: its high word)
    constexpr int PS = 74 - 2 * K, PSR = 32 - PS;                 // the payload's shifts (above)
    const uint32_t f0 = KH_FC0, f1 = KH_FC1, f2 = KH_FC2, f3 = KH_FC3;
    const uint32_t tm = K < 32 ? ~0u << ((32 - K) & 31) : ~0u, kmv = KM;
    const uint64_t f64 = ((uint64_t)fhi << 32) | flo, r64 = ((uint64_t)rhi << 32) | rlo;
#define KH_W64_OPERANDS                                                                                                   \
    : [klo] "=&v"(klo), [khi] "=&v"(khi), [cnta] "=&v"(cnta), [binb] "=&v"(binb)                                          \
    : [flo] "v"(flo), [fhi] "v"(fhi), [rlo] "v"(rlo), [rhi] "v"(rhi), [f] "v"(f64), [r] "v"(r64), [good] "v"(good),      \
      [waste] "v"(waste), [tm] "v"(tm), [kmv] "v"(kmv), [f0] "s"(f0), [f1] "s"(f1), [f2] "s"(f2), [f3] "s"(f3),          \
      [rot] "s"(rot), [as] "n"(AS), [km] "n"(KM), [gb] "n"(15 - J), [ps] "n"(PS), [psr] "n"(PSR)                         \
    : "v120", "v122", "v123", "v124", "s98", "s99"
    // the payload from the hashed halves (A = v122 left-aligned, B = v123): high word A << 10 | B >> (2K - 42), low word B << (74 - 2K)
#define KH_W64_PAY "v_lshrrev_b32 v124, %[psr], v123\n v_lshl_or_b32 %[khi], v122, 10, v124\n v_lshlrev_b32 %[klo], %[ps], v123\n"
#define KH_W64_CANON                                                \
    "v_cmp_lt_u64_e64 s[98:99], %[f], %[r]\n"                       \
    "v_cndmask_b32_e64 %[klo], %[rlo], %[flo], s[98:99]\n"          \
    "v_cndmask_b32_e64 %[khi], %[rhi], %[fhi], s[98:99]\n"
#define KH_W64_SPLIT "v_alignbit_b32 v122, %[khi], %[klo], %[as]\n v_and_b32 v123, %[km], %[klo]\n"   /* A (+ next bits: round 1 clears them), B */
#define KH_W64_SPLIT_K32 "v_mov_b32 v122, %[khi]\n v_mov_b32 v123, %[klo]\n"
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (K <= 24) {
        asm(KH_W64_CANON KH_W64_SPLIT KH_WIN_ROUNDS24 KH_W64_PAY KH_WIN_ADDR KH_W64_OPERANDS);
    } else if constexpr (K < 32) {
        asm(KH_W64_CANON KH_W64_SPLIT KH_WIN_ROUNDS32 KH_W64_PAY KH_WIN_ADDR KH_W64_OPERANDS);
    } else {
        asm(KH_W64_CANON KH_W64_SPLIT_K32 KH_WIN_ROUNDS32 KH_W64_PAY KH_WIN_ADDR KH_W64_OPERANDS);
    }
#else
    (void)f64; (void)r64; (void)rot; (void)f0; (void)f1; (void)f2; (void)f3; (void)KM; (void)AS; (void)PS; (void)PSR; (void)tm; (void)kmv;
    klo = khi = cnta = binb = 0;
#endif
#undef KH_W64_PAY
#undef KH_W64_SPLIT_K32
#undef KH_W64_SPLIT
#undef KH_W64_CANON
#undef KH_W64_OPERANDS
}
#endif  // __HIPCC__

}  // namespace kh
