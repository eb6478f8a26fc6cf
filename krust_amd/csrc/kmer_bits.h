// kmer_bits.h -- 2-bit k-mer arithmetic shared by host and device code.
//
// Device-side restatement of the reference's per-window work in closed form:
//   PACK_TABLE        src/kmer.rs:21-32    -> base_code()
//   from_sub validity src/kmer.rs:266-286  -> base_valid()
//   pack_bytes        src/kmer.rs:467-471  -> (acc << 2) | code, first base most significant
//   canonical         src/kmer.rs:348-390  -> min(fwd, revcomp(fwd)): codes A<C<G<T are in
//                                             ASCII order, so integer min == the reference's
//                                             lexicographic byte compare; on a tie (palindrome)
//                                             both sides are the same bits.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KH_HD __host__ __device__ __forceinline__
#else
#define KH_HD inline
#endif

#define KH_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull  // never a canonical key: canonical(T^k) == A^k == 0

// splitmix64 finaliser; used for table placement, shard ownership and the synthetic generator.
KH_HD uint64_t kh_mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull;
    z ^= z >> 27; z *= 0x94d049bb133111ebull;
    z ^= z >> 31;
    return z;
}

KH_HD uint64_t kh_kmask(uint32_t k) { return k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull); }

// A/a=0 C/c=1 G/g=2 T/t=3 (src/kmer.rs:21-32). ((b>>1)^(b>>2))&3 maps both cases;
// the popular (b>>1)&3 would give A,C,T,G order and change the canonical choice.
KH_HD uint32_t kh_base_code(uint32_t b) { return ((b >> 1) ^ (b >> 2)) & 3u; }

// Accepted bytes are exactly ACGTacgt (src/kmer.rs:271-273).
KH_HD uint32_t kh_base_valid(uint32_t b) {
    uint32_t u = b & 0xDFu;
    return (uint32_t)((u == 'A') | (u == 'C') | (u == 'G') | (u == 'T'));
}

KH_HD uint64_t kh_brev64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    return (x >> 32) | (x << 32);
#endif
}

// Reverse complement of a packed k-mer: complement = 3-code = ~code; reversing all 64 bits
// reverses the 2-bit groups and swaps the bits inside each group, so swap them back.
KH_HD uint64_t kh_revcomp(uint64_t fwd, uint32_t k) {
    uint64_t x = kh_brev64(~fwd);
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    return x >> (64 - 2 * k);
}

KH_HD uint64_t kh_canonical_bits(uint64_t fwd, uint32_t k) {
    uint64_t rc = kh_revcomp(fwd, k);
    return fwd < rc ? fwd : rc;
}

// ---- table hash: a BIJECTION on the 2k-bit key space, left-aligned in 64 bits -----------------
// Four Feistel rounds over the two k-bit halves of the packed k-mer; the round function is a
// 32-bit multiply, keeping the top k bits of the low word (rounds 1 and 3: kh_feistel_f) or bits k .. 2k-1 of the
// full product (rounds 2 and 4, since round 6: kh_feistel_g below).  Everything is 32-bit arithmetic (a 64-bit
// multiply is four 32-bit ones, this costs four in total), and
// being a bijection it lets the partitioned path carry 32-bit payloads instead of 64-bit keys
// whenever 2k minus the level-1 partition bits fits in 32 (k <= 21 at the headline table size):
// the bits of H that a partition level has consumed are implied by where the payload is stored,
// and the key is recovered with kh_unhash_n().  Quality (chi-square of region / in-region start
// occupancy on genomic, sequential, low-complexity and strided keys) matches splitmix64.
// For 16 <= k <= 24 the half fits 24 bits and the product uses the 24-bit multiplier (v_mul_u32_u24) with
// the constant's low 24 bits; both variants pass the same occupancy tests.  The choice depends only on k,
// so it is one function.  (Chosen on the assumption that v_mul_lo_u32 is quarter rate; measured on gfx950,
// tools/ubench/valu_rates.hip, both cost ~5 cycles per wave.  It stays because it is part of the table format.)
// MODE: which multiplier.  KH_MUL_AUTO decides from k at run time (one uniform branch per round: fine
// for cold code, but in the extraction kernels it splits every window into a dozen basic blocks and
// blocks instruction scheduling), so the hot kernels are instantiated for KH_MUL_24 / KH_MUL_32 and
// the host picks by k.
enum { KH_MUL_AUTO = 0, KH_MUL_24 = 1, KH_MUL_32 = 2 };
KH_HD bool kh_k_uses_mul24(uint32_t k) { return k >= 16 && k <= 24; }

template <int MODE = KH_MUL_AUTO>
KH_HD uint32_t kh_feistel_f(uint32_t r, uint32_t c, uint32_t k) {
    uint32_t t;
    if (MODE == KH_MUL_AUTO) {
        // decided at run time WITHOUT a branch: the low word of r x c is the same from the 24-bit and from the 32-bit
        // multiplier once the constant is chosen (r < 2^k <= 2^24 where the 24-bit one applies), so the choice is a
        // uniform select of the constant.  (Round 4: the branch per round split every key of region_count_kernel64 and
        // of the 64-bit level 2 into a dozen basic blocks -- 16 scalar branches per key.)
        t = r * (kh_k_uses_mul24(k) ? ((c & 0xFFFFFFu) | 1u) : c);
    } else if (MODE == KH_MUL_24) {
#if defined(__HIP_DEVICE_COMPILE__)
        // (__umul24() is a masked plain multiply to the compiler, re-selected as v_mul_u32_u24 only where
        // instruction selection can prove both operands 24-bit: in the extraction kernels two of the four rounds
        // come out as v_mul_lo_u32.  Forcing v_mul_u32_u24 -- by inline asm, or by the llvm.amdgcn.mul.u24
        // intrinsic -- was measured SLOWER on level 1 every time (S100M: +1.5 .. +14 ms, the scheduling and
        // register allocation around it change for the worse), so it stays as written.)
        t = __umul24(r, (c & 0xFFFFFFu) | 1u);
#else
        t = (uint32_t)((uint64_t)r * ((c & 0xFFFFFFu) | 1u));
#endif
    } else {
        t = r * c;
    }
    // the round keeps the TOP k bits of the product's low word: every input bit reaches them through
    // the carries.  (A t ^= t >> 15 here bought nothing measurable -- tools/hash_quality.py: region /
    // start chi-square and probe lengths equal splitmix64's with or without it -- and cost two of the
    // five instructions of a round in the extraction kernels.)
    return k < 32 ? (t >> (32 - k)) : t;
}
// ROUNDS 2 AND 4 (round 6 of the build: a new table hash): g(a) = bits k .. 2k-1 of the FULL product a x C -- on the device
// v_mul_hi_u32 of the half held LEFT-aligned, (a << (32 - k)) x C >> 32 == (a x C) >> k -- where rounds 1 and 3 keep the top k
// bits of the product's low word.  With the upper half of the hash kept left-aligned and the lower half right-aligned in the
// written-out window (window.hip.h), a round of either kind is a multiply and ONE v_bitop3 (xor under a mask): two instructions
// instead of multiply + shift + xor, -4 per window.  Which half gets which round matters: the level-1 digit comes from the half
// that rounds 1 and 3 update (tools/hash_quality.py: with the roles swapped the region chi-square fails from k = 27 up).
// Quality as the hash of rounds 1-5 and splitmix64 for k = 11 .. 32 on the tool's five key sets.
KH_HD uint32_t kh_feistel_g(uint32_t a, uint32_t c, uint32_t k) {
    const uint32_t mask = k < 32 ? ((1u << k) - 1u) : 0xFFFFFFFFu;
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a << ((32u - k) & 31u), c) & mask;  // (a < 2^k: the shift loses nothing)
#else
    return (uint32_t)(((uint64_t)a * c) >> k) & mask;
#endif
}
#define KH_FC0 0x9E3779B1u
#define KH_FC1 0x85EBCA77u
#define KH_FC2 0xC2B2AE3Du
#define KH_FC3 0x27D4EB2Fu

template <int MODE = KH_MUL_AUTO>
KH_HD uint64_t kh_hash_n(uint64_t key, uint32_t k) {
    const uint32_t mask = k < 32 ? ((1u << k) - 1u) : 0xFFFFFFFFu;
    uint32_t L = (uint32_t)(k < 32 ? (key >> k) : (key >> 32)) & mask, R = (uint32_t)key & mask, t;
    t = (L ^ kh_feistel_f<MODE>(R, KH_FC0, k)) & mask; L = R; R = t;
    t = (L ^ kh_feistel_g(R, KH_FC1, k)) & mask; L = R; R = t;
    t = (L ^ kh_feistel_f<MODE>(R, KH_FC2, k)) & mask; L = R; R = t;
    t = (L ^ kh_feistel_g(R, KH_FC3, k)) & mask; L = R; R = t;
    return k < 32 ? (((uint64_t)L << k) | R) : (((uint64_t)L << 32) | R);
}

template <int MODE = KH_MUL_AUTO>
KH_HD uint64_t kh_unhash_n(uint64_t h, uint32_t k) {
    const uint32_t mask = k < 32 ? ((1u << k) - 1u) : 0xFFFFFFFFu;
    uint32_t L = (uint32_t)(k < 32 ? (h >> k) : (h >> 32)) & mask, R = (uint32_t)h & mask, t;
    t = (R ^ kh_feistel_g(L, KH_FC3, k)) & mask; R = L; L = t;
    t = (R ^ kh_feistel_f<MODE>(L, KH_FC2, k)) & mask; R = L; L = t;
    t = (R ^ kh_feistel_g(L, KH_FC1, k)) & mask; R = L; L = t;
    t = (R ^ kh_feistel_f<MODE>(L, KH_FC0, k)) & mask; R = L; L = t;
    return k < 32 ? (((uint64_t)L << k) | R) : (((uint64_t)L << 32) | R);
}

// H: the 2k hash bits left-aligned in 64 bits.  Table placement reads it from the top:
//   region = H >> (64 - rbits), in-region start = the next REGION_BITS bits.
template <int MODE = KH_MUL_AUTO>
KH_HD uint64_t kh_table_hash(uint64_t key, uint32_t k) { return kh_hash_n<MODE>(key, k) << (64 - 2 * k); }
template <int MODE = KH_MUL_AUTO>
KH_HD uint64_t kh_table_unhash(uint64_t H, uint32_t k) { return kh_unhash_n<MODE>(H >> (64 - 2 * k), k); }

// Owner shard for the multi-GPU merge: a fast-range of the TOP bits of the table hash, so that
// ownership is a contiguous range of table regions (for a power-of-two shard count it is simply the
// top log2(n) bits).  A rank's table can therefore be exported region by region already grouped by
// owner, and an owner's shard is itself a table over the remaining hash bits (H << log2 n).
KH_HD uint32_t kh_owner_of_hash(uint64_t H, uint32_t nparts) {
    return (uint32_t)(((H >> 32) * (uint64_t)nparts) >> 32);
}
KH_HD uint32_t kh_owner_of(uint64_t key, uint32_t k, uint32_t nparts) {
    return kh_owner_of_hash(kh_table_hash(key, k), nparts);
}

// ---- synthetic reads (counter based; bit-identical to oracle/kmer_oracle.c ko_synth_reads) ----
#define KH_GOLDEN 0x9E3779B97F4A7C15ull
KH_HD uint64_t kh_stream_key(uint64_t seed, uint64_t s) { return kh_mix64(seed + (s + 1) * KH_GOLDEN); }
KH_HD uint64_t kh_draw(uint64_t key, uint64_t idx) { return kh_mix64(key + idx * KH_GOLDEN); }
