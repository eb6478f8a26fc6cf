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

// Owner shard for the multi-GPU key-partitioned merge.  Uses the high half of a re-mixed
// hash so that it is independent of the table placement bits.
KH_HD uint32_t kh_owner_of(uint64_t key, uint32_t nparts) {
    uint64_t h = kh_mix64(key ^ 0x6a09e667f3bcc909ull);
    return (uint32_t)(((h >> 32) * (uint64_t)nparts) >> 32);
}

// ---- synthetic reads (counter based; bit-identical to oracle/kmer_oracle.c ko_synth_reads) ----
#define KH_GOLDEN 0x9E3779B97F4A7C15ull
KH_HD uint64_t kh_stream_key(uint64_t seed, uint64_t s) { return kh_mix64(seed + (s + 1) * KH_GOLDEN); }
KH_HD uint64_t kh_draw(uint64_t key, uint64_t idx) { return kh_mix64(key + idx * KH_GOLDEN); }
