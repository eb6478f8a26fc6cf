// CPU check of krust_amd/csrc/window.hip.h (built and run by tests/test_window_plan.py; no GPU, no HIP):
// for every k = 11..32 and every window J = 0..15 of a lane, the fields win_fields<K, J>() cuts out of the lane's
// three code words and their reversed complements must be the packed forward k-mer (src/kmer.rs:467-471: first base
// most significant, A,C,G,T = 0..3) and its reverse complement (kmer_bits.h kh_revcomp), and win_outputs_ref<K>()
// -- the definition the asm of win_hash32 / win_hash64 is held to by the GPU parity tests -- must agree with
// kh_table_hash's top bits (level-1 digit) and the 32-bit payload of partition.hip.h (Pay<uint32_t>::make).
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "../krust_amd/csrc/window.hip.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { return rng_state = kh_mix64(rng_state + 0x632BE59BD9B4E019ull); }

static int failures = 0;

template <int K, int J>
static void check_one(const uint8_t *codes /* 48 two-bit codes, oldest first */, uint32_t w0, uint32_t w1, uint32_t w2) {
    const uint32_t c0 = kh::w_rev2_complement(w2), c1 = kh::w_rev2_complement(w1), c2 = kh::w_rev2_complement(w0);
    uint32_t flo, fhi, rlo, rhi;
    kh::win_fields<K, J>(w0, w1, w2, c0, c1, c2, flo, fhi, rlo, rhi);
    uint64_t want = 0;
    for (int i = 0; i < K; ++i) want = (want << 2) | codes[32 + J - K + 1 + i];
    const uint64_t fwd = ((uint64_t)fhi << 32) | flo, rc = ((uint64_t)rhi << 32) | rlo;
    if (fwd != want || rc != kh_revcomp(want, K)) {
        if (failures++ < 10) fprintf(stderr, "K=%d J=%d: fwd %llx want %llx rc %llx want %llx\n", K, J, (unsigned long long)fwd,
                                     (unsigned long long)want, (unsigned long long)rc, (unsigned long long)kh_revcomp(want, K));
    }
    // level-1 digit / payload definition vs the generic forms used by the C++ window
    const uint64_t key = fwd < rc ? fwd : rc;
    uint32_t p1, pay;
    kh::win_outputs_ref<K>(key, p1, pay);
    const uint64_t H = kh_table_hash(key, K);
    const uint32_t p1_want = (uint32_t)(H >> 54);
    const uint32_t pay_want = (uint32_t)((H << 10) >> 32);
    if (p1 != p1_want || (2 * K - 10 <= 32 && pay != pay_want)) {
        if (failures++ < 10) fprintf(stderr, "K=%d J=%d: p1 %u want %u pay %x want %x\n", K, J, p1, p1_want, pay, pay_want);
    }
}

// The arithmetic of the written-out rounds (window.hip.h, round 6), instruction for instruction on the host: A = the key's top 32 bits
// (what one v_alignbit / v_lshlrev leaves: the upper half LEFT-aligned with the key's next bits below it), B = the lower half; rounds
// 1 and 3 are (A ^ (B x C)) & top-K-bits -- which also clears what was below A --, rounds 2 and 4 B ^= mulhi(A, C) & low-K-bits.
// It must be kh_hash_n, the definition everything else uses, for every k the windows exist for.
static uint64_t asm_model(uint64_t key, int K) {
    const uint32_t km = K < 32 ? (1u << K) - 1u : 0xFFFFFFFFu, tm = K < 32 ? ~0u << (32 - K) : 0xFFFFFFFFu;
    uint32_t A = K == 32 ? (uint32_t)(key >> 32) : 2 * K >= 32 ? (uint32_t)(key >> (2 * K - 32)) : (uint32_t)key << (32 - 2 * K);
    uint32_t B = (uint32_t)key & km;
    const bool m24 = K >= 16 && K <= 24;
    const uint32_t c0 = m24 ? (KH_FC0 & 0xFFFFFFu) | 1u : KH_FC0, c2 = m24 ? (KH_FC2 & 0xFFFFFFu) | 1u : KH_FC2;
    A = (A ^ (B * c0)) & tm;
    B ^= (uint32_t)(((uint64_t)A * KH_FC1) >> 32) & km;
    A = (A ^ (B * c2)) & tm;
    B ^= (uint32_t)(((uint64_t)A * KH_FC3) >> 32) & km;
    const uint64_t L = K < 32 ? A >> (32 - K) : A;
    return (L << (K < 32 ? K : 32)) | B;
}

static void check_hash() {
    for (int k = 1; k <= 32; ++k) {
        const uint64_t kmask = kh_kmask((uint32_t)k);
        for (int rep = 0; rep < 4000; ++rep) {
            const uint64_t key = (rep == 0 ? 0ull : rep == 1 ? ~0ull : rep == 2 ? 1ull : rep == 3 ? 1ull << (2 * k - 1) : rnd()) & kmask;
            const uint64_t h = kh_hash_n((uint64_t)key, (uint32_t)k);
            bool bad = (h & ~kmask) != 0 || kh_unhash_n(h, (uint32_t)k) != key;   // a bijection of the 2k-bit keys
            if (k >= 11) bad = bad || asm_model(key, k) != h;
            if (bad && failures++ < 10) fprintf(stderr, "hash k=%d key %llx: h %llx back %llx model %llx\n", k, (unsigned long long)key,
                                                (unsigned long long)h, (unsigned long long)kh_unhash_n(h, (uint32_t)k), (unsigned long long)asm_model(key, k));
        }
    }
}

template <int K, int... Js>
static void check_k(std::integer_sequence<int, Js...>) {
    for (int rep = 0; rep < 200; ++rep) {
        uint8_t codes[48];
        uint32_t w[3] = {0, 0, 0};  // w[0] = w2 (oldest 16 bases) ... w[2] = w0
        for (int m = 0; m < 48; ++m) {
            codes[m] = (uint8_t)(rep < 4 ? (rep & 3) : (rnd() & 3));   // (homopolymers first: palindromic / extreme fields)
            w[m / 16] |= (uint32_t)codes[m] << (30 - 2 * (m % 16));
        }
        (check_one<K, Js>(codes, w[2], w[1], w[0]), ...);
    }
}

template <int... Ks>
static void check_all(std::integer_sequence<int, Ks...>) {
    (check_k<Ks + 11>(std::make_integer_sequence<int, 16>()), ...);
}

int main() {
    check_all(std::make_integer_sequence<int, 22>());  // k = 11..32
    check_hash();
    if (failures) {
        fprintf(stderr, "%d failures\n", failures);
        return 1;
    }
    printf("WINDOW_PLAN_OK 22 x 16\n");
    return 0;
}
