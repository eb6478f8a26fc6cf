// stub_kmerhip.cpp -- TEST INFRASTRUCTURE ONLY: a recording stand-in for the kh_* C ABI (include/kmerhip.h).
//
// Used by exactly one build: `make -C krust_amd/host asan`, which compiles the C++ host layer (reader,
// record cutting, chunk growth, KMIX index, CLI) with -fsanitize=address,undefined so that its buffer handling
// can be exercised and fuzzed on a machine without a GPU (GPU AddressSanitizer runs are not available on this
// pool; the reference fuzzes the callers of its parsers the same way: fuzz/fuzz_targets/*.rs).  It counts
// NOTHING -- every result is empty -- and it is never linked into `kmerust` or libkmerust_host.so.
//
//   KH_STUB_LOG=<file>   every push is appended: "PUSH <n> <has_qual>\n" + bases [+ qual], "TEXT <n> <format>\n" + text
//   KH_STUB_TEXT=1       kh_push_text accepts the text (default: KH_ERR_FORMAT, so the host line parser runs);
//                        KH_STUB_TEXT=refuse:<i> accepts until the i-th chunk and refuses that one
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kmerhip.h"
#include "../../krust_amd/csrc/kmer_bits.h"

struct kh_ctx {
    uint32_t k;
    std::string last_error;
};
struct kh_group {
    std::vector<kh_ctx *> ctx;
};

namespace {
std::mutex g_mu;
long g_text_chunks = 0;
void log_blob(const char *head, const void *a, uint64_t na, const void *b, uint64_t nb) {
    const char *path = getenv("KH_STUB_LOG");
    if (!path) return;
    std::lock_guard<std::mutex> lk(g_mu);
    FILE *f = fopen(path, "ab");
    if (!f) return;
    fputs(head, f);
    if (na) fwrite(a, 1, na, f);
    if (nb) fwrite(b, 1, nb, f);
    fclose(f);
}
}  // namespace

extern "C" {
int kh_abi_version(void) { return KMERHIP_ABI_VERSION; }
int kh_create(kh_ctx **out, const kh_config *cfg) {
    if (!out || !cfg || cfg->struct_size != sizeof(kh_config)) return KH_ERR_BAD_ARG;
    if (cfg->k < 1 || cfg->k > 32) return KH_ERR_BAD_K;
    *out = new kh_ctx{cfg->k, ""};
    return KH_OK;
}
void kh_destroy(kh_ctx *c) { delete c; }
int kh_reset(kh_ctx *c) {
    log_blob("RESET\n", nullptr, 0, nullptr, 0);
    return c ? KH_OK : KH_ERR_BAD_ARG;
}
int kh_push(kh_ctx *c, const uint8_t *bases, const uint8_t *qual, uint64_t n) {
    if (!c || (n && !bases)) return KH_ERR_BAD_ARG;
    char head[64];
    snprintf(head, sizeof head, "PUSH %llu %d\n", (unsigned long long)n, qual ? 1 : 0);
    // reading every byte is the point: ASan sees an over-long n or a dangling buffer
    log_blob(head, bases, n, qual, qual ? n : 0);
    volatile uint8_t sink = 0;
    for (uint64_t i = 0; i < n; ++i) sink ^= bases[i] ^ (qual ? qual[i] : 0);
    (void)sink;
    return KH_OK;
}
int kh_push_device(kh_ctx *, const uint8_t *, const uint8_t *, uint64_t) { return KH_ERR_NO_DEVICE; }
int kh_push_text(kh_ctx *c, const uint8_t *text, uint64_t n, int format) {
    if (!c || (n && !text)) return KH_ERR_BAD_ARG;
    volatile uint8_t sink = 0;
    for (uint64_t i = 0; i < n; ++i) sink ^= text[i];
    (void)sink;
    const char *mode = getenv("KH_STUB_TEXT");
    if (!mode) return KH_ERR_FORMAT;
    long idx;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        idx = g_text_chunks++;
    }
    if (!strncmp(mode, "refuse:", 7) && idx >= atol(mode + 7)) return KH_ERR_FORMAT;
    char head[64];
    snprintf(head, sizeof head, "TEXT %llu %d\n", (unsigned long long)n, format);
    log_blob(head, text, n, nullptr, 0);
    return KH_OK;
}
int kh_push_text_device(kh_ctx *, const uint8_t *, uint64_t, int) { return KH_ERR_NO_DEVICE; }
// "pinned" memory of the stub is plain heap memory: what matters under ASan is that the host keeps inside it and frees it once
int kh_host_alloc(void **out, uint64_t bytes) {
    if (!out) return KH_ERR_BAD_ARG;
    *out = getenv("KH_STUB_NO_PINNED") ? nullptr : malloc(bytes ? bytes : 1);
    return *out ? KH_OK : KH_ERR_OOM;
}
int kh_host_free(void *p) {
    free(p);
    return KH_OK;
}
int kh_host_register(void *, uint64_t) { return KH_OK; }
int kh_host_unregister(void *) { return KH_OK; }
int kh_finish(kh_ctx *c, kh_stats *st) {
    if (st) memset(st, 0, sizeof(*st));
    return c ? KH_OK : KH_ERR_BAD_ARG;
}
int kh_result_size(kh_ctx *, uint64_t, uint64_t *n) { if (n) *n = 0; return KH_OK; }
int kh_result_copy(kh_ctx *, uint64_t *, uint64_t *, uint64_t, uint64_t, uint64_t *n) { if (n) *n = 0; return KH_OK; }
int kh_result_copy_device(kh_ctx *, uint64_t *, uint64_t *, uint64_t, uint64_t, uint64_t *n) { if (n) *n = 0; return KH_OK; }
int kh_histogram(kh_ctx *, uint64_t, uint64_t *, uint64_t *, uint64_t, uint64_t *n) { if (n) *n = 0; return KH_OK; }
int kh_lookup(kh_ctx *, const uint64_t *, uint64_t n, uint64_t *counts) {
    for (uint64_t i = 0; i < n; ++i) counts[i] = 0;
    return KH_OK;
}
int kh_group_create(kh_group **out, const kh_config *cfg, const int32_t *devices, uint32_t n) {
    if (!out || !cfg || !devices || n < 1 || n > 64) return KH_ERR_BAD_ARG;
    kh_group *g = new kh_group();
    for (uint32_t i = 0; i < n; ++i) {
        kh_ctx *c = nullptr;
        if (kh_create(&c, cfg) != KH_OK) return KH_ERR_BAD_ARG;
        g->ctx.push_back(c);
    }
    *out = g;
    return KH_OK;
}
kh_ctx *kh_group_ctx(kh_group *g, uint32_t r) { return (g && r < g->ctx.size()) ? g->ctx[r] : nullptr; }
uint32_t kh_group_size(const kh_group *g) { return g ? (uint32_t)g->ctx.size() : 0; }
int kh_group_merge(kh_group *g, kh_merge_info *) { return g ? KH_OK : KH_ERR_BAD_ARG; }
void kh_group_destroy(kh_group *g) {
    if (!g) return;
    for (kh_ctx *c : g->ctx) delete c;
    delete g;
}
// the pure helpers are real (kmerust query uses them): same arithmetic as the library, from the shared header
int kh_pack(const uint8_t *bases, uint32_t k, uint64_t *packed, uint32_t *err_pos) {
    if (!bases || !packed) return KH_ERR_BAD_ARG;
    if (k < 1 || k > 32) return KH_ERR_BAD_K;
    uint64_t acc = 0;
    for (uint32_t i = 0; i < k; ++i) {
        if (!kh_base_valid(bases[i])) {
            if (err_pos) *err_pos = i;
            return KH_ERR_BAD_ARG;
        }
        acc = (acc << 2) | kh_base_code(bases[i]);
    }
    *packed = acc;
    return KH_OK;
}
int kh_unpack(uint64_t packed, uint32_t k, uint8_t *out) {
    if (!out) return KH_ERR_BAD_ARG;
    if (k < 1 || k > 32) return KH_ERR_BAD_K;
    for (uint32_t i = 0; i < k; ++i) out[i] = (uint8_t)"ACGT"[(packed >> (2 * (k - 1 - i))) & 3u];
    return KH_OK;
}
int kh_canonical(uint64_t packed, uint32_t k, uint64_t *canonical, int *is_rc) {
    if (!canonical) return KH_ERR_BAD_ARG;
    if (k < 1 || k > 32) return KH_ERR_BAD_K;
    packed &= kh_kmask(k);
    const uint64_t rc = kh_revcomp(packed, k);
    *canonical = packed < rc ? packed : rc;
    if (is_rc) *is_rc = rc < packed;
    return KH_OK;
}
const char *kh_strerror(int s) { return s == KH_OK ? "ok" : s == KH_ERR_FORMAT ? "text layout not accepted" : "stub error"; }
const char *kh_last_error(const kh_ctx *c) { return c ? c->last_error.c_str() : ""; }
}
