// Micro-benchmark: issue cost of the VALU instructions the counting kernels are made of, on gfx950.
// 256 workgroups x 1024 lanes (4 waves per SIMD), per lane 8 independent dependent-chains of N_ITER instructions.
// Prints cycles per wave-instruction per SIMD assuming 2.4 GHz.  Build (cross-compiles without a GPU):
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o valu_rates valu_rates.hip
// Results and what they changed: profiles/README.md (r02c).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 4096
#define OPS(ASM)                                                                                         \
    for (int i = 0; i < N_ITER; ++i) {                                                                   \
        asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c));                                                 \
        asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c));                                                 \
        asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c));                                                 \
        asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c));                                                 \
        asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c));                                                 \
        asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c));                                                 \
        asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c));                                                 \
        asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c));                                                 \
    }
struct Case { const char *name; const char *text; int ninst; };
#define CASES(X) \
    X(0, "v_xor_b32 (VOP2)", "v_xor_b32 %0, %0, %1", 1) \
    X(1, "v_and_b32 (VOP2)", "v_and_b32 %0, %0, %1", 1) \
    X(2, "v_or_b32 (VOP2)", "v_or_b32 %0, %0, %1", 1) \
    X(3, "v_add_u32 (VOP2)", "v_add_u32 %0, %0, %1", 1) \
    X(4, "v_sub_u32 (VOP2)", "v_sub_u32 %0, %0, %1", 1) \
    X(5, "v_lshlrev_b32 imm (VOP2)", "v_lshlrev_b32 %0, 3, %0", 1) \
    X(6, "v_lshrrev_b32 imm (VOP2)", "v_lshrrev_b32 %0, 3, %0", 1) \
    X(7, "v_lshrrev_b32 vgpr (VOP2)", "v_lshrrev_b32 %0, %1, %0", 1) \
    X(8, "v_min_u32 (VOP2)", "v_min_u32 %0, %0, %1", 1) \
    X(9, "v_max_u32 (VOP2)", "v_max_u32 %0, %0, %1", 1) \
    X(10, "v_mov_b32 (VOP1)", "v_mov_b32 %0, %1", 1) \
    X(11, "v_not_b32 (VOP1)", "v_not_b32 %0, %0", 1) \
    X(12, "v_bfrev_b32 (VOP1)", "v_bfrev_b32 %0, %0", 1) \
    X(13, "v_mul_u32_u24 (VOP2)", "v_mul_u32_u24 %0, %0, %1", 1) \
    X(14, "v_mul_lo_u32 (VOP3)", "v_mul_lo_u32 %0, %0, %1", 1) \
    X(15, "v_mul_hi_u32 (VOP3)", "v_mul_hi_u32 %0, %0, %1", 1) \
    X(16, "v_mad_u32_u24 (VOP3)", "v_mad_u32_u24 %0, %0, %1, %2", 1) \
    X(17, "v_alignbit_b32 (VOP3)", "v_alignbit_b32 %0, %0, %1, 7", 1) \
    X(18, "v_lshl_or_b32 (VOP3)", "v_lshl_or_b32 %0, %0, 2, %1", 1) \
    X(19, "v_lshl_add_u32 (VOP3)", "v_lshl_add_u32 %0, %0, 2, %1", 1) \
    X(20, "v_and_or_b32 (VOP3)", "v_and_or_b32 %0, %0, %1, %2", 1) \
    X(21, "v_add3_u32 (VOP3)", "v_add3_u32 %0, %0, %1, %2", 1) \
    X(22, "v_xad_u32 (VOP3)", "v_xad_u32 %0, %0, %1, %2", 1) \
    X(23, "v_bfe_u32 (VOP3)", "v_bfe_u32 %0, %0, 1, 31", 1) \
    X(24, "v_bfi_b32 (VOP3)", "v_bfi_b32 %0, %1, %0, %2", 1) \
    X(25, "v_perm_b32 (VOP3)", "v_perm_b32 %0, %0, %1, %2", 1) \
    X(26, "v_bitop3_b32 (VOP3)", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x78", 1) \
    X(27, "v_xor_b32_e64 (VOP3 enc)", "v_xor_b32_e64 %0, %0, %1", 1) \
    X(28, "v_xor_b32 sgpr src0", "v_xor_b32 %0, s4, %0", 1) \
    X(29, "v_xor_b32 literal", "v_xor_b32 %0, 0x12345, %0", 1) \
    X(30, "v_cmp_lt_u32 vcc + v_cndmask vcc", "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc", 2) \
    X(31, "v_cmp_lt_u32 vcc + 2 v_cndmask vcc", "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %2, vcc", 3) \
    X(32, "v_cmp_lt_u32 vcc + 4 v_cndmask vcc", "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %2, vcc", 5) \
    X(33, "v_cmp_e64 sgpr + v_cndmask_e64", "v_cmp_lt_u32_e64 s[10:11], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]", 2) \
    X(34, "v_cmp_e64 sgpr + 2 v_cndmask_e64", "v_cmp_lt_u32_e64 s[10:11], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n v_cndmask_b32_e64 %0, %0, %2, s[10:11]", 3) \
    X(35, "v_cndmask_e64 stale sgpr mask", "v_cndmask_b32_e64 %0, %0, %1, s[10:11]", 1) \
    X(36, "v_cndmask vcc, stale vcc", "v_cndmask_b32 %0, %0, %1, vcc", 1) \
    X(37, "v_cmp_lt_u32 vcc alone", "v_cmp_lt_u32 vcc, %0, %1", 1) \
    X(38, "v_cmp_lt_u64 vcc alone", "v_cmp_lt_u64 vcc, %3, %4", 1) \
    X(39, "v_cmp_lt_u64 + 2 v_cndmask (min64)", "v_cmp_lt_u64 vcc, %3, %4\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %2, vcc", 3) \
    X(40, "v_lshlrev_b64 (VOP3)", "v_lshlrev_b64 %3, 3, %3", 1) \
    X(41, "v_lshl_add_u64 (VOP3)", "v_lshl_add_u64 %3, %3, 0, %4", 1) \
    X(42, "v_add_co_u32 + v_addc_co_u32", "v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %2, vcc", 2) \
    X(43, "v_mbcnt_lo+hi", "v_mbcnt_lo_u32_b32 %0, %1, %0\n v_mbcnt_hi_u32_b32 %0, %2, %0", 2) \
    X(44, "v_readlane + v_xor", "v_readlane_b32 s12, %1, 5\n v_xor_b32 %0, s12, %0", 2) \
    X(45, "v_mov_b32 dpp row_shr:1 + v_xor", "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_xor_b32 %0, %0, %1", 2) \
    X(46, "v_pk_add_u16 (VOP3P)", "v_pk_add_u16 %0, %0, %1", 1) \
    X(47, "v_xor x2 independent regs", "v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %2", 2)

template <int WHICH>
__global__ __launch_bounds__(1024) void k(uint32_t *out, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = seed | 0x9E3779u, c = seed * 77u + 5u;
    uint64_t d = ((uint64_t)a0 << 32) | b, e = ((uint64_t)b << 32) | a0;
    asm volatile("s_mov_b32 s10, 0x33333333\n s_mov_b32 s11, 0x33333333\n s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555" ::: "s10", "s11", "s12", "vcc");
#undef OPS
#define OPS(ASM)                                                                                         \
    for (int i = 0; i < N_ITER; ++i) {                                                                   \
        asm volatile(ASM : "+v"(a0), "+v"(b), "+v"(c), "+v"(d), "+v"(e) :: "s12", "s10", "s11", "vcc");   \
        asm volatile(ASM : "+v"(a1), "+v"(b), "+v"(c), "+v"(d), "+v"(e) :: "s12", "s10", "s11", "vcc");   \
        asm volatile(ASM : "+v"(a2), "+v"(b), "+v"(c), "+v"(d), "+v"(e) :: "s12", "s10", "s11", "vcc");   \
        asm volatile(ASM : "+v"(a3), "+v"(b), "+v"(c), "+v"(d), "+v"(e) :: "s12", "s10", "s11", "vcc");   \
        asm volatile(ASM : "+v"(a4), "+v"(b), "+v"(c), "+v"(d), "+v"(e) :: "s12", "s10", "s11", "vcc");   \
        asm volatile(ASM : "+v"(a5), "+v"(b), "+v"(c), "+v"(d), "+v"(e) :: "s12", "s10", "s11", "vcc");   \
        asm volatile(ASM : "+v"(a6), "+v"(b), "+v"(c), "+v"(d), "+v"(e) :: "s12", "s10", "s11", "vcc");   \
        asm volatile(ASM : "+v"(a7), "+v"(b), "+v"(c), "+v"(d), "+v"(e) :: "s12", "s10", "s11", "vcc");   \
    }
#define X(I, NAME, TEXT, N) if (WHICH == I) { OPS(TEXT) }
    CASES(X)
#undef X
    out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b ^ c ^ (uint32_t)d ^ (uint32_t)e;
}
template <typename F>
static float run(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    uint32_t *out; hipMalloc(&out, 256 * 1024 * 4);
#define X(I, NAME, TEXT, N) { const float ms = run([&] { hipLaunchKernelGGL(k<I>, dim3(256), dim3(1024), 0, 0, out, 1u); }); \
        printf("%-40s %7.3f ms  %6.2f cycles per wave-instruction (%d per step: %6.2f per step)\n", NAME, ms, ms * 1e-3 * 2.4e9 / (4.0 * 8 * N_ITER * N), N, ms * 1e-3 * 2.4e9 / (4.0 * 8 * N_ITER)); }
    CASES(X)
#undef X
    return 0;
}
