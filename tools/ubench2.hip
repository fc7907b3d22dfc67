// ubench2.hip -- throughput of the integer / select / conversion instructions
// around the pair loop (independent destinations; development tool only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096;

// R = "independent" (8 destinations never read) or dependent chains
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

#define KERNEL(NAME, BODY)                                                      \
__global__ void __launch_bounds__(256) NAME(double *out, double seed,           \
                                            unsigned long long *clk)            \
{                                                                               \
    double s0 = seed + threadIdx.x, s1 = s0 * 1.1, s2 = s0 * 1.2;               \
    int i0 = threadIdx.x * 77 + 5, i1 = i0 * 3 + 1;                             \
    double d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;      \
    int e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0, e6 = 0, e7 = 0;         \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                       \
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                   \
    for (int i = 0; i < ITER; ++i) { BODY }                                     \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                       \
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] =                                \
        d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + e0 + e1 + e2 + e3 + e4 + e5 +   \
        e6 + e7;                                                                \
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; } \
}

#define D64 "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7)
#define D32 "=v"(e0), "=v"(e1), "=v"(e2), "=v"(e3), "=v"(e4), "=v"(e5), "=v"(e6), "=v"(e7)

KERNEL(k_cnd_vcc, asm volatile(
    "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
    "v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n"
    : D32 : "v"(i0), "v"(i1) : "vcc");)
KERNEL(k_cnd_sgpr, asm volatile(
    "v_cndmask_b32 %0, %8, %9, s[20:21]\n v_cndmask_b32 %1, %8, %9, s[20:21]\n v_cndmask_b32 %2, %8, %9, s[20:21]\n v_cndmask_b32 %3, %8, %9, s[20:21]\n"
    "v_cndmask_b32 %4, %8, %9, s[20:21]\n v_cndmask_b32 %5, %8, %9, s[20:21]\n v_cndmask_b32 %6, %8, %9, s[20:21]\n v_cndmask_b32 %7, %8, %9, s[20:21]\n"
    : D32 : "v"(i0), "v"(i1) : "s20", "s21");)
KERNEL(k_cnd_dep, asm volatile(
    "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n"
    "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n"
    : "+v"(e0) : "v"(i1) : "vcc");)
KERNEL(k_xor_dep, asm volatile(
    "v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n"
    "v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n"
    : "+v"(e0) : "v"(i1));)
#define OP2_32(NAME, OP) KERNEL(NAME, asm volatile( \
    OP " %0, %8, %9\n " OP " %1, %8, %9\n " OP " %2, %8, %9\n " OP " %3, %8, %9\n" \
    OP " %4, %8, %9\n " OP " %5, %8, %9\n " OP " %6, %8, %9\n " OP " %7, %8, %9\n" \
    : D32 : "v"(i0), "v"(i1));)
OP2_32(k_xor, "v_xor_b32")
OP2_32(k_mulhi, "v_mul_hi_u32")
OP2_32(k_mullo, "v_mul_lo_u32")
OP2_32(k_add32, "v_add_u32")
OP2_32(k_lshl, "v_lshlrev_b32")
KERNEL(k_bfi, asm volatile(
    "v_bfi_b32 %0, %8, %9, %8\n v_bfi_b32 %1, %8, %9, %8\n v_bfi_b32 %2, %8, %9, %8\n v_bfi_b32 %3, %8, %9, %8\n"
    "v_bfi_b32 %4, %8, %9, %8\n v_bfi_b32 %5, %8, %9, %8\n v_bfi_b32 %6, %8, %9, %8\n v_bfi_b32 %7, %8, %9, %8\n"
    : D32 : "v"(i0), "v"(i1));)
KERNEL(k_mov32, asm volatile(
    "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
    "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
    : D32 : "v"(i0));)
#define OP1_64(NAME, OP) KERNEL(NAME, asm volatile( \
    OP " %0, %8\n " OP " %1, %8\n " OP " %2, %8\n " OP " %3, %8\n" \
    OP " %4, %8\n " OP " %5, %8\n " OP " %6, %8\n " OP " %7, %8\n" \
    : D64 : "v"(s0));)
OP1_64(k_mov64, "v_mov_b64")
OP1_64(k_rndne, "v_rndne_f64")
OP1_64(k_floor, "v_floor_f64")
OP1_64(k_fract, "v_fract_f64")
OP1_64(k_frexpm, "v_frexp_mant_f64")
OP1_64(k_rsq, "v_rsq_f64")
KERNEL(k_frexpe, asm volatile(
    "v_frexp_exp_i32_f64 %0, %8\n v_frexp_exp_i32_f64 %1, %8\n v_frexp_exp_i32_f64 %2, %8\n v_frexp_exp_i32_f64 %3, %8\n"
    "v_frexp_exp_i32_f64 %4, %8\n v_frexp_exp_i32_f64 %5, %8\n v_frexp_exp_i32_f64 %6, %8\n v_frexp_exp_i32_f64 %7, %8\n"
    : D32 : "v"(s0));)
KERNEL(k_cvti, asm volatile(
    "v_cvt_i32_f64 %0, %8\n v_cvt_i32_f64 %1, %8\n v_cvt_i32_f64 %2, %8\n v_cvt_i32_f64 %3, %8\n"
    "v_cvt_i32_f64 %4, %8\n v_cvt_i32_f64 %5, %8\n v_cvt_i32_f64 %6, %8\n v_cvt_i32_f64 %7, %8\n"
    : D32 : "v"(s0));)
KERNEL(k_cvtd, asm volatile(
    "v_cvt_f64_i32 %0, %8\n v_cvt_f64_i32 %1, %8\n v_cvt_f64_i32 %2, %8\n v_cvt_f64_i32 %3, %8\n"
    "v_cvt_f64_i32 %4, %8\n v_cvt_f64_i32 %5, %8\n v_cvt_f64_i32 %6, %8\n v_cvt_f64_i32 %7, %8\n"
    : D64 : "v"(i0));)
KERNEL(k_ldexp, asm volatile(
    "v_ldexp_f64 %0, %8, %9\n v_ldexp_f64 %1, %8, %9\n v_ldexp_f64 %2, %8, %9\n v_ldexp_f64 %3, %8, %9\n"
    "v_ldexp_f64 %4, %8, %9\n v_ldexp_f64 %5, %8, %9\n v_ldexp_f64 %6, %8, %9\n v_ldexp_f64 %7, %8, %9\n"
    : D64 : "v"(s0), "v"(i0));)
KERNEL(k_cmp_sgpr, asm volatile(
    "v_cmp_lt_f64 s[20:21], %0, %1\n v_cmp_lt_f64 s[22:23], %0, %1\n v_cmp_lt_f64 s[20:21], %0, %1\n v_cmp_lt_f64 s[22:23], %0, %1\n"
    "v_cmp_lt_f64 s[20:21], %0, %1\n v_cmp_lt_f64 s[22:23], %0, %1\n v_cmp_lt_f64 s[20:21], %0, %1\n v_cmp_lt_f64 s[22:23], %0, %1\n"
    :: "v"(s0), "v"(s1) : "s20", "s21", "s22", "s23");)
KERNEL(k_fma_sgpr, asm volatile(
    "v_fma_f64 %0, %8, %8, s[20:21]\n v_fma_f64 %1, %8, %8, s[20:21]\n v_fma_f64 %2, %8, %8, s[20:21]\n v_fma_f64 %3, %8, %8, s[20:21]\n"
    "v_fma_f64 %4, %8, %8, s[20:21]\n v_fma_f64 %5, %8, %8, s[20:21]\n v_fma_f64 %6, %8, %8, s[20:21]\n v_fma_f64 %7, %8, %8, s[20:21]\n"
    : D64 : "v"(s0) : "s20", "s21");)
KERNEL(k_fma_lit, asm volatile(
    "v_fma_f64 %0, %8, %8, 0.5\n v_fma_f64 %1, %8, %8, 0.5\n v_fma_f64 %2, %8, %8, 0.5\n v_fma_f64 %3, %8, %8, 0.5\n"
    "v_fma_f64 %4, %8, %8, 0.5\n v_fma_f64 %5, %8, %8, 0.5\n v_fma_f64 %6, %8, %8, 0.5\n v_fma_f64 %7, %8, %8, 0.5\n"
    : D64 : "v"(s0));)
KERNEL(k_dsread128, {
    int addr = (threadIdx.x & 63) * 16;
    asm volatile(
    "ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
    "s_waitcnt lgkmcnt(0)\n"
    "ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
    "s_waitcnt lgkmcnt(0)\n"
    : "=v"(*(float4 *)&d0), "=v"(*(float4 *)&d2), "=v"(*(float4 *)&d4), "=v"(*(float4 *)&d6) : "v"(addr)); })
KERNEL(k_dsread2st64, {
    int addr = (threadIdx.x & 63) * 8;
    asm volatile(
    "ds_read2st64_b64 %0, %4 offset1:2\n ds_read2st64_b64 %1, %4 offset0:4 offset1:6\n ds_read2st64_b64 %2, %4 offset1:2\n ds_read2st64_b64 %3, %4 offset0:4 offset1:6\n"
    "s_waitcnt lgkmcnt(0)\n"
    "ds_read2st64_b64 %0, %4 offset1:2\n ds_read2st64_b64 %1, %4 offset0:4 offset1:6\n ds_read2st64_b64 %2, %4 offset1:2\n ds_read2st64_b64 %3, %4 offset0:4 offset1:6\n"
    "s_waitcnt lgkmcnt(0)\n"
    : "=v"(*(float4 *)&d0), "=v"(*(float4 *)&d2), "=v"(*(float4 *)&d4), "=v"(*(float4 *)&d6) : "v"(addr)); })
KERNEL(k_dsread64, {
    int addr = (threadIdx.x & 63) * 8;
    asm volatile(
    "ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
    "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n"
    "s_waitcnt lgkmcnt(0)\n"
    : D64 : "v"(addr)); })

KERNEL(k_cnd_e64_vcc, asm volatile(
    "v_cndmask_b32_e64 %0, %8, %9, vcc\n v_cndmask_b32_e64 %1, %8, %9, vcc\n v_cndmask_b32_e64 %2, %8, %9, vcc\n v_cndmask_b32_e64 %3, %8, %9, vcc\n"
    "v_cndmask_b32_e64 %4, %8, %9, vcc\n v_cndmask_b32_e64 %5, %8, %9, vcc\n v_cndmask_b32_e64 %6, %8, %9, vcc\n v_cndmask_b32_e64 %7, %8, %9, vcc\n"
    : D32 : "v"(i0), "v"(i1) : "vcc");)
KERNEL(k_cmp_cnd_vcc, asm volatile(
    "v_cmp_lt_f64 vcc, %8, %9\n v_cndmask_b32 %0, %10, %11, vcc\n v_cndmask_b32 %1, %10, %11, vcc\n v_cmp_lt_f64 vcc, %9, %8\n v_cndmask_b32 %2, %10, %11, vcc\n v_cndmask_b32 %3, %10, %11, vcc\n"
    "v_cmp_lt_f64 vcc, %8, %9\n v_cndmask_b32 %4, %10, %11, vcc\n"
    : D32 : "v"(s0), "v"(s1), "v"(i0), "v"(i1) : "vcc");)
KERNEL(k_cmp_cnd_sgpr, asm volatile(
    "v_cmp_lt_f64 s[20:21], %8, %9\n v_cndmask_b32 %0, %10, %11, s[20:21]\n v_cndmask_b32 %1, %10, %11, s[20:21]\n v_cmp_lt_f64 s[22:23], %9, %8\n v_cndmask_b32 %2, %10, %11, s[22:23]\n v_cndmask_b32 %3, %10, %11, s[22:23]\n"
    "v_cmp_lt_f64 s[20:21], %8, %9\n v_cndmask_b32 %4, %10, %11, s[20:21]\n"
    : D32 : "v"(s0), "v"(s1), "v"(i0), "v"(i1) : "s20", "s21", "s22", "s23");)
KERNEL(k_cmp_vcc, asm volatile(
    "v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %0, %1\n"
    "v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %0, %1\n"
    :: "v"(s0), "v"(s1) : "vcc");)
KERNEL(k_ds_add_f64, {
    int addr = ((threadIdx.x + i) & 63) * 8 + (threadIdx.x >> 6) * 4096;
    asm volatile(
    "ds_add_f64 %0, %1\n ds_add_f64 %0, %1 offset:512\n ds_add_f64 %0, %1 offset:1024\n ds_add_f64 %0, %1 offset:1536\n"
    "ds_add_f64 %0, %1 offset:2048\n ds_add_f64 %0, %1 offset:2560\n ds_add_f64 %0, %1 offset:3072\n ds_add_f64 %0, %1 offset:3584\n"
    "s_waitcnt lgkmcnt(0)\n"
    :: "v"(addr), "v"(s0) : "memory"); })
KERNEL(k_ds_add_f64_same, {
    int addr = ((threadIdx.x + i) & 63) * 8 + (threadIdx.x >> 6) * 4096;
    asm volatile(
    "ds_add_f64 %0, %1\n ds_add_f64 %0, %1\n ds_add_f64 %0, %1\n ds_add_f64 %0, %1\n"
    "ds_add_f64 %0, %1\n ds_add_f64 %0, %1\n ds_add_f64 %0, %1\n ds_add_f64 %0, %1\n"
    "s_waitcnt lgkmcnt(0)\n"
    :: "v"(addr), "v"(s0) : "memory"); })
KERNEL(k_ds_write_b64, {
    int addr = ((threadIdx.x + i) & 63) * 8 + (threadIdx.x >> 6) * 4096;
    asm volatile(
    "ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:512\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:1536\n"
    "ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1 offset:2560\n ds_write_b64 %0, %1 offset:3072\n ds_write_b64 %0, %1 offset:3584\n"
    "s_waitcnt lgkmcnt(0)\n"
    :: "v"(addr), "v"(s0) : "memory"); })
KERNEL(k_readlane, asm volatile(
    "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 5\n v_readlane_b32 s20, %0, 7\n v_readlane_b32 s21, %0, 9\n"
    "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 5\n v_readlane_b32 s20, %0, 7\n v_readlane_b32 s21, %0, 9\n"
    :: "v"(i0) : "s20", "s21");)
KERNEL(k_max64, asm volatile(
    "v_max_f64 %0, %8, %9\n v_max_f64 %1, %8, %9\n v_max_f64 %2, %8, %9\n v_max_f64 %3, %8, %9\n"
    "v_max_f64 %4, %8, %9\n v_max_f64 %5, %8, %9\n v_max_f64 %6, %8, %9\n v_max_f64 %7, %8, %9\n"
    : D64 : "v"(s0), "v"(s1));)
KERNEL(k_and_or, asm volatile(
    "v_and_or_b32 %0, %8, %9, %8\n v_and_or_b32 %1, %8, %9, %8\n v_and_or_b32 %2, %8, %9, %8\n v_and_or_b32 %3, %8, %9, %8\n"
    "v_and_or_b32 %4, %8, %9, %8\n v_and_or_b32 %5, %8, %9, %8\n v_and_or_b32 %6, %8, %9, %8\n v_and_or_b32 %7, %8, %9, %8\n"
    : D32 : "v"(i0), "v"(i1));)
KERNEL(k_dpp_mov_ror, asm volatile(
    "v_mov_b32_dpp %0, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
    "v_mov_b32_dpp %2, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
    "v_mov_b32_dpp %4, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
    "v_mov_b32_dpp %6, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
    : D32 : "v"(i0));)
KERNEL(k_dpp_mov_rowshr, asm volatile(
    "v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
    "v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
    "v_mov_b32_dpp %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
    "v_mov_b32_dpp %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
    : D32 : "v"(i0));)
KERNEL(k_mov64_dpp, asm volatile(
    "v_mov_b64_dpp %0, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
    "v_mov_b64_dpp %2, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
    "v_mov_b64_dpp %4, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %5, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
    "v_mov_b64_dpp %6, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %7, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
    : D64 : "v"(s0));)

KERNEL(k_mad_u64_u32, {
    unsigned long long q0; unsigned long long q1; unsigned long long q2; unsigned long long q3;
    asm volatile(
    "v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %4, %5, 0\n v_mad_u64_u32 %2, vcc, %4, %5, 0\n v_mad_u64_u32 %3, vcc, %4, %5, 0\n"
    "v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %4, %5, 0\n v_mad_u64_u32 %2, vcc, %4, %5, 0\n v_mad_u64_u32 %3, vcc, %4, %5, 0\n"
    : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(i0), "v"(i1) : "vcc");
    e0 += (int)q0 + (int)q1 + (int)q2 + (int)q3; })
KERNEL(k_log_f32, asm volatile(
    "v_log_f32 %0, %8\n v_log_f32 %1, %8\n v_log_f32 %2, %8\n v_log_f32 %3, %8\n"
    "v_log_f32 %4, %8\n v_log_f32 %5, %8\n v_log_f32 %6, %8\n v_log_f32 %7, %8\n"
    : D32 : "v"(i0));)
KERNEL(k_cvt_f64_u32, asm volatile(
    "v_cvt_f64_u32 %0, %8\n v_cvt_f64_u32 %1, %8\n v_cvt_f64_u32 %2, %8\n v_cvt_f64_u32 %3, %8\n"
    "v_cvt_f64_u32 %4, %8\n v_cvt_f64_u32 %5, %8\n v_cvt_f64_u32 %6, %8\n v_cvt_f64_u32 %7, %8\n"
    : D64 : "v"(i0));)

template <typename K>
int run(K kern, const char *name, int wavesPerSimd, int opsPerIter)
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int blocks = prop.multiProcessorCount * wavesPerSimd;
    double *out; unsigned long long *clk;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8)); CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0, clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0, clk);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double ghz = (double)h[0] / ((double)h[1] * 10.0);
    double cyc = ms * 1e-3 * ghz * 1e9 / ((double)ITER * opsPerIter * wavesPerSimd);
    printf("%-26s waves/SIMD=%d  %.3f ms  clk=%.2f GHz  cycles/wave-instr=%.2f\n", name, wavesPerSimd, ms, ghz, cyc);
    hipFree(out); hipFree(clk);
    return 0;
}

int main()
{
    for (int w : {8}) {
#define R(K, N) run(K, #K, w, N);
        R(k_cnd_vcc, 8) R(k_cnd_sgpr, 8) R(k_cnd_dep, 8) R(k_xor_dep, 8) R(k_xor, 8) R(k_mulhi, 8) R(k_mullo, 8)
        R(k_add32, 8) R(k_lshl, 8) R(k_bfi, 8) R(k_mov32, 8) R(k_mov64, 8) R(k_rndne, 8) R(k_floor, 8)
        R(k_fract, 8) R(k_frexpm, 8) R(k_rsq, 8) R(k_frexpe, 8) R(k_cvti, 8) R(k_cvtd, 8) R(k_ldexp, 8)
        R(k_cmp_sgpr, 8) R(k_fma_sgpr, 8) R(k_fma_lit, 8) R(k_dsread128, 8) R(k_dsread2st64, 8) R(k_dsread64, 8)
        R(k_cnd_e64_vcc, 8) R(k_cmp_cnd_vcc, 8) R(k_cmp_cnd_sgpr, 8) R(k_cmp_vcc, 8) R(k_ds_add_f64, 8) R(k_ds_add_f64_same, 8) R(k_ds_write_b64, 8) R(k_readlane, 8) R(k_max64, 8) R(k_and_or, 8) R(k_dpp_mov_ror, 8) R(k_dpp_mov_rowshr, 8) R(k_mov64_dpp, 8) R(k_mad_u64_u32, 8) R(k_log_f32, 8) R(k_cvt_f64_u32, 8)
    }
    return 0;
}
