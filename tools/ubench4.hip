// ubench4.hip -- what do scalar instructions cost beside a saturated vector ALU
// on gfx950?  (development tool; numbers in profiles/r03_ubench4_scalar_cost.txt)
// Each kernel repeats a body of 8 fp64 FMAs plus a dose of scalar work: plain
// SALU adds, SALU adds interleaved with the FMAs, untaken / taken branches, and
// the compare -> s_and_saveexec -> s_or exec pattern of a divergent `if`.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 2048;

#define OUTS : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7), "+s"(c0), "+s"(c1)
#define INS : "v"(s0), "v"(s1)
#define F(i) "v_fma_f64 %" #i ", %10, %11, %10\n"
#define SA "s_add_u32 %8, %8, 1\n"
#define SB "s_add_u32 %9, %9, 3\n"
#define FMA8 asm volatile(F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) OUTS INS);
#define S8 asm volatile(SA SB SA SB SA SB SA SB OUTS INS : "scc");
#define FMA8_S8I asm volatile(F(0) SA F(1) SB F(2) SA F(3) SB F(4) SA F(5) SB F(6) SA F(7) SB OUTS INS : "scc");
#define FMA8_S16I asm volatile(F(0) SA SB F(1) SA SB F(2) SA SB F(3) SA SB F(4) SA SB F(5) SA SB F(6) SA SB F(7) SA SB OUTS INS : "scc");
// untaken branches (scc = 0 after s_cmp_eq of different values)
#define BR_UNTAKEN4 asm volatile("s_cmp_eq_u32 0, 1\n s_cbranch_scc1 1f\n s_cbranch_scc1 1f\n s_cbranch_scc1 1f\n s_cbranch_scc1 1f\n1:\n" OUTS INS : "scc");
// taken branches to the next instruction
#define BR_TAKEN4 asm volatile("s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1f\n1: s_cbranch_scc1 2f\n2: s_cbranch_scc1 3f\n3: s_cbranch_scc1 4f\n4:\n" OUTS INS : "scc");
// divergent-if skeleton, twice: compare, save/and exec, one FMA inside, restore
#define EXEC2 asm volatile(                                                   \
    "v_cmp_lt_f64 vcc, %10, %11\n s_and_saveexec_b64 s[20:21], vcc\n"         \
    "s_cbranch_execz 1f\n v_fma_f64 %0, %10, %11, %10\n1: s_or_b64 exec, exec, s[20:21]\n" \
    "v_cmp_gt_f64 vcc, %10, %11\n s_and_saveexec_b64 s[20:21], vcc\n"         \
    "s_cbranch_execz 2f\n v_fma_f64 %1, %10, %11, %10\n2: s_or_b64 exec, exec, s[20:21]\n" \
    OUTS INS : "vcc", "s20", "s21", "scc");
#define F6 asm volatile(F(2) F(3) F(4) F(5) F(6) F(7) OUTS INS);

#define KERNEL(NAME, BODY)                                                      \
__global__ void __launch_bounds__(256) NAME(double *out, double seed,           \
                                            unsigned long long *clk)            \
{                                                                               \
    double s0 = seed + threadIdx.x, s1 = s0 * 1.1;                              \
    double d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;      \
    unsigned c0 = 0, c1 = 0;                                                    \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                       \
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                   \
    for (int i = 0; i < ITER; ++i) { BODY }                                     \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                       \
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 +  \
        d6 + d7 + (double)(c0 + c1);                                            \
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; } \
}

KERNEL(k_fma8, FMA8)
KERNEL(k_fma8_s8, FMA8 S8)
KERNEL(k_fma8_s16, FMA8 S8 S8)
KERNEL(k_fma8_s32, FMA8 S8 S8 S8 S8)
KERNEL(k_fma8_s8_interleaved, FMA8_S8I)
KERNEL(k_fma8_s16_interleaved, FMA8_S16I)
KERNEL(k_fma8_br4_untaken, FMA8 BR_UNTAKEN4)
KERNEL(k_fma8_br4_taken, FMA8 BR_TAKEN4)
KERNEL(k_fma8_exec2, F6 EXEC2)
KERNEL(k_s32, S8 S8 S8 S8)

template <typename K>
int run(K kern, const char *name, int wavesPerSimd)
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
    double cyc = ms * 1e-3 * ghz * 1e9 / ((double)ITER * wavesPerSimd);
    printf("%-26s waves/SIMD=%d  %.3f ms  clk=%.2f GHz  SIMD cycles per wave-iteration=%.1f\n",
           name, wavesPerSimd, ms, ghz, cyc);
    fflush(stdout);
    hipFree(out); hipFree(clk);
    return 0;
}

int main()
{
    for (int w : {1, 2, 8}) {
#define R(K) run(K, #K, w);
        R(k_fma8) R(k_fma8_s8) R(k_fma8_s16) R(k_fma8_s32) R(k_fma8_s8_interleaved)
        R(k_fma8_s16_interleaved) R(k_fma8_br4_untaken) R(k_fma8_br4_taken)
        R(k_fma8_exec2) R(k_s32)
    }
    return 0;
}
