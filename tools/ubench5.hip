// ubench5.hip -- is the fp64 vector pipe of gfx950 POWER-limited under sustained
// load?  (development tool; numbers in profiles/r04_ubench5_power.txt)
// Round 4 removed 38 cheap integer instructions (Philox xors, wide multiplies)
// per chain-step from the headline kernel -- 3.3 % of its issue cycles -- and
// gained 1.0 %.  Earlier micro-benchmarks showed the clock at 2.0-2.1 GHz under
// fp64 FMAs and 2.4 GHz under integer work.  If the chip runs against a power
// cap, time follows the ENERGY of the instruction mix, not its issue slots.
// Each kernel here runs for seconds (the earlier ones ran 0.3 ms): bodies of 8
// instructions per trip -- fp64 FMAs on all lanes; on half / a quarter of the lanes; with all-zero operands (no bits toggling); integer xors; a mix.
// Reported: the clock the kernel ran at (s_memtime against the 100 MHz
// s_memrealtime) and instructions per second per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 1 << 16;

#define F(i) "v_fma_f64 %" #i ", %8, %9, %" #i "\n"
#define M(i) "v_mul_f64 %" #i ", %8, %" #i "\n"
#define X(i) "v_xor_b32 %" #i ", %10, %" #i "\n"
#define DOUTS : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
#define DINS : "v"(s0), "v"(s1), "v"(x0)
#define FMA8 asm volatile(F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) DOUTS DINS);
#define MUL8 asm volatile(M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) DOUTS DINS);
#define XOUTS : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
#define XOR8 asm volatile(X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) XOUTS DINS);

#define KERNEL(NAME, PRE, BODY)                                                 \
__global__ void __launch_bounds__(256) NAME(double *out, double seed,           \
                                            unsigned long long *clk)            \
{                                                                               \
    double s0 = seed + 1e-9 * threadIdx.x, s1 = 1e-3 * s0;                      \
    double d0 = s0, d1 = s1, d2 = s0 + 1, d3 = s1 + 1, d4 = s0 + 2, d5 = s1 + 2,\
           d6 = s0 + 3, d7 = s1 + 3;                                            \
    unsigned x0 = threadIdx.x * 2654435761u + 12345u;                           \
    unsigned i0 = x0, i1 = x0 + 1, i2 = x0 + 2, i3 = x0 + 3, i4 = x0 + 4,        \
             i5 = x0 + 5, i6 = x0 + 6, i7 = x0 + 7;                             \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                       \
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                   \
    /* (the lanes that run the loop are chosen by an ordinary branch: the      \
       compiler manages the exec mask) */                                      \
    PRE { for (int i = 0; i < ITER; ++i) { BODY } }                             \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                       \
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 +  \
        d6 + d7 + (double)(i0 ^ i1 ^ i2 ^ i3 ^ i4 ^ i5 ^ i6 ^ i7);              \
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; } \
}

#define NOPRE if (true)
#define HALF if ((threadIdx.x & 1) == 0)
#define QUARTER if ((threadIdx.x & 3) == 0)
#define ZERO s0 = 0; s1 = 0; d0 = d1 = d2 = d3 = d4 = d5 = d6 = d7 = 0; if (true)
KERNEL(k_fma8, NOPRE, FMA8)
KERNEL(k_fma8_half_lanes, HALF, FMA8)
KERNEL(k_fma8_quarter_lanes, QUARTER, FMA8)
KERNEL(k_fma8_zero_operands, ZERO, FMA8)
KERNEL(k_mul8, NOPRE, MUL8)
KERNEL(k_xor8, NOPRE, XOR8)
KERNEL(k_fma8_xor8, NOPRE, FMA8 XOR8)
KERNEL(k_fma4_xor8, NOPRE, asm volatile(F(0) F(1) F(2) F(3) DOUTS DINS); XOR8)

template <typename K>
int run(K kern, const char *name, int wavesPerSimd, int ninstr, double seconds)
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int blocks = prop.multiProcessorCount * wavesPerSimd;
    double *out; unsigned long long *clk;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8)); CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // one launch to learn the duration, then enough back-to-back launches for
    // `seconds`; the LAST launch is the one timed (steady state)
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0, clk);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    int reps = (int)(seconds * 1e3 / ms) + 1;
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0, clk);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0, clk);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double ghz = (double)h[0] / ((double)h[1] * 10.0);
    double ips = (double)ITER * ninstr * wavesPerSimd / (ms * 1e-3);   // per SIMD
    printf("%-24s waves/SIMD=%d  %8.3f ms  clk=%.3f GHz  %.3f G wave-instr/s per SIMD  "
           "cycles/instr=%.2f\n", name, wavesPerSimd, ms, ghz, ips * 1e-9,
           ghz * 1e9 / ips);
    fflush(stdout);
    hipFree(out); hipFree(clk);
    return 0;
}

int main(int argc, char **argv)
{
    double seconds = argc > 1 ? atof(argv[1]) : 1.5;
    for (int w : {8, 4}) {
#define R(K, N) run(K, #K, w, N, seconds);
        R(k_fma8, 8) R(k_fma8_half_lanes, 8) R(k_fma8_quarter_lanes, 8)
        R(k_fma8_zero_operands, 8) R(k_mul8, 8) R(k_xor8, 8) R(k_fma8_xor8, 16)
        R(k_fma4_xor8, 12)
    }
    return 0;
}
