// ubench3.hip -- does fp64 MFMA run beside the vector ALU on gfx950?
// (development tool; numbers in profiles/r02_ubench3_mfma_overlap.txt)
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 2048;
typedef double v4d __attribute__((ext_vector_type(4)));

#define FMA8 asm volatile(                                                          \
    "v_fma_f64 %0, %8, %9, %8\n v_fma_f64 %1, %8, %9, %8\n v_fma_f64 %2, %8, %9, %8\n" \
    "v_fma_f64 %3, %8, %9, %8\n v_fma_f64 %4, %8, %9, %8\n v_fma_f64 %5, %8, %9, %8\n" \
    "v_fma_f64 %6, %8, %9, %8\n v_fma_f64 %7, %8, %9, %8\n"                         \
    : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7) \
    : "v"(s0), "v"(s1));
#define MFMA2 asm volatile(                                                         \
    "v_mfma_f64_16x16x4_f64 %0, %2, %3, 0\n v_mfma_f64_16x16x4_f64 %1, %2, %3, 0\n" \
    : "=&v"(m0), "=&v"(m1) : "v"(s0), "v"(s1));
#define MFMA4X4 asm volatile(                                                       \
    "v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, 0\n v_mfma_f64_4x4x4_4b_f64 %1, %2, %3, 0\n" \
    : "=&v"(q0), "=&v"(q1) : "v"(s0), "v"(s1));

#define KERNEL(NAME, BODY)                                                      \
__global__ void __launch_bounds__(256) NAME(double *out, double seed,           \
                                            unsigned long long *clk)            \
{                                                                               \
    double s0 = seed + threadIdx.x, s1 = s0 * 1.1;                              \
    double d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;      \
    v4d m0 = {0, 0, 0, 0}, m1 = {0, 0, 0, 0};                                   \
    double q0 = 0, q1 = 0;                                                      \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                       \
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                   \
    for (int i = 0; i < ITER; ++i) { BODY }                                     \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                       \
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 +  \
        d6 + d7 + m0[0] + m0[3] + m1[1] + m1[2] + q0 + q1;                      \
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; } \
}

KERNEL(k_fma8, FMA8)
KERNEL(k_mfma2, MFMA2)
KERNEL(k_fma8_mfma2, FMA8 MFMA2)
KERNEL(k_fma16_mfma2, FMA8 FMA8 MFMA2)
KERNEL(k_mfma4x4_2, MFMA4X4)
KERNEL(k_fma8_mfma4x4_2, FMA8 MFMA4X4)

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
    printf("%-22s waves/SIMD=%d  %.3f ms  clk=%.2f GHz  SIMD cycles per wave-iteration=%.1f\n",
           name, wavesPerSimd, ms, ghz, cyc);
    hipFree(out); hipFree(clk);
    return 0;
}

int main()
{
    for (int w : {1, 4, 8}) {
#define R(K) run(K, #K, w);
        R(k_fma8) R(k_mfma2) R(k_fma8_mfma2) R(k_fma16_mfma2) R(k_mfma4x4_2) R(k_fma8_mfma4x4_2)
    }
    return 0;
}
