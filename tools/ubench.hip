// ubench.hip -- instruction-throughput microbenchmarks for the fp64 pair loop
// (development tool; not part of the product library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 4096;

template <int OP>
__global__ void __launch_bounds__(256) k(double *out, double seed, unsigned long long *clk)
{
    double a0 = seed + threadIdx.x, a1 = a0 * 1.1, a2 = a0 * 1.2, a3 = a0 * 1.3,
           a4 = a0 * 1.4, a5 = a0 * 1.5, a6 = a0 * 1.6, a7 = a0 * 1.7;
    const double b = 1.0000001, c = 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITER; ++i) {
        if (OP == 0) { // fma f64, 8 independent chains
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (OP == 1) { // mul f64
            asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                         "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (OP == 2) { // add f64
            asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                         "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == 3) { // rcp f64
            asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                         "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 4) { // cndmask b32 x8 (vcc)
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n"
                         "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n"
                         : "+v"(*(int *)&a0), "+v"(*(int *)&a1), "+v"(*(int *)&a2), "+v"(*(int *)&a3) :: "vcc");
        } else if (OP == 5) { // fma f32 x8
            float *f = (float *)&a0;
            asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                         "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(f[0]), "+v"(*(float *)&a1), "+v"(*(float *)&a2), "+v"(*(float *)&a3) : "v"(1.0001f), "v"(1e-6f));
        } else if (OP == 6) { // v_cmp_lt_f64 x8
            asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %2\n v_cmp_lt_f64 vcc, %2, %3\n v_cmp_lt_f64 vcc, %3, %0\n"
                         "v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %2\n v_cmp_lt_f64 vcc, %2, %3\n v_cmp_lt_f64 vcc, %3, %0\n"
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
        } else if (OP == 7) { // div_scale + div_fmas + div_fixup (one full IEEE division sequence parts)
            asm volatile("v_div_scale_f64 %0, vcc, %1, %1, %2\n v_div_fmas_f64 %0, %0, %1, %2\n v_div_fixup_f64 %0, %0, %1, %2\n"
                         "v_div_scale_f64 %3, vcc, %1, %1, %2\n v_div_fmas_f64 %3, %3, %1, %2\n v_div_fixup_f64 %3, %3, %1, %2\n"
                         "v_div_scale_f64 %4, vcc, %1, %1, %2\n v_div_fmas_f64 %4, %4, %1, %2\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) :: "vcc");
        } else if (OP == 8) { // ds_bpermute x8
            int idx = ((threadIdx.x + 1) & 63) * 4;
            asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         "ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(*(int *)&a0), "+v"(*(int *)&a1), "+v"(*(int *)&a2), "+v"(*(int *)&a3) : "v"(idx));
        } else if (OP == 9) { // mov dpp wave_ror:1 x8
            asm volatile("v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %2 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %2 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(*(int *)&a0), "+v"(*(int *)&a1), "+v"(*(int *)&a2), "+v"(*(int *)&a3));
        } else if (OP == 10) { // rcp f32 x8
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         : "+v"(*(float *)&a0), "+v"(*(float *)&a1), "+v"(*(float *)&a2), "+v"(*(float *)&a3));
        } else if (OP == 11) { // cvt f32<->f64 x8
            float f0, f1, f2, f3;
            asm volatile("v_cvt_f32_f64 %4, %0\n v_cvt_f32_f64 %5, %1\n v_cvt_f32_f64 %6, %2\n v_cvt_f32_f64 %7, %3\n"
                         "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3));
        } else if (OP == 12) { // dependent fma f64 chain (latency): 8 dependent
            asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                         "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(b), "v"(c));
        } else if (OP == 13) { // v_sqrt_f64
            asm volatile("v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3\n"
                         "v_sqrt_f64 %4, %4\n v_sqrt_f64 %5, %5\n v_sqrt_f64 %6, %6\n v_sqrt_f64 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 14) { // row_ror:1 dpp
            asm volatile("v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(*(int *)&a0), "+v"(*(int *)&a1), "+v"(*(int *)&a2), "+v"(*(int *)&a3));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// rcp accuracy probe + wave_ror direction probe
__global__ void probe(double *out, int *lanes)
{
    double x = 1.0 + (threadIdx.x + 1) * 0.0137;
    double r;
    asm volatile("v_rcp_f64 %0, %1" : "=v"(r) : "v"(x));
    out[threadIdx.x] = fabs(r * x - 1.0);
    float xf = (float)x, rf;
    asm volatile("v_rcp_f32 %0, %1" : "=v"(rf) : "v"(xf));
    out[64 + threadIdx.x] = fabs((double)rf * (double)xf - 1.0);
    int v = threadIdx.x, w;
    asm volatile("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(w) : "v"(v));
    lanes[threadIdx.x] = w;
    int w2 = -1;
    asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(w2) : "v"(v));
    lanes[64 + threadIdx.x] = w2;
}

template <int OP>
int run(const char *name, int wavesPerSimd, int opsPerIter)
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    int blocks = cus * wavesPerSimd;          // 256 threads = 4 waves = 1 per SIMD
    double *out; unsigned long long *clk;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8)); CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0, clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0, clk);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double ghz = (double)h[0] / ((double)h[1] * 10.0);      // memrealtime = 100 MHz
    double inst_per_simd = (double)ITER * opsPerIter * wavesPerSimd;
    double cyc = ms * 1e-3 * ghz * 1e9 / inst_per_simd;
    printf("%-28s waves/SIMD=%d  %.3f ms  clk=%.2f GHz  cycles/wave-instr=%.2f\n", name, wavesPerSimd, ms, ghz, cyc);
    hipFree(out); hipFree(clk);
    return 0;
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", w, 8); run<1>("v_mul_f64", w, 8); run<2>("v_add_f64", w, 8);
        run<3>("v_rcp_f64", w, 8); run<13>("v_sqrt_f64", w, 8); run<4>("v_cndmask_b32", w, 8); run<5>("v_fma_f32", w, 8);
        run<6>("v_cmp_lt_f64", w, 8); run<7>("div_scale/fmas/fixup (8)", w, 8);
        run<8>("ds_bpermute_b32", w, 8); run<9>("v_mov_dpp wave_ror:1", w, 8); run<14>("v_mov_dpp row_ror:1", w, 8);
        run<10>("v_rcp_f32", w, 8); run<11>("v_cvt f32<->f64", w, 8); run<12>("v_fma_f64 dependent", w, 8);
    }
    double *out; int *lanes; CHECK(hipMalloc(&out, 128 * 8)); CHECK(hipMalloc(&lanes, 128 * 4));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, lanes);
    std::vector<double> h(128); std::vector<int> l(128);
    CHECK(hipMemcpy(h.data(), out, 128 * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(l.data(), lanes, 128 * 4, hipMemcpyDeviceToHost));
    double m64 = 0, m32 = 0; for (int i = 0; i < 64; ++i) { if (h[i] > m64) m64 = h[i]; if (h[64 + i] > m32) m32 = h[64 + i]; }
    printf("max rel err v_rcp_f64 = %.3e   v_rcp_f32 = %.3e\n", m64, m32);
    printf("wave_ror:1 lane0<-%d lane1<-%d lane63<-%d lane32<-%d | row_ror:1 lane0<-%d lane1<-%d lane15<-%d lane16<-%d\n", l[0], l[1], l[63], l[32], l[64], l[65], l[79], l[80]);
    return 0;
}
