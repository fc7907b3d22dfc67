// mfma4_layout.hip -- which lanes does v_mfma_f64_4x4x4_4b sum?  (development tool)
// A = 2^(lane % 16) (+ 2^20 * block), B = 1: every output lane prints the set of
// input lanes it received, as a bit mask; then the same with A = 1, B = 2^(lane % 16).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out)
{
    const int lane = threadIdx.x;
    const double code = (double)(1u << (lane % 16)) + 1048576.0 * (double)(1u << (lane / 16));
    out[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(code, 1.0, 0.0, 0, 0, 0);
    out[64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, code, 0.0, 0, 0, 0);
}
int main()
{
    double *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int t = 0; t < 2; ++t) {
        printf("%s\n", t == 0 ? "A = code, B = ones" : "A = ones, B = code");
        for (int l = 0; l < 64; ++l) {
            unsigned long long v = (unsigned long long)h[64 * t + l];
            printf("  lane %2d: lanes-in-block mask %04llx  blocks mask %llx\n", l, v & 0xffff, v >> 20);
        }
    }
    return 0;
}
