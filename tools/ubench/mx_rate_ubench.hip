// Rate of v_mfma_scale_f32_32x32x64_f8f6f4 by operand formats (tools only): does a product of an fp8 (or fp6) operand with an fp4
// one run at the fp4 rate?  Register-only stream of 8 independent MFMAs per iteration, one wave per SIMD, 256 CUs.
// cbsz / blgp: 0 = fp8 e4m3, 1 = bf8, 2 = fp6 e2m3, 3 = bf6 e3m2, 4 = fp4 e2m1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int FA, int FB>
__global__ __launch_bounds__(256) void k(int iters, float *out, int seed)
{
    v16f c[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) c[i][r] = 0.f;
    v8i a, b;
#pragma unroll
    for (int t = 0; t < 8; t++) { a[t] = (seed * (t + 1) + (int)threadIdx.x * 0x01010101) & 0x22222222; b[t] = (seed + t) & 0x22222222; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[i], FA, FB, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += c[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int FA, int FB> void go(float *out, const char *name)
{
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FA, FB>), dim3(blocks), dim3(256), 0, 0, 200, out, 3);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<FA, FB>), dim3(blocks), dim3(256), 0, 0, iters, out, 3);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 64 * 8.0 * iters * blocks * 4;
    printf("%-12s %8.1f TFLOP/s  (%.1f SIMD cycles per MFMA at 2.4 GHz)\n", name, flops / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (iters * 8.0));
}

int main()
{
    float *out; (void)hipMalloc(&out, 256 * 256 * 4);
    go<0, 0>(out, "fp8 x fp8"); go<0, 4>(out, "fp8 x fp4"); go<4, 0>(out, "fp4 x fp8"); go<2, 4>(out, "fp6 x fp4");
    go<2, 2>(out, "fp6 x fp6"); go<3, 4>(out, "bf6 x fp4"); go<4, 4>(out, "fp4 x fp4");
    return 0;
}
