// fp64 MFMA throughput of gfx950: a register-only stream of v_mfma_f64_16x16x4_f64 on independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0)
{
    f64x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (f64x4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) out[0] = s;
}
template <int NACC>
void run(int waves_per_simd)
{
    double *d;
    hipMalloc(&d, 8);
    const int iters = 20000, blocks = 256 * waves_per_simd;       // 256 threads = 4 waves = one per SIMD
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0, 1.0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001, 0.9999);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * blocks * 4;
    const double mfmas_per_simd = (double)NACC * iters * waves_per_simd;
    printf("acc=%d waves/SIMD=%d: %.1f TFLOP/s, %.1f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)\n", NACC, waves_per_simd,
           flops / ms / 1e9, ms * 1e6 / mfmas_per_simd, ms * 1e6 / mfmas_per_simd * 2.4);
    hipFree(d);
}
int main()
{
    run<4>(1); run<8>(1); run<12>(1); run<8>(2); run<12>(2);
    return 0;
}
