// Do VALU ops hide behind v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 x fp4)?  (tools only)
// Loop body: 4 independent MFMAs + NV independent v_and_b32, one wave per SIMD; SIMD cycles per iteration at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NV, int SCALED>
__global__ __launch_bounds__(256) void k(int iters, float *out, uint32_t seed)
{
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    v8i a, b;
#pragma unroll
    for (int t = 0; t < 8; t++) { a[t] = (int)((seed * (t + 1) + threadIdx.x * 0x01010101u) & 0x33333333u); b[t] = (int)((seed + t) & 0x33333333u); }
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = seed + threadIdx.x * (i + 1);
    const int sc = SCALED ? (int)0x80808080 : 0x7F7F7F7F;
    for (int it = 0; it < iters; it++) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, sc, 0, sc);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, sc, 0, sc);
        c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 4, 0, sc, 0, sc);
        c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 4, 4, 0, sc, 0, sc);
#pragma unroll
        for (int v = 0; v < NV; v++) asm volatile("v_and_b32 %0, 0x3030303, %0" : "+v"(x[v & 7]));
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r] + c3[r];
#pragma unroll
    for (int i = 0; i < 8; i++) s += (float)x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int SCALED> void go(float *out)
{
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, SCALED>), dim3(blocks), dim3(256), 0, 0, 100, out, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, SCALED>), dim3(blocks), dim3(256), 0, 0, iters, out, 1u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("scale %s NV=%2d  %7.1f cycles per (4 MFMA + NV VALU)\n", SCALED ? "2  " : "1  ", NV, ms * 1e-3 * 2.4e9 / iters);
}

int main()
{
    float *out; (void)hipMalloc(&out, 256 * 256 * 4);
    go<0, 0>(out); go<8, 0>(out); go<16, 0>(out); go<24, 0>(out); go<32, 0>(out); go<48, 0>(out);
    go<0, 1>(out); go<16, 1>(out); go<32, 1>(out);
    return 0;
}
