// Does __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4 with scale operands 0 (which the compiler lowers to the UNSCALED, single
// v_mfma_f32_16x16x128_f8f6f4) equal unit E8M0 scales (127)?  One product of ones: every element must be 128.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float *o)
{
    v8i a = {0x22222222, 0x22222222, 0x22222222, 0x22222222, 0, 0, 0, 0};   // e2m1 1.0 in every nibble
    v4f c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};
    c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, a, c0, 4, 4, 0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, a, c1, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, a, c2, 4, 4, 0, (int)0x80808080, 0, (int)0x80808080);
    if (threadIdx.x == 0) { o[0] = c0[0]; o[1] = c1[0]; o[2] = c2[0]; }
}
int main()
{
    float *d, h[3];
    hipMalloc(&d, 12);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("scale operands 0 (unscaled instruction): %g   E8M0 127: %g   E8M0 128: %g\n", h[0], h[1], h[2]);
    return 0;
}
