// Round 6: what v_cvt_scalef32_pk_f16_fp4 computes (tools only): every byte value through byte selects 0..3 with scales 1, 3, 0.75, 2^-3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(float *out, float s)
{
    const uint32_t b = threadIdx.x, w = b | (b << 8) | (b << 16) | (b << 24);
    const h2 a = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, s, 0), c = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, s, 3);
    out[4 * b] = (float)a[0]; out[4 * b + 1] = (float)a[1]; out[4 * b + 2] = (float)c[0]; out[4 * b + 3] = (float)c[1];
}
__global__ void ksel(float *out)
{
    const uint32_t w = 0x42210400u;      // bytes 0..3 = 0x00, 0x04 (2, 0), 0x21 (0.5, 1), 0x42 (1, 2)
    typedef _Float16 hh __attribute__((ext_vector_type(2)));
    const hh a = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 0), b = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 1),
             c = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 2), d = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 3);
    out[0] = (float)a[0]; out[1] = (float)a[1]; out[2] = (float)b[0]; out[3] = (float)b[1];
    out[4] = (float)c[0]; out[5] = (float)c[1]; out[6] = (float)d[0]; out[7] = (float)d[1];
}
int main()
{
    {
        float *d8, h8[8];
        hipMalloc(&d8, sizeof(h8));
        hipLaunchKernelGGL(ksel, dim3(1), dim3(1), 0, 0, d8);
        hipMemcpy(h8, d8, sizeof(h8), hipMemcpyDeviceToHost);
        printf("word 0x42210400 (bytes (0,0) (2,0) (0.5,1) (1,2)): builtin byte select 0 -> (%g, %g), 1 -> (%g, %g), 2 -> (%g, %g), 3 -> (%g, %g)\n",
               h8[0], h8[1], h8[2], h8[3], h8[4], h8[5], h8[6], h8[7]);
    }
    float *d, h[1024];
    hipMalloc(&d, sizeof(h));
    const float e2m1[8] = {0, 0.5f, 1, 1.5f, 2, 3, 4, 6};
    for (float s : {1.0f, 3.0f, 0.75f, 0.125f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, s);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad_full = 0, bad_pow2 = 0, sel = 0;
        for (int b = 0; b < 256; b++) {
            const float lo = e2m1[b & 7] * ((b & 8) ? -1 : 1), hi = e2m1[(b >> 4) & 7] * ((b & 128) ? -1 : 1);
            int ex; frexpf(s, &ex); const float p2 = ldexpf(1.f, ex - 1);
            bad_full += (h[4 * b] != lo * s) + (h[4 * b + 1] != hi * s);
            bad_pow2 += (h[4 * b] != lo * p2) + (h[4 * b + 1] != hi * p2);
            sel += (h[4 * b] != h[4 * b + 2]) + (h[4 * b + 1] != h[4 * b + 3]);
        }
        printf("scale %g: byte 0x21 -> (%g, %g); mismatches if the full scale multiplies %d, if only its power of two %d; byte selects 0 / 3 differ %d\n",
               s, h[4 * 0x21], h[4 * 0x21 + 1], bad_full, bad_pow2, sel);
    }
    return 0;
}
