// prints the register layout of v_mfma_f64_16x16x4_f64 (A = lane (l&15, l>>4), B = lane (l>>4, l&15) assumed)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(double *out)
{
    const int l = threadIdx.x;
    // A[i][k] = 100*i + k (i = l&15, k = l>>4);  B[k][j] = (k == 0) ? 1 : 0 scaled by (j + 1)
    const double a = 100.0 * (l & 15) + (l >> 4);
    const double b = ((l >> 4) == 0) ? (double)((l & 15) + 1) : 0.0;
    f64x4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[l * 4 + r] = d[r];
}
int main()
{
    double *o; hipMalloc(&o, 64 * 4 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
    double h[256]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    // expected D[i][j] = A[i][0] * B[0][j] = 100 i * (j + 1)
    for (int l = 0; l < 64; l += 5)
        for (int r = 0; r < 4; r++) {
            const double v = h[l * 4 + r];
            // solve: find (i, j) with 100 i (j+1) == v
            int fi = -1, fj = -1;
            for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) if (100.0 * i * (j + 1) == v && v != 0) { fi = i; fj = j; }
            printf("lane %2d reg %d: value %8.0f -> (i=%d, j=%d)\n", l, r, v, fi, fj);
        }
    return 0;
}
