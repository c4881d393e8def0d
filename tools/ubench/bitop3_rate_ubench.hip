// Issue rate of v_bitop3_b32 against v_and_b32 / v_and_or_b32 / v_lshl_or_b32 / v_xor3? (tools only): 256 independent-ish ops per
// iteration on 8 registers, one wave per SIMD; SIMD cycles per instruction at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(256) void k(int iters, uint32_t *out, uint32_t seed)
{
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = seed * (i + 3) + threadIdx.x;
    uint32_t m = seed | 0x11111111u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int v = 0; v < 256; v++) {
            if (KIND == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[v & 7]) : "v"(x[(v + 3) & 7]));
            else if (KIND == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x40" : "+v"(x[v & 7]) : "v"(x[(v + 3) & 7]), "s"(m));
            else if (KIND == 2) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(x[(v + 3) & 7]), "s"(m));
            else if (KIND == 3) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x[v & 7]) : "v"(x[(v + 3) & 7]));
            else if (KIND == 4) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x40" : "+v"(x[v & 7]) : "v"(x[(v + 3) & 7]), "v"(x[(v + 5) & 7]));
            else if (KIND == 5) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x[v & 7]));
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND> void go(uint32_t *out, const char *name)
{
    const int iters = 2000;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256), 0, 0, 10, out, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256), 0, 0, iters, out, 1u);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-34s %5.2f cycles per instruction (one wave per SIMD)\n", name, ms * 1e-3 * 2.4e9 / (iters * 256.0));
}
int main()
{
    uint32_t *out; (void)hipMalloc(&out, 256 * 256 * 4);
    go<0>(out, "v_and_b32"); go<1>(out, "v_bitop3_b32 (v, v, s)"); go<4>(out, "v_bitop3_b32 (v, v, v)"); go<2>(out, "v_and_or_b32 (v, v, s)");
    go<3>(out, "v_lshl_or_b32"); go<5>(out, "v_lshrrev_b32");
    return 0;
}
