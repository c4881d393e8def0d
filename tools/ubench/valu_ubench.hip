// VALU integer throughput micro-benchmark (tools only): what is the real lane-op/s ceiling for
// the bit-op mix of the pair_popcount kernel on MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a[16], s = seed ^ threadIdx.x;
    for (int i = 0; i < 16; i++) a[i] = s * (i + 3);
    uint32_t x = s * 7, y = s * 13;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == 0) a[i] = a[i] & (x + i);                       // v_and (plus const add folded?)
                else if (OP == 1) a[i] = __popc(a[i] ^ x) + a[i];          // xor + bcnt-accumulate
                else if (OP == 2) a[i] += __popc(x & (y + i));             // and + bcnt
                else if (OP == 3) a[i] = (a[i] & x) | y;                   // and_or
                else if (OP == 4) a[i] = a[i] * x + y;                     // mad (for reference)
            }
            x = x * 3 + 1; y = y * 5 + 1;
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 16; i++) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP>
void run(const char *name, int ops_per_elem)
{
    uint32_t *out; hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 123u, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 123u, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * 256 * iters * 8 * 16 * ops_per_elem;
    printf("%-28s %8.3f ms  %7.2f Tlane-op/s\n", name, ms, ops / ms / 1e9);
    hipFree(out);
}
int main()
{
    run<0>("v_and (1 op)", 1);
    run<1>("xor + bcnt_acc (2 ops)", 2);
    run<2>("and + bcnt_acc (2 ops)", 2);
    run<3>("and_or (1 op)", 1);
    run<4>("mad_u32 (1 op)", 1);
    return 0;
}
