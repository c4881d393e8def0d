// Round 6: what one VALU / LDS instruction costs in the shadow of an MFMA, per instruction shape and waves per SIMD (tools only).
// Per loop step a wave issues 16 independent MFMAs, each followed by NV VALU adds (SDWA byte add, as the lookups' address op) and NL
// ds_read_b32; clocks per MFMA from the wall time at the measured rate of a zero-VALU run is what matters: printed as ns per MFMA.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NV, int NL, int WPS>
__global__ __launch_bounds__(256, WPS) void k(int iters, float *out, uint32_t seed)
{
    __shared__ uint32_t tab[4096];
    for (int e = threadIdx.x; e < 4096; e += 256) tab[e] = e * 2654435761u;
    __syncthreads();
    u4 a = u4{seed, seed * 3u, seed * 5u, seed * 7u} & 0x3BFF3BFFu, b = a ^ 0x01010101u;
    constexpr int NT = SHAPE == 0 ? (WPS == 1 ? 16 : 8) : (WPS == 1 ? 64 : 32);
    v16f c0[SHAPE == 0 ? NT : 1];
    v4f c1[SHAPE == 1 ? NT : 1];
#pragma unroll
    for (int i = 0; i < (SHAPE == 0 ? NT : 1); i++)
#pragma unroll
        for (int r = 0; r < 16; r++) c0[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < (SHAPE == 1 ? NT : 1); i++)
#pragma unroll
        for (int r = 0; r < 4; r++) c1[i][r] = 0.f;
    uint32_t v[4] = {threadIdx.x, threadIdx.x * 3u, seed, seed + threadIdx.x};
    uint32_t ld = 0;
    const uint32_t tb = (uint32_t)(uintptr_t)&tab[0] + 4 * (threadIdx.x & 63);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < NT; m++) {
            if (SHAPE == 0) c0[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16((h8)a, (h8)b, c0[m], 0, 0, 0);
            else c1[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16((h8)a, (h8)b, c1[m], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; q++)
                asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(v[q & 3]) : "v"(v[(q + 1) & 3]), "v"(v[(q + 2) & 3]));
#pragma unroll
            for (int q = 0; q < NL; q++) {
                uint32_t r_;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r_) : "v"(tb), "n"(256 * q));
                ld ^= r_;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = (float)(v[0] + v[1] + v[2] + v[3] + ld);
    if (SHAPE == 0) for (int i = 0; i < NT; i++) for (int r = 0; r < 16; r++) s += c0[i][r];
    else for (int i = 0; i < NT; i++) for (int r = 0; r < 4; r++) s += c1[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int NV, int NL, int WPS>
static void run(int iters, float *d_out)
{
    constexpr int NT = SHAPE == 0 ? (WPS == 1 ? 16 : 8) : (WPS == 1 ? 64 : 32);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, NV, NL, WPS>), dim3(256 * WPS), dim3(256), 0, 0, iters / 4, d_out, 3u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, NV, NL, WPS>), dim3(256 * WPS), dim3(256), 0, 0, iters, d_out, 3u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    // ns per MFMA-equivalent of 32 768 flops on one SIMD (a 32x32x16 = 1, a 16x16x32 = 1/2)
    const double per = best * 1e6 / ((double)iters * NT * WPS) * (SHAPE == 0 ? 1.0 : 2.0);
    printf("%s %d wave(s)/SIMD, per MFMA %d VALU + %d LDS: %.1f ns per 32768 flops per SIMD\n", SHAPE == 0 ? "32x32x16" : "16x16x32", WPS, NV, NL, per);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    float *d_out;
    hipMalloc(&d_out, 512 * 256 * 4);
    run<0, 0, 0, 1>(iters, d_out); run<0, 1, 0, 1>(iters, d_out); run<0, 2, 0, 1>(iters, d_out); run<0, 2, 2, 1>(iters, d_out); run<0, 4, 0, 1>(iters, d_out);
    run<1, 0, 0, 1>(iters, d_out); run<1, 1, 0, 1>(iters, d_out); run<1, 0, 1, 1>(iters, d_out); run<1, 1, 1, 1>(iters, d_out); run<1, 2, 0, 1>(iters, d_out);
    run<0, 0, 0, 2>(iters, d_out); run<0, 2, 2, 2>(iters, d_out);
    run<1, 0, 0, 2>(iters, d_out); run<1, 1, 1, 2>(iters, d_out); run<1, 2, 2, 2>(iters, d_out);
    return 0;
}
