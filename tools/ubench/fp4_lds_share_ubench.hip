// Would sharing the operand decode of the general fp4 counter kernels between the waves of a workgroup pay?  (tools only)
//
// pair_mfma_fp4_kernel<PM_IBS>: 2 x 2 waves per workgroup, each wave 64 x 64 = 2 + 2 sample groups x 4 value types; per k-step (64
// SNPs) a wave issues 16 MFMAs (v_mfma_scale_f32_32x32x64_f8f6f4, fp4 x fp4) and ~112 VALU of decode (14 per 16-SNP unit, 8 units) --
// and every row / column group is decoded by TWO waves.  The alternative: each wave decodes 2 of the workgroup's 8 groups (56 VALU),
// keeps them, writes them to LDS (2 groups x 4 types x 4 dwords = 8 ds_write_b128 per lane) and reads the 2 groups it lacks (8
// ds_read_b128), double-buffered, one s_barrier per k-step.
//
// This models exactly that instruction mix per k-step, one workgroup of 4 waves per CU, 256 CUs:
//   A  16 MFMA + 112 VALU                                   (today's k-step)
//   B  16 MFMA +  56 VALU + 8 ds_write_b128 + 8 ds_read_b128 + s_barrier   (shared decode)
//   C  16 MFMA +  56 VALU                                   (what B would cost if LDS and the barrier were free)
//   D  16 MFMA                                              (the matrix pipe alone)
// The VALU ops are v_bitop3-class ops whose results feed the next k-step's MFMA operands (nothing can be dropped); the LDS
// addresses follow the real layout (lane l writes 16 bytes at 16 l of its group's slab: conflict-free).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NV, int LDS>
__global__ __launch_bounds__(256, 1) void k(int iters, float *out, uint32_t seed)
{
    __shared__ v4i slab[2][8][4][64];                 // [buffer][group][type][lane]: 64 KiB
    v16f c[16];
#pragma unroll
    for (int i = 0; i < 16; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) c[i][r] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = (seed * (i + 3) + threadIdx.x * 0x9E3779B1u) & 0x11111111u;
    v4i op[4][4];                                      // 4 groups (2 row + 2 column) x 4 types, 4 dwords each
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
        for (int t = 0; t < 4; t++) op[g][t] = v4i{(int)x[g], (int)x[t + 4], (int)x[g + 8], (int)x[t + 12]};
    for (int it = 0; it < iters; it++) {
        const int buf = it & 1;
        // 16 MFMAs: 4 products x (2 x 2 tiles)
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const v8i a = __builtin_shufflevector(op[i][p], op[i][p], 0, 1, 2, 3, -1, -1, -1, -1);
                    const v8i b = __builtin_shufflevector(op[2 + j][p], op[2 + j][p], 0, 1, 2, 3, -1, -1, -1, -1);
                    c[p * 4 + i * 2 + j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[p * 4 + i * 2 + j], 4, 4, 0, (int)0x80808080,
                                                                                          0, (int)0x80808080);
                }
        // decode of the NEXT k-step: NV bit operations whose results become operand dwords
#pragma unroll
        for (int v = 0; v < NV; v++)
            asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(x[v & 15]) : "v"(x[(v + 5) & 15]), "v"(x[(v + 11) & 15]));
        constexpr int NG = LDS ? 2 : 4;                // groups this wave decodes itself
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int t = 0; t < 4; t++) op[g][t] = v4i{(int)x[(4 * g + t) & 15], (int)x[(4 * g + t + 1) & 15], (int)x[(4 * g + t + 2) & 15], (int)x[(4 * g + t + 3) & 15]};
        if (LDS) {
#pragma unroll
            for (int g = 0; g < 2; g++)
#pragma unroll
                for (int t = 0; t < 4; t++) slab[buf][2 * wave + g][t][lane] = op[g][t];
            __syncthreads();
            const int pw = wave ^ 1, qw = wave ^ 2;    // the partner waves holding the other row group / column group
#pragma unroll
            for (int t = 0; t < 4; t++) { op[2][t] = slab[buf][2 * pw][t][lane]; op[3][t] = slab[buf][2 * qw + 1][t][lane]; }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += c[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)x[0];
}

template <int NV, int LDS> void go(float *out, const char *name)
{
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, LDS>), dim3(blocks), dim3(256), 0, 0, 200, out, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, LDS>), dim3(blocks), dim3(256), 0, 0, iters, out, 1u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 64 * 16.0 * iters * blocks * 4;
    printf("%-58s %7.1f us per 1000 k-steps  %7.1f TFLOP/s  (%.0f%% of the 9099 measured fp4 peak)\n", name, ms * 1e3 / iters * 1000,
           flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 9099 * 100);
}

int main()
{
    float *out; (void)hipMalloc(&out, 256 * 256 * 4);
    go<112, 0>(out, "A  16 MFMA + 112 VALU (today)");
    go<56, 1>(out, "B  16 MFMA + 56 VALU + 8 ds_write_b128 + 8 ds_read_b128 + barrier");
    go<56, 0>(out, "C  16 MFMA + 56 VALU (B without LDS / barrier)");
    go<0, 0>(out, "D  16 MFMA");
    return 0;
}
