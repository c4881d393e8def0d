// The two-product counter kernel of blocks without missing calls (g.g' from the code bytes, h.h' through a table) on
// int8 MFMAs (32 SNPs per instruction) and on fp4 MFMAs (64 SNPs per instruction, operands packed as nibbles), plain loops,
// one wave per SIMD, 128 x 64 per wave: what the nibble packing costs against what the doubled K buys (tools only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int TM = 4, TN = 2, R = TM + TN;
#define T_H8 0x00000100u      /* int8: het -> 1 */
#define T_H4 0x00000200u      /* fp4 nibble in a byte: het -> 0x2 = 1.0 */

template <int FP4, int VAR>
__global__ __launch_bounds__(256, 1) void k(const uint32_t *__restrict__ W, int64_t ncols, int n_q /* 32-SNP steps */, int n_tc,
                                            float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, kh = lane >> 5;
    const int tr = (blockIdx.x / n_tc) % n_tc, tc = blockIdx.x % n_tc;
    const uint32_t *__restrict__ pa = W + (int64_t)kh * ncols + (int64_t)tr * (64 * TM) + wr * (32 * TM) + li;
    const uint32_t *__restrict__ pb = W + (int64_t)kh * ncols + (int64_t)tc * (64 * TN) + wc * (32 * TN) + li;
    v16i ci[2][TM][TN];
    v16f cf[2][TM][TN];
    for (int a = 0; a < 2; a++)
        for (int i = 0; i < TM; i++)
            for (int j = 0; j < TN; j++)
                for (int r = 0; r < 16; r++) { ci[a][i][j][r] = 0; cf[a][i][j][r] = 0.f; }
    if (!FP4) {
        for (int q = 0; q < n_q; q++) {
            uint32_t cw[R], e[R][4];
#pragma unroll
            for (int g = 0; g < R; g++) cw[g] = (g < TM ? pa : pb)[(int64_t)2 * q * ncols + 32 * (g < TM ? g : g - TM)];
#pragma unroll
            for (int g = 0; g < R; g++)
#pragma unroll
                for (int u = 0; u < 4; u++) e[g][u] = (cw[g] >> (2 * u)) & 0x03030303u;
            v4i G[R], H[R];
#pragma unroll
            for (int g = 0; g < R; g++)
#pragma unroll
                for (int u = 0; u < 4; u++) { G[g][u] = (int)e[g][u]; H[g][u] = (int)__builtin_amdgcn_perm(0u, T_H8, e[g][u]); }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    ci[0][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(G[i], G[TM + j], ci[0][i][j], 0, 0, 0);
                    ci[1][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(H[i], H[TM + j], ci[1][i][j], 0, 0, 0);
                }
        }
    } else {
        for (int q = 0; q < n_q; q += 2) {                 // 64 SNPs: two words per lane and sample group
            v8i G[R], H[R];
#pragma unroll
            for (int g = 0; g < R; g++) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t w = (VAR == 1) ? (uint32_t)(q * 2654435761u + g * 97u + h + lane) : (g < TM ? pa : pb)[(int64_t)2 * (q + h) * ncols + 32 * (g < TM ? g : g - TM)];
                    uint32_t e[4], p[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { e[u] = (w >> (2 * u)) & 0x03030303u; p[u] = __builtin_amdgcn_perm(0u, T_H4, e[u]); }
                    // nibble = 2 * code: 0, 1, 2 -> E2M1 0.0, 1.0, 2.0
                    if (VAR == 2) { G[g][2 * h] = (int)w; G[g][2 * h + 1] = (int)(w >> 1); H[g][2 * h] = (int)(w >> 2); H[g][2 * h + 1] = (int)(w >> 3); continue; }
                    G[g][2 * h] = (int)((e[0] | (e[1] << 4)) << 1);
                    G[g][2 * h + 1] = (int)((e[2] | (e[3] << 4)) << 1);
                    H[g][2 * h] = (int)(p[0] | (p[1] << 4));
                    H[g][2 * h + 1] = (int)(p[2] | (p[3] << 4));
                }
#pragma unroll
                for (int t = 4; t < 8; t++) { G[g][t] = 0; H[g][t] = 0; }
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    cf[0][i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(G[i], G[TM + j], cf[0][i][j], 4, 4, 0, 127, 0, 127);
                    cf[1][i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(H[i], H[TM + j], cf[1][i][j], 4, 4, 0, 127, 0, 127);
                }
        }
    }
    float s = 0;
    for (int a = 0; a < 2; a++)
        for (int i = 0; i < TM; i++)
            for (int j = 0; j < TN; j++)
                for (int r = 0; r < 16; r++) s += FP4 ? cf[a][i][j][r] * (a + 1) : (float)ci[a][i][j][r] * (a + 1);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    const int n = 10240, L = 65536, n_d = L / 16, n_q = L / 32;
    const int n_tc = n / (64 * TM);                    // square grid of 256-row tiles (columns reuse the row range)
    std::vector<uint32_t> w((size_t)n_d * n);
    srand(3);
    for (auto &x : w) { uint32_t v = 0; for (int t = 0; t < 16; t++) v |= (uint32_t)(rand() % 3) << (2 * t); x = v; }
    uint32_t *dw; float *out;
    hipMalloc(&dw, w.size() * 4 + (1 << 20)); hipMalloc(&out, (size_t)n_tc * n_tc * 256 * 4);
    hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> o0((size_t)n_tc * n_tc * 256), o1(o0.size());
    for (int fp4 = 0; fp4 < 4; fp4++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0, 0);
            if (fp4 == 1) hipLaunchKernelGGL((k<1, 0>), dim3(n_tc * n_tc), dim3(256), 0, 0, dw, (int64_t)n, n_q, n_tc, out);
            else if (fp4 == 2) hipLaunchKernelGGL((k<1, 1>), dim3(n_tc * n_tc), dim3(256), 0, 0, dw, (int64_t)n, n_q, n_tc, out);
            else if (fp4 == 3) hipLaunchKernelGGL((k<1, 2>), dim3(n_tc * n_tc), dim3(256), 0, 0, dw, (int64_t)n, n_q, n_tc, out);
            else hipLaunchKernelGGL((k<0, 0>), dim3(n_tc * n_tc), dim3(256), 0, 0, dw, (int64_t)n, n_q, n_tc, out);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double pairs = (double)n_tc * n_tc * (256.0 * 128.0);
            printf("%s: %.3f ms  -> %.3e pair-genotypes/s over %d x %d tiles of 256 x 128\n", fp4 == 0 ? "i8 " : fp4 == 1 ? "fp4" : fp4 == 2 ? "fp4, no loads in the loop" : "fp4, no decode", ms,
                   pairs * L / (ms * 1e-3), n_tc, n_tc);
        }
        if (fp4 < 2) hipMemcpy((fp4 ? o1 : o0).data(), out, o0.size() * 4, hipMemcpyDeviceToHost);
    }
    size_t bad = 0;
    for (size_t i = 0; i < o0.size(); i++) bad += (o0[i] != o1[i]);
    printf("checksums that differ between the two forms: %zu of %zu\n", bad, o0.size());
    return 0;
}
