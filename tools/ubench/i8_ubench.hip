// Feasibility of the IBS / KING pair counters as exact int8 MFMA contractions (tools only).
//   counts are sums over SNPs of products of per-genotype indicator-like int8 values:
//     v = called, h = het, y = hom, s = v - 2h, x = [g==0] - [g==2]
//   IBS : nvalid = v.v'   ibs1 = (v.v' - s.s')/2   ibs0 = (y.y' - x.x')/2      4 slots, 3 accumulators
//   KING: nLoci = v.v', N1 = h.v', N2 = v.h', h.h', (y.y' - x.x')              6 slots, 5 accumulators
// Operands are decoded in registers from sample-major 2-bit words: (w >> 2u) & 0x03030303 gives four
// clean codes per dword, v_perm_b32 with a 4-byte table maps codes to int8 values.
//   DEC 0: operands decoded once (pure MFMA stream)   1: words loaded + decoded every k-step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define T_V 0x00010101u
#define T_H 0x00000100u
#define T_S 0x0001FF01u
#define T_Y 0x00010001u
#define T_X 0x00FF0001u
#define T_NX 0x000100FFu

template <int MODE> struct Slots;
template <> struct Slots<0> {
    static constexpr int NS = 4, NA = 3;
    static constexpr uint32_t ta[4] = {T_V, T_S, T_Y, T_X};
    static constexpr uint32_t tb[4] = {T_V, T_S, T_Y, T_NX};
    static constexpr int acc[4] = {0, 1, 2, 2};
};
template <> struct Slots<1> {
    static constexpr int NS = 6, NA = 5;
    static constexpr uint32_t ta[6] = {T_V, T_H, T_V, T_H, T_Y, T_X};
    static constexpr uint32_t tb[6] = {T_V, T_V, T_H, T_H, T_Y, T_NX};
    static constexpr int acc[6] = {0, 1, 2, 3, 4, 4};
};
constexpr uint32_t Slots<0>::ta[4]; constexpr uint32_t Slots<0>::tb[4]; constexpr int Slots<0>::acc[4];
constexpr uint32_t Slots<1>::ta[6]; constexpr uint32_t Slots<1>::tb[6]; constexpr int Slots<1>::acc[6];

__device__ __forceinline__ v4i decode(uint32_t tbl, const uint32_t *e)
{
    v4i r;
    r[0] = (int)__builtin_amdgcn_perm(0u, tbl, e[0]);
    r[1] = (int)__builtin_amdgcn_perm(0u, tbl, e[1]);
    r[2] = (int)__builtin_amdgcn_perm(0u, tbl, e[2]);
    r[3] = (int)__builtin_amdgcn_perm(0u, tbl, e[3]);
    return r;
}

template <int TM, int TN, int MODE, int DEC, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const uint32_t *__restrict__ W, int64_t ncols, int n_q, int n_tc,
                                              int *__restrict__ out)
{
    typedef Slots<MODE> S;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, kh = lane >> 5;
    // 8x8 super-tiles of workgroup tiles
    const int st = blockIdx.x >> 6, w = blockIdx.x & 63;
    const int nsc = n_tc / 8;
    const int tr = ((st / nsc) * 8 + (w >> 3)) % n_tc, tc = (st % nsc) * 8 + (w & 7);
    const uint32_t *__restrict__ pa = W + (int64_t)kh * ncols + (int64_t)tr * (64 * TM) + wr * (32 * TM) + li;
    const uint32_t *__restrict__ pb = W + (int64_t)kh * ncols + (int64_t)tc * (64 * TN) + wc * (32 * TN) + li;
    v16i c[S::NA][TM][TN];
#pragma unroll
    for (int a = 0; a < S::NA; a++)
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) c[a][i][j][r] = 0;
    uint32_t cw[TM + TN];
#pragma unroll
    for (int i = 0; i < TM; i++) cw[i] = pa[32 * i];
#pragma unroll
    for (int j = 0; j < TN; j++) cw[TM + j] = pb[32 * j];
    for (int q = 0; q < n_q; q++) {
        uint32_t e[TM + TN][4];
#pragma unroll
        for (int g = 0; g < TM + TN; g++)
#pragma unroll
            for (int u = 0; u < 4; u++) e[g][u] = (cw[g] >> (2 * u)) & 0x03030303u;
        if (DEC) {
            const int qn = (q + 1 < n_q) ? q + 1 : q;
#pragma unroll
            for (int i = 0; i < TM; i++) cw[i] = pa[(int64_t)2 * qn * ncols + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; j++) cw[TM + j] = pb[(int64_t)2 * qn * ncols + 32 * j];
        }
#pragma unroll
        for (int s = 0; s < S::NS; s++) {
            v4i A[TM], B[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) A[i] = decode(S::ta[s], e[i]);
#pragma unroll
            for (int j = 0; j < TN; j++) B[j] = decode(S::tb[s], e[TM + j]);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    c[S::acc[s]][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[i], B[j], c[S::acc[s]][i][j], 0, 0, 0);
        }
    }
    int sum = 0;
#pragma unroll
    for (int a = 0; a < S::NA; a++)
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) sum += c[a][i][j][r] * (a + 1);
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int TM, int TN, int MODE, int DEC, int WPS>
void go(const char *name, const uint32_t *W, int64_t ncols, int n_q, int *out)
{
    typedef Slots<MODE> S;
    const int n_tc = (int)(ncols / (64 * (TM > TN ? TM : TN)));
    const int blocks = 2048;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<TM, TN, MODE, DEC, WPS>), dim3(blocks), dim3(256), 0, 0, W, ncols, n_q, n_tc, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<TM, TN, MODE, DEC, WPS>), dim3(blocks), dim3(256), 0, 0, W, ncols, n_q, n_tc, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double pairsnp = (double)blocks * 4 * TM * TN * 1024.0 * 32.0 * n_q;
    const double ops = pairsnp * 2 * S::NS;
    printf("%-40s TM=%d TN=%d wps=%d  %7.3f ms  %7.1f TOPS  %.3e pair-SNP/s\n", name, TM, TN, WPS, ms, ops / ms / 1e9,
           pairsnp / ms * 1e3);
}

int main()
{
    const int64_t ncols = 10240; const int K = 16384, n_d = K / 16, n_q = K / 32;
    std::vector<uint32_t> h((size_t)n_d * ncols);
    uint64_t x = 88172645463325252ull;
    for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)x; }
    uint32_t *W; int *out;
    hipMalloc(&W, h.size() * 4); hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 2048 * 256 * 4);
    go<2, 2, 0, 0, 2>("IBS pure MFMA", W, ncols, n_q, out);
    go<2, 2, 0, 1, 2>("IBS loads+decode", W, ncols, n_q, out);
    go<2, 2, 0, 1, 1>("IBS loads+decode", W, ncols, n_q, out);
    go<1, 2, 0, 1, 2>("IBS loads+decode", W, ncols, n_q, out);
    go<1, 2, 0, 1, 3>("IBS loads+decode", W, ncols, n_q, out);
    go<1, 2, 1, 0, 2>("KING pure MFMA", W, ncols, n_q, out);
    go<1, 2, 1, 1, 2>("KING loads+decode", W, ncols, n_q, out);
    go<2, 2, 1, 1, 1>("KING loads+decode", W, ncols, n_q, out);
    go<1, 1, 1, 1, 4>("KING loads+decode", W, ncols, n_q, out);
    return 0;
}
