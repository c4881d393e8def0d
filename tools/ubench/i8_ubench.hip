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
template <> struct Slots<2> {            // split pass: 2*ibs0 = y.y' - x.x'
    static constexpr int NS = 2, NA = 1;
    static constexpr uint32_t ta[2] = {T_Y, T_X};
    static constexpr uint32_t tb[2] = {T_Y, T_NX};
    static constexpr int acc[2] = {0, 0};
};
template <> struct Slots<3> {            // split pass: nvalid = v.v', s.s'
    static constexpr int NS = 2, NA = 2;
    static constexpr uint32_t ta[2] = {T_V, T_S};
    static constexpr uint32_t tb[2] = {T_V, T_S};
    static constexpr int acc[2] = {0, 1};
};
template <> struct Slots<4> {            // split pass: KING v.v', h.v', v.h', h.h'
    static constexpr int NS = 4, NA = 4;
    static constexpr uint32_t ta[4] = {T_V, T_H, T_V, T_H};
    static constexpr uint32_t tb[4] = {T_V, T_V, T_H, T_H};
    static constexpr int acc[4] = {0, 1, 2, 3};
};
constexpr uint32_t Slots<2>::ta[2]; constexpr uint32_t Slots<2>::tb[2]; constexpr int Slots<2>::acc[2];
constexpr uint32_t Slots<3>::ta[2]; constexpr uint32_t Slots<3>::tb[2]; constexpr int Slots<3>::acc[2];
constexpr uint32_t Slots<4>::ta[4]; constexpr uint32_t Slots<4>::tb[4]; constexpr int Slots<4>::acc[4];
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

// software-pipelined variant (two operand register sets, next slot decoded under the current MFMAs)
#include <utility>
template <int TM, int TN, int MODE, int SG> struct Pipe {
    typedef Slots<MODE> S;
    static constexpr int NA = S::NA;
    const uint32_t *pa, *pb;
    int64_t kstride;
    uint32_t cw[TM + TN], e[TM + TN][4];
    v4i A[2][TM], B[2][TN];
    __device__ __forceinline__ void load_words()
    {
#pragma unroll
        for (int i = 0; i < TM; i++) cw[i] = pa[32 * i];
#pragma unroll
        for (int j = 0; j < TN; j++) cw[TM + j] = pb[32 * j];
        pa += kstride; pb += kstride;
    }
    __device__ __forceinline__ void extract()
    {
#pragma unroll
        for (int g = 0; g < TM + TN; g++)
#pragma unroll
            for (int u = 0; u < 4; u++) e[g][u] = (cw[g] >> (2 * u)) & 0x03030303u;
    }
    template <int SLOT, int SET> __device__ __forceinline__ void dec()
    {
#pragma unroll
        for (int i = 0; i < TM; i++) A[SET][i] = decode(S::ta[SLOT], e[i]);
#pragma unroll
        for (int j = 0; j < TN; j++) B[SET][j] = decode(S::tb[SLOT], e[TM + j]);
    }
    template <int s> __device__ __forceinline__ void phase(v16i (&c)[NA][TM][TN])
    {
        constexpr int cur = s & 1, nxt = cur ^ 1;
        constexpr bool last = (s == S::NS - 1);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
                c[S::acc[s]][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[cur][i], B[cur][j], c[S::acc[s]][i][j], 0, 0, 0);
        if (last) { extract(); load_words(); dec<0, nxt>(); }
        else dec<last ? 0 : s + 1, nxt>();
        if (SG) {
            constexpr int nv = last ? (7 + 4) * (TM + TN) : 4 * (TM + TN);
            constexpr int per = (nv + TM * TN - 1) / (TM * TN);
#pragma unroll
            for (int m = 0; m < TM * TN; m++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int... Is> __device__ __forceinline__ void kstep(v16i (&c)[NA][TM][TN], std::integer_sequence<int, Is...>)
    {
        (phase<Is>(c), ...);
    }
};

__device__ long long g_clk[2];
template <int TM, int TN, int MODE, int SG, int WPS>
__global__ __launch_bounds__(256, WPS) void kp(const uint32_t *__restrict__ W, int64_t ncols, int n_q, int n_tc,
                                               int *__restrict__ out)
{
    typedef Slots<MODE> S;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, kh = lane >> 5;
    const int st = blockIdx.x >> 6, w = blockIdx.x & 63;
    const int nsc = n_tc / 8;
    const int tr = ((st / nsc) * 8 + (w >> 3)) % n_tc, tc = (st % nsc) * 8 + (w & 7);
    v16i c[S::NA][TM][TN];
#pragma unroll
    for (int a = 0; a < S::NA; a++)
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) c[a][i][j][r] = 0;
    const long long t0 = clock64(), r0 = wall_clock64();
    Pipe<TM, TN, MODE, SG> p;
    p.pa = W + (int64_t)kh * ncols + ((int64_t)tr * (64 * TM) + wr * (32 * TM)) % (ncols - 32 * TM) + li;
    p.pb = W + (int64_t)kh * ncols + ((int64_t)tc * (64 * TN) + wc * (32 * TN)) % (ncols - 32 * TN) + li;
    p.kstride = 2 * ncols;
    p.load_words(); p.extract(); p.load_words(); p.template dec<0, 0>();
    for (int q = 0; q < n_q - 2; q++) p.kstep(c, std::make_integer_sequence<int, S::NS>{});
    const long long t1 = clock64(), r1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = t1 - t0; g_clk[1] = r1 - r0; }
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

template <int TM, int TN, int MODE, int SG, int WPS>
void gop(const char *name, const uint32_t *W, int64_t ncols, int n_q, int *out)
{
    typedef Slots<MODE> S;
    const int n_tc = 32;
    const int blocks = 2048 * 4 / (TM * TN);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((kp<TM, TN, MODE, SG, WPS>), dim3(blocks), dim3(256), 0, 0, W, ncols, n_q, n_tc, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((kp<TM, TN, MODE, SG, WPS>), dim3(blocks), dim3(256), 0, 0, W, ncols, n_q, n_tc, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double pairsnp = (double)blocks * 4 * TM * TN * 1024.0 * 32.0 * (n_q - 2);
    const double ops = pairsnp * 2 * S::NS;
    long long clk[2]; hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
    printf("%-28s TM=%d TN=%d wps=%d sg=%d  %7.3f ms  %7.1f TOPS  %.3e pair-SNP/s  shader clk %.0f MHz (%.0f cyc/kstep/wave)\n", name, TM, TN, WPS, SG, ms,
           ops / ms / 1e9, pairsnp / ms * 1e3, (double)clk[0] / ((double)clk[1] / 100.0), (double)clk[0] / (n_q - 2));
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

// ---- flush study: IBS 2x2 pipelined + counter update into a [3][n][n] uint32 panel in HBM ----
__device__ unsigned long long g_fl[4];
template <int FL>
__global__ __launch_bounds__(256, 2) void kpf(const uint32_t *__restrict__ W, int64_t ncols, int n_q, int n_t,
                                              uint32_t *__restrict__ acc, int64_t plane)
{
    typedef Slots<0> S;
    constexpr int TM = 2, TN = 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, kh = lane >> 5;
    const int st = blockIdx.x >> 4, w = blockIdx.x & 15;          // 4x4 super tiles
    const int nsc = n_t / 4;
    const int tr = (st / nsc) * 4 + (w >> 2), tc = (st % nsc) * 4 + (w & 3);
    if (tr >= n_t) return;
    v16i c[S::NA][TM][TN];
#pragma unroll
    for (int a = 0; a < S::NA; a++)
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) c[a][i][j][r] = 0;
    const int row_base = tr * 128 + wr * 64;
    const int64_t col_base = (int64_t)tc * 128 + wc * 64;
    Pipe<TM, TN, 0, 1> p;
    p.pa = W + (int64_t)kh * ncols + row_base + li;
    p.pb = W + (int64_t)kh * ncols + col_base + li;
    p.kstride = 2 * ncols;
    const long long t0 = clock64();
    p.load_words(); p.extract(); p.load_words(); p.template dec<0, 0>();
    for (int q = 0; q < n_q - 2; q++) p.kstep(c, std::make_integer_sequence<int, S::NS>{});
    const long long t1 = clock64();
    if (FL > 0) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
                uint32_t *p0 = acc + (int64_t)(row_base + 32 * i + 4 * kh) * ncols + col_base + 32 * j + li;
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    uint32_t *pg = p0 + (int64_t)(8 * rg) * ncols;
                    if (FL == 1) {
#pragma unroll
                        for (int r = 0; r < 4; r++)
#pragma unroll
                            for (int k = 0; k < 3; k++) atomicAdd(pg + (int64_t)r * ncols + k * plane, (uint32_t)c[k][i][j][4 * rg + r]);
                    } else if (FL == 4) {
#pragma unroll
                        for (int r = 0; r < 4; r++)
#pragma unroll
                            for (int k = 0; k < 3; k++) __builtin_nontemporal_store((uint32_t)c[k][i][j][4 * rg + r], pg + (int64_t)r * ncols + k * plane);
                    } else {
                        uint32_t old[3][4];
#pragma unroll
                        for (int k = 0; k < 3; k++)
#pragma unroll
                            for (int r = 0; r < 4; r++)
                                old[k][r] = FL == 2 ? __builtin_nontemporal_load(pg + (int64_t)r * ncols + k * plane) : pg[(int64_t)r * ncols + k * plane];
#pragma unroll
                        for (int r = 0; r < 4; r++)
#pragma unroll
                            for (int k = 0; k < 3; k++) {
                                const uint32_t v = old[k][r] + (uint32_t)c[k][i][j][4 * rg + r];
                                if (FL == 2) __builtin_nontemporal_store(v, pg + (int64_t)r * ncols + k * plane);
                                else pg[(int64_t)r * ncols + k * plane] = v;
                            }
                    }
                }
            }
        __builtin_amdgcn_s_waitcnt(0);
    } else {
        int sum = 0;
#pragma unroll
        for (int a = 0; a < S::NA; a++)
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) sum += c[a][i][j][r] * (a + 1);
        acc[blockIdx.x * 256 + threadIdx.x] = sum;
    }
    const long long t2 = clock64();
    if (lane == 0) {
        atomicAdd(&g_fl[0], (unsigned long long)(t1 - t0));
        atomicAdd(&g_fl[1], (unsigned long long)(t2 - t1));
        atomicAdd(&g_fl[2], 1ull);
    }
}
template <int FL> void gof(const char *name, const uint32_t *W, int64_t ncols, int n_q, uint32_t *acc)
{
    const int n_t = (int)(ncols / 128);
    const int blocks = n_t * n_t / 2 / 16 * 16;          // about the upper triangle's tile count
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((kpf<FL>), dim3(blocks), dim3(256), 0, 0, W, ncols, n_q, n_t, acc, ncols * ncols);
    hipDeviceSynchronize();
    unsigned long long z[4] = {0, 0, 0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(g_fl), z, sizeof(z));
    hipEventRecord(a);
    hipLaunchKernelGGL((kpf<FL>), dim3(blocks), dim3(256), 0, 0, W, ncols, n_q, n_t, acc, ncols * ncols);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipMemcpyFromSymbol(z, HIP_SYMBOL(g_fl), sizeof(z));
    const double pairsnp = (double)blocks * 16384.0 * 32.0 * (n_q - 2);
    printf("%-28s blocks=%d  %7.3f ms  %7.1f TOPS  loop %.0f cyc/wave, flush %.0f cyc/wave (%.1f%%)\n", name, blocks, ms,
           pairsnp * 8 / ms / 1e9, (double)z[0] / z[2], (double)z[1] / z[2], 100.0 * z[1] / (z[0] + z[1]));
}

int main()
{
    const int64_t ncols = 10240; const int K = 16384, n_d = K / 16, n_q = K / 32;
    std::vector<uint32_t> h((size_t)(n_d + 8) * ncols);
    uint64_t x = 88172645463325252ull;
    for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)x; }
    uint32_t *W; int *out;
    hipMalloc(&W, h.size() * 4); hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 8192 * 256 * 4);
    {
        uint32_t *acc; hipMalloc(&acc, (size_t)3 * ncols * ncols * 4); hipMemset(acc, 0, (size_t)3 * ncols * ncols * 4);
        for (int rep = 0; rep < 4; rep++) {
            gof<0>("IBS no flush", W, ncols, n_q, acc);
            gof<1>("IBS atomic flush", W, ncols, n_q, acc);
            gof<2>("IBS nt load/store flush", W, ncols, n_q, acc);
            gof<3>("IBS plain load/store flush", W, ncols, n_q, acc);
            gof<4>("IBS store-only flush", W, ncols, n_q, acc);
        }
        hipFree(acc);
    }
    gop<2, 2, 0, 1, 2>("IBS pipelined", W, ncols, n_q, out);
    gop<4, 4, 2, 1, 1>("pass y.y-x.x", W, ncols, n_q, out);
    gop<4, 4, 2, 0, 1>("pass y.y-x.x", W, ncols, n_q, out);
    gop<4, 2, 2, 1, 2>("pass y.y-x.x", W, ncols, n_q, out);
    gop<4, 2, 3, 1, 1>("pass v.v,s.s", W, ncols, n_q, out);
    gop<2, 2, 3, 1, 2>("pass v.v,s.s", W, ncols, n_q, out);
    gop<2, 2, 4, 1, 1>("pass KING v,h", W, ncols, n_q, out);
    gop<2, 2, 4, 0, 1>("pass KING v,h", W, ncols, n_q, out);
    gop<1, 2, 1, 1, 2>("KING pipelined", W, ncols, n_q, out);
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
