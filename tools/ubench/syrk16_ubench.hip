// Feasibility of the GRM SYRK on fp16 MFMAs with split operands (tools only).
//   z = hi + lo, hi = fp16(z), lo = fp16(z - hi)  (22 significant bits);  z z' ~ hi hi' + hi lo' + lo hi'
//   (products of two fp16 are exact in fp32; only lo lo' ~ 2^-22 z z' is dropped), fp32 accumulate.
// Three v_mfma_f32_32x32x16_f16 replace eight v_mfma_f32_32x32x2_f32 per 16 SNPs: 96 instead of 512
// matrix-pipe cycles.  Operand decode as in syrk_mfma_kernel: pair-coded words, one v_add_u32_sdwa + one
// ds_read_b64 per SNP pair; the 8-byte table entry is {hi0, hi1, lo0, lo1} (fp16), so the four lookups of a
// lane give its 8-SNP operand registers directly (hi: dword 0 of each, lo: dword 1).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

constexpr int LUTCH = 256;                 // SNPs per LDS table chunk
__device__ long long g_clk[2];

template <int TM, int TN, int WPS, int NPROD, int PIPE>
__global__ __launch_bounds__(256, WPS) void k(const uint32_t *__restrict__ w8, int64_t ncols, const uint2 *__restrict__ lut,
                                              int n_q, int n_t, float *__restrict__ out)
{
    constexpr int CHE = (LUTCH / 2) * 16;          // uint2 entries per chunk
    __shared__ uint2 slut[2][CHE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, kh = lane >> 5;
    const int st = blockIdx.x >> 4, w = blockIdx.x & 15;
    const int nsc = n_t / 4;
    const int tr = ((st / nsc) * 4 + (w >> 2)) % n_t, tc = (st % nsc) * 4 + (w & 3);
    const uint32_t *__restrict__ pa = w8 + (int64_t)kh * ncols + ((int64_t)tr * 64 * TM + wr * 32 * TM) % (ncols - 32 * TM) + li;
    const uint32_t *__restrict__ pb = w8 + (int64_t)kh * ncols + ((int64_t)tc * 64 * TN + wc * 32 * TN) % (ncols - 32 * TN) + li;
    f16v c[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) c[i][j][r] = 0.f;
    constexpr int QCH = LUTCH / 16;
    const int n_chunk = n_q / QCH;
    for (int e = tid; e < CHE; e += 256) slut[0][e] = lut[e];
    uint32_t wa[TM], wb[TN];
#pragma unroll
    for (int i = 0; i < TM; i++) wa[i] = pa[32 * i];
#pragma unroll
    for (int j = 0; j < TN; j++) wb[j] = pb[32 * j];
    __syncthreads();
    const long long t0 = clock64(), r0 = wall_clock64();
    // operand registers of the current 16-SNP group (decoded one group ahead)
    u4 Ah[2][TM], Al[2][TM], Bh[2][TN], Bl[2][TN];
    auto decode = [&](int set, const char *tb, const uint32_t *a, const uint32_t *b) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const uint2 t = *reinterpret_cast<const uint2 *>(tb + ((a[i] >> (8 * p)) & 0xFFu) + 128 * p);
                Ah[set][i][p] = t.x; Al[set][i][p] = t.y;
            }
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const uint2 t = *reinterpret_cast<const uint2 *>(tb + ((b[j] >> (8 * p)) & 0xFFu) + 128 * p);
                Bh[set][j][p] = t.x; Bl[set][j][p] = t.y;
            }
        }
    };
    auto mfmas = [&](int set) {     // product-major: dependent MFMAs (same accumulator) are TM*TN apart
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
                c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16((h8)Ah[set][i], (h8)Bh[set][j], c[i][j], 0, 0, 0);
        if (NPROD >= 2) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16((h8)Ah[set][i], (h8)Bl[set][j], c[i][j], 0, 0, 0);
        }
        if (NPROD >= 3) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16((h8)Al[set][i], (h8)Bh[set][j], c[i][j], 0, 0, 0);
        }
    };
    for (int ch = 0; ch < n_chunk; ch++) {
        const int cur = ch & 1;
        const char *tb = reinterpret_cast<const char *>(&slut[cur][0]) + 512 * kh;
        if (!PIPE) {
            for (int q = ch * QCH; q < (ch + 1) * QCH; q++) {
                uint32_t a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; i++) a[i] = wa[i];
#pragma unroll
                for (int j = 0; j < TN; j++) b[j] = wb[j];
                {
                    const int64_t off = (int64_t)(q + 1) * 2 * ncols;
#pragma unroll
                    for (int i = 0; i < TM; i++) wa[i] = pa[off + 32 * i];
#pragma unroll
                    for (int j = 0; j < TN; j++) wb[j] = pb[off + 32 * j];
                }
                decode(0, tb, a, b);
                tb += 1024;
                mfmas(0);
            }
        } else {
            // group q's operands were decoded during group q-1 (first group of a chunk: here)
            decode(0, tb, wa, wb);
            tb += 1024;
            for (int q = ch * QCH; q < (ch + 1) * QCH; q += 2) {
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    {
                        const int64_t off = (int64_t)(q + half + 1) * 2 * ncols;
#pragma unroll
                        for (int i = 0; i < TM; i++) wa[i] = pa[off + 32 * i];
#pragma unroll
                        for (int j = 0; j < TN; j++) wb[j] = pb[off + 32 * j];
                    }
                    mfmas(half);
                    // decode the next group into the other set (the last group of a chunk decodes garbage
                    // from the stale table: harmless here, the real kernel re-decodes after the table swap)
                    __builtin_amdgcn_s_waitcnt(0x0F70 | 0);   // vmcnt(0): words of the next group
                    decode(half ^ 1, tb, wa, wb);
                    tb += 1024;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (ch + 1 < n_chunk) {
            const uint2 *__restrict__ src = lut + (int64_t)(ch + 1) * CHE;
            for (int e = tid; e < CHE; e += 256) slut[cur ^ 1][e] = src[e];
            __syncthreads();
        }
    }
    const long long t1 = clock64(), r1 = wall_clock64();
    if (blockIdx.x == 0 && tid == 0) { g_clk[0] = t1 - t0; g_clk[1] = r1 - r0; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) s += c[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int TM, int TN, int WPS, int NPROD, int PIPE>
void go(const uint32_t *w8, int64_t ncols, const uint2 *lut, int n_q, float *out)
{
    const int n_t = 32;
    const int blocks = 4096 * 4 / (TM * TN);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k<TM, TN, WPS, NPROD, PIPE>), dim3(blocks), dim3(256), 0, 0, w8, ncols, lut, n_q, n_t, out);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    }
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    long long clk[2]; (void)hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
    const double pairsnp = (double)blocks * 4 * TM * TN * 1024.0 * 16.0 * n_q;
    printf("TM=%d TN=%d wps=%d products=%d pipe=%d  %8.3f ms  %7.1f useful TFLOP/s (2 flop per pair-SNP)  %.3e pair-SNP/s  "
           "%.0f cyc per 16 SNPs per wave, clk %.0f MHz\n", TM, TN, WPS, NPROD, PIPE, ms, pairsnp * 2 / ms / 1e9,
           pairsnp / ms * 1e3, (double)clk[0] / n_q, (double)clk[0] / ((double)clk[1] / 100.0));
}

template <int TM, int TN, int SG>
__global__ __launch_bounds__(256, 1) void k1(const uint32_t *__restrict__ w8, int64_t ncols, const uint2 *__restrict__ lut,
                                             int n_q, int n_t, float *__restrict__ out)
{
    constexpr int CHE = (LUTCH / 2) * 16;
    __shared__ uint2 slut[2][CHE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, kh = lane >> 5;
    const int st = blockIdx.x >> 4, w = blockIdx.x & 15;
    const int nsc = n_t / 4;
    const int tr = ((st / nsc) * 4 + (w >> 2)) % n_t, tc = (st % nsc) * 4 + (w & 3);
    const uint32_t *__restrict__ pa = w8 + (int64_t)kh * ncols + ((int64_t)tr * 64 * TM + wr * 32 * TM) % (ncols - 32 * TM) + li;
    const uint32_t *__restrict__ pb = w8 + (int64_t)kh * ncols + ((int64_t)tc * 64 * TN + wc * 32 * TN) % (ncols - 32 * TN) + li;
    f16v c[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) c[i][j][r] = 0.f;
    constexpr int QCH = LUTCH / 16;
    const int n_chunk = n_q / QCH;
    for (int e = tid; e < CHE; e += 256) slut[0][e] = lut[e];
    uint32_t wa[2][TM], wb[2][TN];
    u4 Ah[2][TM], Al[2][TM], Bh[2][TN], Bl[2][TN];
    auto loadw = [&](int set, int q) {
        const int64_t off = (int64_t)q * 2 * ncols;
#pragma unroll
        for (int i = 0; i < TM; i++) wa[set][i] = pa[off + 32 * i];
#pragma unroll
        for (int j = 0; j < TN; j++) wb[set][j] = pb[off + 32 * j];
    };
    auto decode = [&](int set, const char *tb) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const uint2 t = *reinterpret_cast<const uint2 *>(tb + ((wa[set][i] >> (8 * p)) & 0xFFu) + 128 * p);
                Ah[set][i][p] = t.x; Al[set][i][p] = t.y;
            }
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const uint2 t = *reinterpret_cast<const uint2 *>(tb + ((wb[set][j] >> (8 * p)) & 0xFFu) + 128 * p);
                Bh[set][j][p] = t.x; Bl[set][j][p] = t.y;
            }
        }
    };
    auto mfmas = [&](int set) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
                c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16((h8)Ah[set][i], (h8)Bh[set][j], c[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
                c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16((h8)Ah[set][i], (h8)Bl[set][j], c[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
                c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16((h8)Al[set][i], (h8)Bh[set][j], c[i][j], 0, 0, 0);
    };
    loadw(0, 0); loadw(1, 1);
    __syncthreads();
    const long long t0 = clock64(), r0 = wall_clock64();
    for (int ch = 0; ch < n_chunk; ch++) {
        const int cur = ch & 1;
        const char *tb = reinterpret_cast<const char *>(&slut[cur][0]) + 512 * kh;
        decode(0, tb); tb += 1024;              // group 0 of the chunk (words in set 0)
        loadw(0, ch * QCH + 2);
        for (int q = 0; q < QCH; q += 2) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                mfmas(half);
                decode(half ^ 1, tb); tb += 1024;            // next group (its words were loaded two groups ago)
                loadw(half ^ 1, ch * QCH + q + half + 3);
                if (SG) {
                    // 3*TM*TN MFMAs, 4*(TM+TN) lookups: spread the lookups over the MFMA stream
                    constexpr int NM = 3 * TM * TN, NL = 4 * (TM + TN);
#pragma unroll
                    for (int m = 0; m < NL; m++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, NM / NL > 0 ? NM / NL : 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ch + 1 < n_chunk) {
            const uint2 *__restrict__ src = lut + (int64_t)(ch + 1) * CHE;
            for (int e = tid; e < CHE; e += 256) slut[cur ^ 1][e] = src[e];
            __syncthreads();
        }
    }
    const long long t1 = clock64(), r1 = wall_clock64();
    if (blockIdx.x == 0 && tid == 0) { g_clk[0] = t1 - t0; g_clk[1] = r1 - r0; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) s += c[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int TM, int TN, int SG>
void go1(const uint32_t *w8, int64_t ncols, const uint2 *lut, int n_q, float *out)
{
    const int n_t = 32;
    const int blocks = 4096 * 4 / (TM * TN);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k1<TM, TN, SG>), dim3(blocks), dim3(256), 0, 0, w8, ncols, lut, n_q, n_t, out);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    }
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    long long clk[2]; (void)hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
    const double pairsnp = (double)blocks * 4 * TM * TN * 1024.0 * 16.0 * n_q;
    printf("1 wave/SIMD TM=%d TN=%d sg=%d  %8.3f ms  %7.1f useful TFLOP/s  %.3e pair-SNP/s  %.0f cyc per 16 SNPs per wave, clk %.0f MHz\n",
           TM, TN, SG, ms, pairsnp * 2 / ms / 1e9, pairsnp / ms * 1e3, (double)clk[0] / n_q, (double)clk[0] / ((double)clk[1] / 100.0));
}

int main()
{
    const int64_t ncols = 10240; const int K = 16384, n_d = K / 8, n_q = K / 16;
    std::vector<uint32_t> h((size_t)(n_d + 8) * ncols);
    uint64_t x = 88172645463325252ull;
    for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)x & 0x78787878u; }   // byte = 8 * (0..15)
    std::vector<uint2> l((size_t)(K / 2) * 16);
    for (size_t i = 0; i < l.size(); i++) {
        const __half a = __float2half(0.5f + (i % 7) * 0.25f), b = __float2half(1e-3f * (i % 5));
        uint16_t ua, ub; memcpy(&ua, &a, 2); memcpy(&ub, &b, 2);
        l[i].x = ua | ((uint32_t)ua << 16); l[i].y = ub | ((uint32_t)ub << 16);
    }
    uint32_t *w8; uint2 *lut; float *out;
    (void)hipMalloc(&w8, h.size() * 4); (void)hipMemcpy(w8, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&lut, l.size() * 8); (void)hipMemcpy(lut, l.data(), l.size() * 8, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 16384 * 256 * 4);
    go<2, 4, 2, 3, 0>(w8, ncols, lut, n_q, out);
    go1<4, 4, 0>(w8, ncols, lut, n_q, out);
    go1<4, 4, 1>(w8, ncols, lut, n_q, out);
    go1<3, 4, 1>(w8, ncols, lut, n_q, out);
    go1<2, 4, 1>(w8, ncols, lut, n_q, out);
    return 0;
}
