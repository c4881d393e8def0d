// Round 6 (VERDICT r05 #4b, #6): the K loop of the single-product SYRK in three instruction forms, as instruction-mix models with
// real LDS lookups, plus the numerical premise of the 16x16x32 form (tools only; not part of libsnpgpu).
//
//   A  today: per 16 SNPs a wave (128 x 128 tile, one wave per SIMD) issues 16 v_mfma_f32_32x32x16_f16 and 32 table lookups
//      (ds_read_b32, one address op each) -- syrk_uv_kernel's mix
//   B  v_mfma_f32_16x16x32_f16: the same tile as 8 x 8 sub-tiles; per 32 SNPs 64 MFMAs (half the cycles each) and 64 lookups -- the same
//      lookups and flops per SNP, twice the MFMA instructions; would allow fp32 runs twice as long IF an instruction rounds once per 32
//      products instead of once per 16
//   C  KING-homo's both-missing weight contraction on v_mfma_i32_32x32x32_i8 (7-bit weight factors): per 32 SNPs 16 MFMAs (twice the
//      fp16 rate), the column operand from 32 half-dword lookups + 16 combines, the binary row operand decoded in registers (4 VALU
//      per dword, 16 dwords)
//   D  the matrix pipe alone for each shape (no lookups)
//   F  B with the addressing of the shipped kernel: ONE vector op per lookup (v_add_u32_sdwa: table base + byte b of the word)
//   E  16x16x32 with the operands made WITHOUT table lookups: the pair byte holds two e2m1 nibbles of (g - c); one
//      v_cvt_scalef32_pk_f16_fp4 (byte select by op_sel) + one v_pk_mul_f16 by the lane's factor pair per operand dword, the factor
//      pairs of a 32-SNP group read once per side (two ds_read_b128 per 64 MFMAs instead of 64 ds_read_b32 + 64 address ops)
// and, on real data, the accumulation error of K = 11 264 products per element in fp32 through 32x32x16 against 16x16x32.
// usage: r06_kloop_ubench [iters]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t lds32(uint32_t addr)
{
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

template <int BYTE> __device__ __forceinline__ uint32_t add_byte(uint32_t base, uint32_t w)      // base + byte BYTE of w: one VALU
{
    uint32_t a;
    if constexpr (BYTE == 0) asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a) : "v"(base), "v"(w));
    else if constexpr (BYTE == 1) asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a) : "v"(base), "v"(w));
    else if constexpr (BYTE == 2) asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a) : "v"(base), "v"(w));
    else asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(a) : "v"(base), "v"(w));
    return a;
}
__device__ __forceinline__ uint32_t add_byte_n(uint32_t base, uint32_t w, int b)
{
    switch (b & 3) { case 0: return add_byte<0>(base, w); case 1: return add_byte<1>(base, w); case 2: return add_byte<2>(base, w); default: return add_byte<3>(base, w); }
}
__device__ __forceinline__ uint32_t lds32_o64(uint32_t addr)
{
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1 offset:64" : "=v"(v) : "v"(addr));
    return v;
}

__device__ __forceinline__ u4 lds128(uint32_t addr)
{
    u4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

__device__ __forceinline__ h2 cvt_fp4_pair(uint32_t w, int b)      // byte b of w: two e2m1 nibbles -> two fp16 (b folds after unrolling)
{
    switch (b & 3) {
    case 0: return __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 0);
    case 1: return __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 1);
    case 2: return __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 2);
    default: return __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 3);
    }
}

// SHAPE 0: 32x32x16 f16; 1: 16x16x32 f16; 2: 32x32x32 i8 with a register-decoded binary row operand.  LK: lookups on / off.
template <int SHAPE, int LK>
__global__ __launch_bounds__(256, 1) void kloop(int iters, float *out, uint32_t seed)
{
    __shared__ uint32_t slut[2][8192];                 // 64 KiB of table, as syrk_uv_kernel's two chunks
    // fp16 pairs shaped like the single-product tables: (g - c) x an fp16 factor with a full mantissa, either sign, a third zeros
    for (int e = threadIdx.x; e < 16384; e += 256) {
        const uint32_t x = (e * 2654435761u + seed) * 2246822519u;
        const uint32_t lo = (x % 3u == 0u) ? 0u : (0x3C00u | (x & 0x83FFu) | ((x >> 11) & 0x0400u));
        const uint32_t y = x * 3266489917u;
        const uint32_t hi = (y % 3u == 0u) ? 0u : (0x3C00u | (y & 0x83FFu) | ((y >> 11) & 0x0400u));
        (&slut[0][0])[e] = lo | (hi << 16);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t tb = (uint32_t)(uintptr_t)(&slut[0][0]) + 4 * (lane & 31);
    const uint32_t tq = (uint32_t)(uintptr_t)(&slut[0][0]) + 32 * (lane >> 4);     // factor pairs of the lane's K quarter
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = (seed * (i + 7) + threadIdx.x * 0x9E3779B1u);
    uint32_t wc[SHAPE == 3 ? 8 : 1];                   // E: the column side's own words (the lookup forms tell the sides apart by table offset)
#pragma unroll
    for (int i = 0; i < (SHAPE == 3 ? 8 : 1); i++) wc[i] = (seed * (i + 19) + threadIdx.x * 0x85EBCA6Bu);
    constexpr int NA = (SHAPE == 1 || SHAPE == 3 || SHAPE == 4) ? 8 : 4;             // operand registers (4 dwords each) per side
    u4 A[2][NA], B[2][NA];
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int i = 0; i < NA; i++) { A[s][i] = (u4{w[0], w[1], w[2], w[3]} & 0x87FF87FFu) | 0x38003800u; B[s][i] = (u4{w[4], w[5], w[6], w[7]} & 0x87FF87FFu) | 0x38003800u; }
    v16f c0[SHAPE == 0 ? 16 : 1];
    v4f c1[(SHAPE == 1 || SHAPE == 3 || SHAPE == 4) ? 64 : 1];
    v16i c2[SHAPE == 2 ? 16 : 1];
    if constexpr (SHAPE == 0) {
#pragma unroll
        for (int i = 0; i < 16; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) c0[i][r] = 0.f;
    }
    if constexpr (SHAPE == 1 || SHAPE == 3 || SHAPE == 4) {
#pragma unroll
        for (int i = 0; i < 64; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) c1[i][r] = 0.f;
    }
    if constexpr (SHAPE == 2) {
#pragma unroll
        for (int i = 0; i < 16; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) c2[i][r] = 0;
    }

    // one lookup: byte `b` of key word `k` -> table address -> one dword of operand register `dst`
#define LOOKUP(dst, k, b, off)                                                    \
    do {                                                                          \
        uint32_t a_ = tb + (((w[k] >> (8 * (b))) & 0xFFu) << 7) + (off);          \
        dst = lds32(a_);                                                          \
    } while (0)
#define STEP(s, t)                                                       \
    do {                                                                 \
        if constexpr (SHAPE == 0) { \
_Pragma("unroll") \
            for (int m = 0; m < 16; m++) { \
                c0[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16((h8)A[s][m >> 2], (h8)B[s][m & 3], c0[m], 0, 0, 0); \
                if (LK) { \
                    const int L0 = 2 * m, L1 = 2 * m + 1; \
                    if (L0 < 16) LOOKUP(A[t][L0 >> 2][L0 & 3], L0 >> 2, L0 & 3, 0); else LOOKUP(B[t][(L0 - 16) >> 2][L0 & 3], 4 + ((L0 - 16) >> 2), L0 & 3, 64); \
                    if (L1 < 16) LOOKUP(A[t][L1 >> 2][L1 & 3], L1 >> 2, L1 & 3, 0); else LOOKUP(B[t][(L1 - 16) >> 2][L1 & 3], 4 + ((L1 - 16) >> 2), L1 & 3, 64); \
                } \
                __builtin_amdgcn_sched_barrier(0); \
            } \
        } else if constexpr (SHAPE == 1) { \
_Pragma("unroll") \
            for (int m = 0; m < 64; m++) { \
                c1[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16((h8)A[s][m >> 3], (h8)B[s][m & 7], c1[m], 0, 0, 0); \
                if (LK) { \
                    if (m < 32) LOOKUP(A[t][m >> 2][m & 3], (m >> 2) & 7, m & 3, 0); else LOOKUP(B[t][(m - 32) >> 2][m & 3], ((m - 32) >> 2) & 7, m & 3, 64); \
                } \
                __builtin_amdgcn_sched_barrier(0); \
            } \
        } else if constexpr (SHAPE == 4) { \
_Pragma("unroll") \
            for (int m = 0; m < 64; m++) { \
                c1[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16((h8)A[s][m >> 3], (h8)B[s][m & 7], c1[m], 0, 0, 0); \
                if (LK) { \
                    const uint32_t a_ = add_byte_n(tb, w[(m >> 2) & 7], m & 3); \
                    if (m < 32) A[t][m >> 2][m & 3] = lds32(a_); else B[t][(m - 32) >> 2][m & 3] = lds32_o64(a_); \
                } \
                __builtin_amdgcn_sched_barrier(0); \
            } \
        } else if constexpr (SHAPE == 3) { \
            u4 fu_ = u4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u}, fv_ = fu_; \
            if (LK) { fu_ = lds128(tq + ((w[0] >> 3) & 0x1F00u)); fv_ = lds128(tq + 16 + ((w[1] >> 3) & 0x1F00u)); } \
_Pragma("unroll") \
            for (int m = 0; m < 64; m++) { \
                c1[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16((h8)A[s][m >> 3], (h8)B[s][m & 7], c1[m], 0, 0, 0); \
                if (LK) { \
                    const h2 x_ = cvt_fp4_pair(m < 32 ? w[(m >> 2) & 7] : wc[(m >> 2) & 7], m & 3); \
                    const uint32_t f_ = m < 32 ? fu_[m & 3] : fv_[m & 3]; \
                    const h2 y_ = x_ * __builtin_bit_cast(h2, f_); \
                    if (m < 32) A[t][m >> 2][m & 3] = __builtin_bit_cast(uint32_t, y_); else B[t][(m - 32) >> 2][m & 3] = __builtin_bit_cast(uint32_t, y_); \
                } \
                __builtin_amdgcn_sched_barrier(0); \
            } \
        } else { \
_Pragma("unroll") \
            for (int m = 0; m < 16; m++) { \
                c2[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8((v4i)A[s][m >> 2], (v4i)B[s][m & 3], c2[m], 0, 0, 0); \
                if (LK) { \
 \
                    uint32_t lo_, hi_; \
                    LOOKUP(lo_, 4 + (m >> 2), m & 3, 64); \
                    LOOKUP(hi_, 4 + (m >> 2), (m + 1) & 3, 66); \
                    B[t][m >> 2][m & 3] = (lo_ & 0xFFFFu) | (hi_ << 16); \
 \
                    const uint32_t x_ = (w[m >> 2] >> (8 * (m & 3))) & ((w[m >> 2] >> (8 * (m & 3) + 1))) & 0x55u; \
                    A[t][m >> 2][m & 3] = (x_ * 0x00204081u) & 0x01010101u; \
                } \
                __builtin_amdgcn_sched_barrier(0); \
            } \
        } \
_Pragma("unroll") \
        for (int i = 0; i < 8; i++) w[i] = w[i] * 1664525u + 1013904223u; \
        if constexpr (SHAPE == 4) { \
_Pragma("unroll") \
            for (int i = 0; i < 8; i++) w[i] &= 0x7C7C7C7Cu;      /* bytes = 4 x a 5-bit index: every address stays inside the table */ \
        } \
        if constexpr (SHAPE == 3) { \
_Pragma("unroll") \
            for (int i = 0; i < 8; i++) wc[i] = wc[i] * 22695477u + 1u; \
        } \
    } while (0)
    for (int it = 0; it < iters; it += 2) { STEP(0, 1); STEP(1, 0); }
#undef STEP
    float sres = 0.f;
    if constexpr (SHAPE == 0) {
#pragma unroll
        for (int i = 0; i < 16; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) sres += c0[i][r];
    }
    if constexpr (SHAPE == 1 || SHAPE == 3 || SHAPE == 4) {
#pragma unroll
        for (int i = 0; i < 64; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) sres += c1[i][r];
    }
    if constexpr (SHAPE == 2) {
#pragma unroll
        for (int i = 0; i < 16; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) sres += (float)c2[i][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = sres;
#undef LOOKUP
}

// accuracy: one 32 x 32 output block, K products per element accumulated in fp32 by the two instructions from the same fp16 data
// (row-major a[32][K], b[32][K]); out32[32][32], out16[32][32]
__global__ void acc_kernel(const _Float16 *__restrict__ a, const _Float16 *__restrict__ b, int K, float *__restrict__ out32, float *__restrict__ out16)
{
    const int l = threadIdx.x;
    {   // 32x32x16: lane l holds row (l & 31), k = 8 (l >> 5) .. + 8 of each 16-slice
        v16f c;
        for (int r = 0; r < 16; r++) c[r] = 0.f;
        for (int k0 = 0; k0 < K; k0 += 16) {
            h8 x, y;
            for (int e = 0; e < 8; e++) { x[e] = a[(l & 31) * K + k0 + 8 * (l >> 5) + e]; y[e] = b[(l & 31) * K + k0 + 8 * (l >> 5) + e]; }
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0);
        }
        for (int r = 0; r < 16; r++) out32[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
    }
    for (int ti = 0; ti < 2; ti++)
        for (int tj = 0; tj < 2; tj++) {   // 16x16x32: lane l holds row (l & 15), k = 8 (l >> 4) .. + 8 of each 32-slice
            v4f c = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < K; k0 += 32) {
                h8 x, y;
                for (int e = 0; e < 8; e++) { x[e] = a[(16 * ti + (l & 15)) * K + k0 + 8 * (l >> 4) + e]; y[e] = b[(16 * tj + (l & 15)) * K + k0 + 8 * (l >> 4) + e]; }
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0);
            }
            for (int r = 0; r < 4; r++) out16[(16 * ti + 4 * (l >> 4) + r) * 32 + 16 * tj + (l & 15)] = c[r];
        }
}

template <int SHAPE, int LK>
static double run(int iters, float *d_out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kloop<SHAPE, LK>), dim3(256), dim3(256), 0, 0, iters / 8, d_out, 3u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kloop<SHAPE, LK>), dim3(256), dim3(256), 0, 0, iters, d_out, 3u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float *d_out;
    hipMalloc(&d_out, 256 * 256 * 4);
    // SNPs per K-step: A 16, B 32, C 32 -> microseconds per 1024 SNPs of a 128 x 128 wave tile
    const double a1 = run<0, 1>(iters, d_out), a0 = run<0, 0>(iters, d_out);
    const double b1 = run<1, 1>(iters, d_out), b0 = run<1, 0>(iters, d_out);
    const double c1 = run<2, 1>(iters, d_out), c0 = run<2, 0>(iters, d_out);
    const double e1 = run<3, 1>(iters, d_out), f1 = run<4, 1>(iters, d_out), e1x = run<3, 1>(iters, d_out), f1x = run<4, 1>(iters, d_out);
    auto per = [&](double ms, int snps) { return ms * 1e3 / ((double)iters * snps) * 1024.0; };
    printf("us per 1024 SNPs of a 128 x 128 wave tile, one wave per SIMD, 256 CUs, %d K-steps:\n", iters);
    printf("A  32x32x16 f16, 16 MFMA + 32 lookups per 16 SNPs : %.2f   (matrix pipe alone %.2f)\n", per(a1, 16), per(a0, 16));
    printf("B  16x16x32 f16, 64 MFMA + 64 lookups per 32 SNPs : %.2f   (matrix pipe alone %.2f)   B / A = %.3f\n", per(b1, 32), per(b0, 32), per(b1, 32) / per(a1, 16));
    printf("C  32x32x32 i8, 16 MFMA + 32 half lookups + 16 combines + register-decoded binary row operand per 32 SNPs : %.2f   (matrix pipe alone %.2f)   C / A = %.3f\n",
           per(c1, 32), per(c0, 32), per(c1, 32) / per(a1, 16));

    printf("F  16x16x32 f16, 64 MFMA + 64 (v_add_u32_sdwa + ds_read_b32) per 32 SNPs : %.2f   (again %.2f)\n", per(f1, 32), per(f1x, 32));
    printf("E  16x16x32 f16, 64 MFMA + 64 (v_cvt_scalef32_pk_f16_fp4 + v_pk_mul_f16) + 2 ds_read_b128 per 32 SNPs : %.2f   (again %.2f)   E / F = %.3f\n",
           per(e1, 32), per(e1x, 32), (per(e1, 32) + per(e1x, 32)) / (per(f1, 32) + per(f1x, 32)));
    // accuracy premise of B
    const int K = 11264;
    std::vector<_Float16> ha(32 * K), hb(32 * K);
    srand(11);
    for (int s = 0; s < K; s++) {
        const double p = 0.05 + 0.9 * (rand() / (double)RAND_MAX), t = 1.0 / (p * (1 - p));
        const _Float16 u = (_Float16)sqrt(t), v = (_Float16)(t / (double)u);
        const int ca = (int)lround(2 * p), cb = (rand() & 1) ? ca : (2 * p > ca ? ca + 1 : ca - 1);
        for (int i = 0; i < 32; i++) {
            const int g1 = (rand() / (double)RAND_MAX < p) + (rand() / (double)RAND_MAX < p), g2 = (rand() / (double)RAND_MAX < p) + (rand() / (double)RAND_MAX < p);
            ha[i * K + s] = (_Float16)((double)(g1 - ca) * (double)u);
            hb[i * K + s] = (_Float16)((double)(g2 - cb) * (double)v);
        }
    }
    _Float16 *da, *db; float *o32, *o16;
    hipMalloc(&da, ha.size() * 2); hipMalloc(&db, hb.size() * 2); hipMalloc(&o32, 4096); hipMalloc(&o16, 4096);
    hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, da, db, K, o32, o16);
    std::vector<float> r32(1024), r16(1024);
    hipMemcpy(r32.data(), o32, 4096, hipMemcpyDeviceToHost); hipMemcpy(r16.data(), o16, 4096, hipMemcpyDeviceToHost);
    double e32 = 0, e16 = 0, m32 = 0, m16 = 0, scale = 0;
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            double ref = 0, sq = 0;
            for (int s = 0; s < K; s++) { const double pr = (double)ha[i * K + s] * (double)hb[j * K + s]; ref += pr; sq += pr * pr; }
            const double d32 = r32[i * 32 + j] - ref, d16 = r16[i * 32 + j] - ref;
            e32 += d32 * d32; e16 += d16 * d16; scale += sq;
            m32 = fmax(m32, fabs(d32)); m16 = fmax(m16, fabs(d16));
        }
    const double rw = sqrt(scale / 1024);        // rms size of an element's sum (a random walk of K products)
    printf("accumulation error of K = %d products per element, relative to the rms of the sums (%.1f):\n", K, rw);
    printf("   32x32x16: rms %.3e max %.3e      16x16x32: rms %.3e max %.3e      ratio (rms) %.3f  (1 / sqrt 2 = 0.707 if one rounding per instruction)\n",
           sqrt(e32 / 1024) / rw, m32 / rw, sqrt(e16 / 1024) / rw, m16 / rw, sqrt(e16 / e32));
    return 0;
}
