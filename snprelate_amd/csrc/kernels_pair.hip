// The N x N pairwise accumulators (the hot path):
//
//  pair_mfma_i8_kernel   IBS / KING / beta counters and the GCTA both-missing counts as exact int8 MFMA contractions
//      (default; replaces CIBSCount::thread_ibs_num src/genIBS.cpp:154-273, CKINGRobust::thread_ibs_num
//       src/genKING.cpp:292-426, the integer half of CKINGHomo::thread_ibs_num src/genKING.cpp:66-200, CIndivBeta
//       src/genBeta.cpp:65-183 and the serial missing-denominator loop of CGCTA_AlgArith::Run src/genPCA.cpp:1201-1224)
//  pair_popcount_kernel  the same counters as wavefront bit-ops (SNPGPU_PAIR_BACKEND=popcount)
//  syrk_uv_kernel        centred / scaled genotype outer products on fp16 MFMAs, ONE product per SNP: integer-centred
//      genotypes x the two fp16 factors of the SNP weight, one wave per SIMD (default for GRM / PCA blocks WITHOUT missing
//      calls; with syrk_x1_kernel it replaces CProdMat_AlgArith::MulAdd src/genPCA.cpp:229-312 and the TransposeGenotype /
//      GenoSub / GenoMul preparation src/genPCA.h:93-108, genPCA.cpp:315-368)
//  syrk_x1_kernel        the same sums from an exact row operand x hi / lo-split column operand (two products), one wave per
//      SIMD (GRM / PCA blocks WITH missing calls; every block with SNPGPU_SYRK_UV=0)
//  syrk_h3_kernel        <2, true>: the same product at two waves per SIMD; <2, false>: constant row table (the masked
//      p(1-p) sums of KING-homo src/genKING.cpp:115-154, EIGMIX's both-missing weights); <3, false>: three-product split
//  syrk_mfma_kernel      the same sums on fp32 MFMAs (SNPGPU_SYRK=f32; the tile north_star names)
#include <algorithm>
#include "snpgpu_internal.h"
#include <utility>

namespace snpgpu {

// ---------------------------------------------------------------------------
// tile enumeration shared by both kernels: block id -> (XCD, super-tile, tile)
// Block b is observed to run on XCD b%8; the blocks of one XCD walk the super-tiles
// {xcd, xcd+8, ...} so that concurrently resident workgroups share rows/columns in that
// XCD's L2.  The mapping only affects speed, never results.
struct TileCoord { int tr, tc; bool valid; };

__device__ __forceinline__ TileCoord map_tile(const int *__restrict__ prefix, const int *__restrict__ first,
                                              int n_sr, int n_super, int S, int n_tr, int n_tc,
                                              int tile_r, int tile_c)
{
    TileCoord t; t.valid = false; t.tr = t.tc = 0;
    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int slot = id >> 3;
    const int ss = S * S;
    const int sq = slot / ss, within = slot - sq * ss;
    const int st = sq * 8 + xcd;
    if (st >= n_super) return t;
    int lo = 0, hi = n_sr;  // find sr with prefix[sr] <= st < prefix[sr+1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= st) lo = mid; else hi = mid;
    }
    const int sr = lo, sc = first[sr] + (st - prefix[sr]);
    t.tr = sr * S + within / S;
    t.tc = sc * S + within % S;
    // inside the panel and touching the upper triangle (panel-relative coordinates)
    t.valid = (t.tr < n_tr) && (t.tc < n_tc) && ((int64_t)(t.tc + 1) * tile_c > (int64_t)t.tr * tile_r);
    return t;
}

// ---------------------------------------------------------------------------
// bit-plane pair counters.  One wave = 8 rows x 128 columns: the row samples' plane words are
// wave-uniform (scalar loads, SGPR operands), each lane owns two column samples.
// Instruction costs measured on MI355X (tools/ubench/valu_asm_ubench.hip): v_and/v_xor/v_bitop3
// 2.4 cycles per wave64 instruction, v_bcnt_u32_b32 and v_and_or_b32 4.2 -> popcounts dominate and
// 3-input logic goes through v_bitop3_b32.
template <int MODE> struct PairOps;

// f(a,b,c) truth tables for v_bitop3_b32 (a=0xF0, b=0xCC, c=0xAA)
#define BITOP3_A_OR_BC 0xF8     /* a | (b & c)  */
#define BITOP3_AXB_AND_C 0x28   /* (a ^ b) & c  */

template <> struct PairOps<PM_IBS> {   // 4 logic + 3 popcount ops / 32 SNP pairs
    typedef uint4 PV;
    static constexpr int C = 3;       // {nvalid, ibs1, ibs0}
    static __device__ __forceinline__ void run(const uint4 &r, const uint4 &c, uint32_t *cnt)
    {
        const uint32_t t0 = r.x & c.x;                     // both called
        cnt[0] += __popc(t0);
        cnt[1] += __popc(__builtin_amdgcn_bitop3_b32(c.y, r.y, t0, BITOP3_AXB_AND_C));   // one het -> IBS1
        cnt[2] += __popc(__builtin_amdgcn_bitop3_b32(r.z & c.w, r.w, c.z, BITOP3_A_OR_BC));  // opposite hom -> IBS0
    }
};
template <> struct PairOps<PM_KING_ROBUST> {   // 6 logic + 5 popcount ops / 32 SNP pairs
    typedef uint4 PV;
    static constexpr int C = 5;       // {nLoci, ibs1, ibs0, N1_Aa, N2_Aa}
    static __device__ __forceinline__ void run(const uint4 &r, const uint4 &c, uint32_t *cnt)
    {
        cnt[0] += __popc(r.x & c.x);
        const uint32_t a = r.y & c.x;                      // row het, column called
        const uint32_t b = r.x & c.y;                      // column het, row called
        cnt[3] += __popc(a);
        cnt[4] += __popc(b);
        cnt[1] += __popc(a ^ b);
        cnt[2] += __popc(__builtin_amdgcn_bitop3_b32(r.z & c.w, r.w, c.z, BITOP3_A_OR_BC));
    }
};
template <> struct PairOps<PM_KING_HOMO> {
    typedef uint4 PV;
    static constexpr int C = 2;       // {ibs1, ibs0}
    static __device__ __forceinline__ void run(const uint4 &r, const uint4 &c, uint32_t *cnt)
    {
        cnt[0] += __popc(__builtin_amdgcn_bitop3_b32(c.y, r.y, r.x & c.x, BITOP3_AXB_AND_C));
        cnt[1] += __popc(__builtin_amdgcn_bitop3_b32(r.z & c.w, r.w, c.z, BITOP3_A_OR_BC));
    }
};
template <> struct PairOps<PM_BETA> {          // CIndivBeta::thread_ibs_num, src/genBeta.cpp:65-183
    typedef uint4 PV;
    static constexpr int C = 3;       // {num, at least one het (both called), both homozygous and equal}
    static __device__ __forceinline__ void run(const uint4 &r, const uint4 &c, uint32_t *cnt)
    {
        const uint32_t t0 = r.x & c.x;
        cnt[0] += __popc(t0);
        cnt[1] += __popc(__builtin_amdgcn_bitop3_b32(c.y, r.y, t0, 0xA8));                   // (Hi | Hj) & t0
        cnt[2] += __popc(__builtin_amdgcn_bitop3_b32(r.z & c.z, r.w, c.w, BITOP3_A_OR_BC));  // (Oi&Oj) | (Ti&Tj)
    }
};
template <> struct PairOps<PM_GCTA_MISS> {     // uint2 = 64 SNPs
    typedef uint2 PV;
    static constexpr int C = 1;       // {both missing at a polymorphic SNP}
    static __device__ __forceinline__ void run(const uint2 &r, const uint2 &c, uint32_t *cnt)
    {
        cnt[0] += __popc(r.x & c.x);
        cnt[0] += __popc(r.y & c.y);
    }
};

// Row operands are stored [row group of 8][word][8 rows] so that the 8 rows of a wave for one
// word are 8*sizeof(PV) consecutive bytes (two s_load_dwordx16 for uint4 planes).
template <int MODE>
__global__ __launch_bounds__(256) void pair_popcount_kernel(
    const typename PairOps<MODE>::PV *__restrict__ rowp, const typename PairOps<MODE>::PV *__restrict__ colp,
    int KWv /* plane vectors per sample */, int64_t ncols_pad, uint32_t *__restrict__ acc, int64_t acc_plane,
    const int *__restrict__ prefix, const int *__restrict__ first, int n_sr, int n_super, int n_tr, int n_tc,
    const unsigned long long *__restrict__ d_skip_if_zero)
{
    typedef typename PairOps<MODE>::PV PV;
    constexpr int C = PairOps<MODE>::C;
    constexpr int A = PC_ROWS_PER_WAVE, BC = PC_COLS_PER_LANE;
    if (d_skip_if_zero && *d_skip_if_zero == 0ull) return;
    const TileCoord t = map_tile(prefix, first, n_sr, n_super, PC_SUPER, n_tr, n_tc, PC_TILE_R, PC_TILE_C);
    if (!t.valid) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int row_base = t.tr * PC_TILE_R + wave * A;          // wave-uniform, multiple of 8
    const int64_t col_base = (int64_t)t.tc * PC_TILE_C + lane;

    uint32_t cnt[A][BC][C];
#pragma unroll
    for (int a = 0; a < A; a++)
#pragma unroll
        for (int b = 0; b < BC; b++)
#pragma unroll
            for (int c = 0; c < C; c++) cnt[a][b][c] = 0;

    const PV *__restrict__ rp = rowp + (int64_t)(row_base / A) * KWv * A;   // [word][8 rows]
    const PV *__restrict__ cp = colp + col_base;

    // One word per iteration, loads at the top: with 5-7 resident waves per SIMD the other waves'
    // VALU work covers the load latency (measured faster than explicit ping-pong prefetching, which
    // costs registers and therefore occupancy -- tools/ubench/pc_ubench.hip).
    for (int kw = 0; kw < KWv; kw++) {
        PV r[A], c[BC];
#pragma unroll
        for (int a = 0; a < A; a++) r[a] = rp[(int64_t)kw * A + a];
#pragma unroll
        for (int b = 0; b < BC; b++) c[b] = cp[(int64_t)kw * ncols_pad + b * 64];
#pragma unroll
        for (int a = 0; a < A; a++)
#pragma unroll
            for (int b = 0; b < BC; b++) PairOps<MODE>::run(r[a], c[b], cnt[a][b]);
    }
    // accumulate into the panel's counters.  Each element has exactly one owner per launch, so the
    // atomics never contend: they are used as fire-and-forget adds (no load -> wait -> store chain).
#pragma unroll
    for (int a = 0; a < A; a++)
#pragma unroll
        for (int b = 0; b < BC; b++) {
            uint32_t *p = acc + (int64_t)(row_base + a) * ncols_pad + col_base + b * 64;
#pragma unroll
            for (int c = 0; c < C; c++)   // the ibs0 plane of the IBS / KING counters holds 2 * ibs0 (I8Scheme<PM_IBS_NOMISS>)
                atomicAdd(p + (int64_t)c * acc_plane, (((MODE == PM_IBS || MODE == PM_KING_ROBUST) && c == 2) ||
                                                       (MODE == PM_KING_HOMO && c == 1)) ? 2u * cnt[a][b][c] : cnt[a][b][c]);
        }
}

template <int MODE>
static int launch_pc(hipStream_t st, const TileGrid &tg, const void *rowp, const void *colp, int KW,
                     int64_t ncols_pad, uint32_t *acc, int64_t acc_plane, const unsigned long long *skip)
{
    typedef typename PairOps<MODE>::PV PV;
    const int KWv = (sizeof(PV) == 16) ? KW : KW / 2;
    hipLaunchKernelGGL(pair_popcount_kernel<MODE>, dim3((unsigned)tg.grid), dim3(256), 0, st, (const PV *)rowp,
                       (const PV *)colp, KWv, ncols_pad, acc, acc_plane, tg.d_prefix, tg.d_first, tg.n_sr,
                       tg.n_super, tg.n_tr, tg.n_tc, skip);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_pair_popcount(hipStream_t st, int mode, const TileGrid &tg, const void *rowp, const void *colp, int KW,
                         int64_t ncols_pad, uint32_t *acc, int64_t acc_plane,
                         const unsigned long long *d_skip_if_zero)
{
    switch (mode) {
    case PM_IBS: return launch_pc<PM_IBS>(st, tg, rowp, colp, KW, ncols_pad, acc, acc_plane, d_skip_if_zero);
    case PM_KING_ROBUST: return launch_pc<PM_KING_ROBUST>(st, tg, rowp, colp, KW, ncols_pad, acc, acc_plane, d_skip_if_zero);
    case PM_KING_HOMO: return launch_pc<PM_KING_HOMO>(st, tg, rowp, colp, KW, ncols_pad, acc, acc_plane, d_skip_if_zero);
    case PM_GCTA_MISS: return launch_pc<PM_GCTA_MISS>(st, tg, rowp, colp, KW, ncols_pad, acc, acc_plane, d_skip_if_zero);
    case PM_BETA: return launch_pc<PM_BETA>(st, tg, rowp, colp, KW, ncols_pad, acc, acc_plane, d_skip_if_zero);
    }
    set_error("launch_pair_popcount: bad mode");
    return 1;
}

// ---------------------------------------------------------------------------
// SYRK on fp32 MFMA, barrier-free main loop.
// Measured on MI355X (tools/ubench/syrk_ubench.hip): fp32 MFMA shares the SIMD datapath with VALU
// and LDS returns -- every VALU op or lane-divergent ds_read next to f32 MFMAs costs ~4-5 SIMD
// cycles -- so the decode work per MFMA is what separates this kernel from the 98.7 % pure-MFMA
// loop.  Hence PAIR-coded genotype words and a 16-entry float2 table per SNP pair in LDS: TWO
// operand values cost ONE VALU op (v_add_u32_sdwa: table address = base + byte) plus ONE
// conflict-free ds_read_b64, i.e. 0.5 decode events per MFMA with a 64 x 64 tile per wave
// (2 x 2 v_mfma_f32_32x32x2_f32 accumulators; micro-benchmark: 90 % of peak.  A 64 x 128 tile per
// wave measures the same at N = 100 000 and 8 % less at N = 20 000: more tail, fewer waves).
// Workgroup = 4 waves (2x2), tile 128 x 128.  Lane l of a wave needs Z[sample = l&31][snp] for the
// MFMA steps of 16-SNP group q with snp = 16q + 8h + 2p + e (h = l>>5, pair p = 0..3, e = 0/1; any K
// order is legal as long as both operands use it): half h reads dword 2q+h of ITS sample
// (coalesced 128-byte rows of W8) and walks its 4 bytes.  No operand tile lives in LDS, waves never
// wait for each other inside the K loop; the only barrier is the table swap every MM_LUTCH SNPs.
// Accumulation is fp32 for at most MM_PROMOTE = 1024 SNPs (relative rounding error ~1.5e-6 on the
// diagonal, less elsewhere), then the partial is added to the fp64 panel accumulator in HBM with
// fire-and-forget global_atomic_add_f64 (one owner per element and launch: no contention).
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 4) void syrk_mfma_kernel(
    const uint32_t *__restrict__ w8, int64_t ncols_pad, const float2 *__restrict__ lut, int n_q,
    double *__restrict__ acc, int64_t ld, int64_t tiles_c, const int *__restrict__ prefix, const int *__restrict__ first, int n_sr,
    int n_super, int n_tr, int n_tc, const unsigned long long *__restrict__ d_skip_if_zero)
{
    if (d_skip_if_zero && *d_skip_if_zero == 0ull) return;
    const TileCoord t = map_tile(prefix, first, n_sr, n_super, MM_SUPER, n_tr, n_tc, MM_TILE_R, MM_TILE_C);
    if (!t.valid) return;
    constexpr int CHE = (MM_LUTCH / 2) * 16;      // float2 entries per table chunk (128 B per SNP pair)
    __shared__ float2 slut[2][CHE];               // 2 x 16 KiB decode tables

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;       // wave position in the 2x2 grid
    const int li = lane & 31, kh = lane >> 5;
    constexpr int TM = 2, TN = 2;

    const uint32_t *__restrict__ pa = w8 + (int64_t)kh * ncols_pad + (int64_t)t.tr * MM_TILE_R + wr * 64 + li;
    const uint32_t *__restrict__ pb = w8 + (int64_t)kh * ncols_pad + (int64_t)t.tc * MM_TILE_C + wc * 64 + li;

    f32x16 c32[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) c32[i][j][r] = 0.f;
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    double *__restrict__ pacc = acc + acc_off(ld, tiles_c, (int64_t)t.tr * MM_TILE_R + wr * 64 + 4 * kh,
                                              (int64_t)t.tc * MM_TILE_C + wc * 64 + li);
    const int64_t rs = tiles_c ? ACC_TILE : ld;    // row stride inside this wave's part of the accumulator

    constexpr int QCH = MM_LUTCH / 16;             // 16-SNP groups per table chunk
    const int n_chunk = (n_q + QCH - 1) / QCH;
    const int n_ent = n_q * 8 * 16;                // float2 entries of the whole block's table

    for (int e = tid; e < CHE; e += 256) slut[0][e] = (e < n_ent) ? lut[e] : make_float2(0.f, 0.f);
    uint32_t wa[TM], wb[TN];
#pragma unroll
    for (int i = 0; i < TM; i++) wa[i] = pa[32 * i];
#pragma unroll
    for (int j = 0; j < TN; j++) wb[j] = pb[32 * j];
    __syncthreads();

    for (int c = 0; c < n_chunk; c++) {
        const int cur = c & 1;
        const int q_beg = c * QCH;
        const int q_end = (q_beg + QCH < n_q) ? (q_beg + QCH) : n_q;
        const bool more = (c + 1 < n_chunk);
        // byte address of the table of this lane-half's first SNP pair of the chunk (4 pairs per half)
        const char *tb = reinterpret_cast<const char *>(&slut[cur][0]) + 512 * kh;
        for (int q = q_beg; q < q_end; q++) {
            uint32_t a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) a[i] = wa[i];
#pragma unroll
            for (int j = 0; j < TN; j++) b[j] = wb[j];
            if (q + 1 < n_q) {                     // prefetch the next 16 SNPs
                const int64_t off = (int64_t)(q + 1) * 2 * ncols_pad;
#pragma unroll
                for (int i = 0; i < TM; i++) wa[i] = pa[off + 32 * i];
#pragma unroll
                for (int j = 0; j < TN; j++) wb[j] = pb[off + 32 * j];
            }
#pragma unroll
            for (int p = 0; p < 4; p++) {
                float2 za[TM], zb[TN];
#pragma unroll
                for (int i = 0; i < TM; i++)
                    za[i] = *reinterpret_cast<const float2 *>(tb + ((a[i] >> (8 * p)) & 0xFFu) + 128 * p);
#pragma unroll
                for (int j = 0; j < TN; j++)
                    zb[j] = *reinterpret_cast<const float2 *>(tb + ((b[j] >> (8 * p)) & 0xFFu) + 128 * p);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        c32[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(za[i].x, zb[j].x, c32[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        c32[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(za[i].y, zb[j].y, c32[i][j], 0, 0, 0);
            }
            tb += 1024;                            // 8 SNP pairs x 128 B
        }
        // every MM_PROMOTE SNPs (and at the end) flush the fp32 partial into the fp64 panel accumulator
        if (!more || ((c + 1) % (MM_PROMOTE / MM_LUTCH)) == 0) {
            double *pflush = pacc;                  // opaque: the 32 row addresses are computed here, not hoisted out of
            asm volatile("" : "+v"(pflush));        // the K loop (where the compiler kept them alive in scratch: 27 spills)
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                    double *__restrict__ pr = pflush + (int64_t)row * rs;
#pragma unroll
                    for (int j = 0; j < TN; j++) {
                        unsafeAtomicAdd(pr + 32 * j, (double)c32[i][j][r]);
                        c32[i][j][r] = 0.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep address/convert temporaries short-lived
                }
        }
        if (more) {                                // next chunk's table into the other buffer
            const float2 *__restrict__ src = lut + (int64_t)(c + 1) * CHE;
            for (int e = tid; e < CHE; e += 256)
                slut[cur ^ 1][e] = ((c + 1) * CHE + e < n_ent) ? src[e] : make_float2(0.f, 0.f);
            __syncthreads();
        }
    }
}

int launch_syrk(hipStream_t st, const TileGrid &tg, const uint32_t *w8, int64_t ncols_pad, const float2 *lut,
                int n_q, double *acc, int64_t ld, int64_t tiles_c, const unsigned long long *d_skip_if_zero)
{
    if (n_q <= 0) return 0;
    hipLaunchKernelGGL(syrk_mfma_kernel, dim3((unsigned)tg.grid), dim3(256), 0, st, w8, ncols_pad, lut, n_q, acc, ld, tiles_c,
                       tg.d_prefix, tg.d_first, tg.n_sr, tg.n_super, tg.n_tr, tg.n_tc, d_skip_if_zero);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// SYRK on fp16 MFMAs with split operands: z = hi + lo, hi = fp16(z), lo = fp16(z - hi) (22 significant
// bits, built from the fp64 table values), and
//     z z' = hi hi' + hi lo' + lo hi'  (+ lo lo' ~ 2^-22 z z', dropped)
// Products of two fp16 numbers are exact in fp32, the accumulation is fp32 inside the MFMA and fp64
// across MM_PROMOTE SNPs exactly as in syrk_mfma_kernel, so the result carries the same rounding
// budget as the fp32-MFMA form (measured against the fp64 oracle in tests/test_gpu_parity.py) while
// three v_mfma_f32_32x32x16_f16 (96 matrix-pipe cycles) replace eight v_mfma_f32_32x32x2_f32 (512) per
// 16 SNPs and 32 x 32 pairs.  Operand decode as above: pair-coded words, one v_add_u32_sdwa + one
// ds_read_b64 per SNP pair; the 8-byte table entry is {hi0 | hi1 << 16, lo0 | lo1 << 16}, so the four
// lookups of a lane ARE its 8-SNP operand registers (hi: dword 0 of each, lo: dword 1).
// Workgroup = 4 waves (2 x 2), tile 256 x 128, each wave 128 x 64 = 4 x 2 accumulators; two operand
// register sets: group q+1 is decoded while the 24 MFMAs of group q run (fp16 MFMAs overlap with
// VALU/LDS work, unlike fp32 MFMAs).  Work items as in the int8 pair kernel ({tile, K part}); the
// flush is an fp64 atomic add, so K parts may share a tile.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

//
// NP = 2 ("exact row side", blocks WITHOUT missing calls): with z = y (g - avg) per SNP and a centre c close to avg
// that has only a few binary digits,
//     z_i z_j = (g_i - c) * [y z_j]  -  (avg - c) * [y z_j]
// the row operand g - c is exact in fp16, so only the column operand w = y z needs the hi/lo split: TWO MFMAs per
// 32 x 32 x 16 instead of three, and a row operand that toggles few multiplier bits (the kernel runs against the
// socket power cap, HISTORY.md 4.5).  The second term does not depend on i: the kernel leaves it out, colcorr_kernel sums
// it per column over all blocks (fp64, ctx->colterm) and colterm_settle_kernel subtracts it from every row of the panel
// once, before a result is read (kernels_final.hip).
// The table builder writes w instead of z when the block has no missing call; with one, a missing row
// genotype would need the real-valued centre avg and the three-product kernel runs instead (the two launches
// are gated on the block's missing flag, as in the int8 pair kernel).
// The masked sums of KING-homo and EIGMIX are exact-row-side products by nature, for every block and without a
// column term: sum_s v_i v_j c_s = v_i * [c v_j] with the call indicator v (a_kind 1), and the weighted
// both-missing sums m_i * [d m_j] with the missing indicator m (a_kind 2).
// E16 (NP == 2, a_kind 0): 16-byte table entries {hi pair, lo pair, ROW pair, -} -- the row operand is per SNP,
// (g - c_s) 2^shift with c_s = avg_s rounded to a few binary digits (build_lut_kernel), so that the products have
// the variance of the centred form whatever the allele frequency (a fixed centre 1 costs a factor 1/(2p) in
// variance, i.e. accuracy, on rare variants); the words carry code * 16 and a chunk holds 256 SNPs.
// Row pair of a 16-byte table entry (bytes 8..15 hold it twice) as an EIGHT-byte LDS read.  ds_read_b32 banks are
// (a/4) mod 32, so the dword at 16 c + 8 of entry c shares its bank with entry c + 8 -- the second genotype of the pair
// being 0 or 2 -- and nearly every 32-lane group paid a two-way conflict (SQ_LDS_BANK_CONFLICT = 33 % of SQ_LDS_IDX_ACTIVE
// in round 1).  ds_read_b64 banks are (a/4) mod 64: the 16 entries sit on 16 different bank pairs, and the instruction
// costs the same two LDS cycles as a conflict-free ds_read_b32 (MI355X_MICROARCH.md, LDS).
__device__ __forceinline__ uint32_t h3_row_pair(const char *p)
{
    // volatile: the compiler must not narrow the access to the one dword that is used
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t v = *(const volatile __attribute__((address_space(3))) u32x2_t *)(p);
    return v.x;
}

template <int NP, bool E16>
__global__ __launch_bounds__(256, 2) void syrk_h3_kernel(
    const uint32_t *__restrict__ w8, int64_t ncols_pad, const uint2 *__restrict__ lut, int n_q,
    double *__restrict__ acc, int64_t ld, int64_t tiles_c, const int4 *__restrict__ work,
    const unsigned long long *__restrict__ d_skip_if_zero, const unsigned long long *__restrict__ d_missing,
    int64_t n_rows_real, int a_kind, int promote_chunks)
{
    if (d_skip_if_zero && *d_skip_if_zero == 0ull) return;
    if (d_missing && ((*d_missing != 0ull) != (NP == 3))) return;
    constexpr int TM = 4, TN = 2;    // the exact / cheaper-to-decode row side gets the four 32-sample groups
    constexpr int CHS = E16 ? H3_LUTCH / 2 : H3_LUTCH;   // SNPs per table chunk
    constexpr int PST = E16 ? 256 : 128;                 // bytes of table per SNP pair (16 entries)
    constexpr int CHE = (H3_LUTCH / 2) * 16;       // 8-byte units per chunk: 32 KiB either way
    constexpr int QCH = CHS / 16;                  // 16-SNP groups per chunk
    __shared__ uint2 slut[2][CHE];                 // 2 x 32 KiB
    __shared__ uint2 sgt[16];                      // NP == 2, !E16: pair code -> {fp16(a0) | fp16(a1) << 16} of the indicator, 8-byte stride

    const int4 item = work[blockIdx.x];
    if (item.w == 0) return;
    const int n_chunk_all = (n_q + QCH - 1) / QCH;
    const int per = (n_chunk_all + item.w - 1) / item.w;
    const int c_beg = item.z * per;
    const int c_end = (c_beg + per < n_chunk_all) ? (c_beg + per) : n_chunk_all;
    if (c_beg >= c_end) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, kh = lane >> 5;
    const uint32_t *__restrict__ pa = w8 + (int64_t)kh * ncols_pad + (int64_t)item.x * H3_TILE_R + wr * (32 * TM) + li;
    const uint32_t *__restrict__ pb = w8 + (int64_t)kh * ncols_pad + (int64_t)item.y * H3_TILE_C + wc * (32 * TN) + li;
    double *__restrict__ pacc = acc + acc_off(ld, tiles_c, (int64_t)item.x * H3_TILE_R + wr * (32 * TM) + 4 * kh,
                                              (int64_t)item.y * H3_TILE_C + wc * (32 * TN) + li);
    const int64_t rs = tiles_c ? ACC_TILE : ld;    // row stride inside this wave's part of the accumulator

    f32x16 c32[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) c32[i][j][r] = 0.f;

    if (NP == 2 && !E16 && threadIdx.x < 16) {
        // constant row operand per genotype code 0, 1, 2, 3:  a_kind 1: called, a_kind 2: missing
        const uint32_t one = 0x3C00u;
        const uint32_t c0 = threadIdx.x & 3, c1 = threadIdx.x >> 2;
        const uint32_t h0 = (a_kind == 1) ? (c0 != 3 ? one : 0u) : (c0 == 3 ? one : 0u);
        const uint32_t h1 = (a_kind == 1) ? (c1 != 3 ? one : 0u) : (c1 == 3 ? one : 0u);
        sgt[threadIdx.x] = make_uint2(h0 | (h1 << 16), 0u);
    }
    u32x4 Ah[1][TM], Al[1][NP == 3 ? TM : 1], Bh[1][TN], Bl[1][TN];
    uint32_t wa[TM], wb[TN], wa2[TM], wb2[TN];     // words of the current and of the next group
    const char *gt = reinterpret_cast<const char *>(&sgt[0]);
#define H3_LOAD_WORDS(q, A_, B_)                                                          \
    do {                                                                                  \
        const int64_t off_ = (int64_t)(q) * 2 * ncols_pad;                                \
        _Pragma("unroll") for (int i = 0; i < TM; i++) A_[i] = pa[off_ + 32 * i];         \
        _Pragma("unroll") for (int j = 0; j < TN; j++) B_[j] = pb[off_ + 32 * j];         \
    } while (0)
#define H3_DECODE(set, tb, wa, wb)                                                                \
    do {                                                                                  \
        _Pragma("unroll") for (int p = 0; p < 4; p++) {                                   \
            _Pragma("unroll") for (int i = 0; i < TM; i++) {                              \
                if (NP == 3) {                                                            \
                    const uint2 t_ = *reinterpret_cast<const uint2 *>((tb) + ((wa[i] >> (8 * p)) & 0xFFu) + PST * p); \
                    Ah[set][i][p] = t_.x; Al[set][NP == 3 ? i : 0][p] = t_.y;             \
                } else if (E16) {                                                         \
                    Ah[set][i][p] = h3_row_pair((tb) + ((wa[i] >> (8 * p)) & 0xFFu) + PST * p + 8); \
                } else {                                                                  \
                    Ah[set][i][p] = *reinterpret_cast<const uint32_t *>(gt + ((wa[i] >> (8 * p)) & 0xFFu)); \
                }                                                                         \
            }                                                                             \
            _Pragma("unroll") for (int j = 0; j < TN; j++) {                              \
                const uint2 t_ = *reinterpret_cast<const uint2 *>((tb) + ((wb[j] >> (8 * p)) & 0xFFu) + PST * p); \
                Bh[set][j][p] = t_.x; Bl[set][j][p] = t_.y;                               \
            }                                                                             \
        }                                                                                 \
    } while (0)
#define H3_MFMAS(set)  /* product-major: MFMAs on the same accumulator are TM*TN instructions apart */ \
    do {                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; i++)                                    \
            _Pragma("unroll") for (int j = 0; j < TN; j++)                                \
                c32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16((f16x8)Ah[set][i], (f16x8)Bh[set][j], c32[i][j], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < TM; i++)                                    \
            _Pragma("unroll") for (int j = 0; j < TN; j++)                                \
                c32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16((f16x8)Ah[set][i], (f16x8)Bl[set][j], c32[i][j], 0, 0, 0); \
        if (NP == 3) {                                                                    \
            _Pragma("unroll") for (int i = 0; i < TM; i++)                                \
                _Pragma("unroll") for (int j = 0; j < TN; j++)                            \
                    c32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16((f16x8)Al[set][NP == 3 ? i : 0], (f16x8)Bh[set][j], c32[i][j], 0, 0, 0); \
        }                                                                                 \
    } while (0)

    // Table chunks (32 KiB) travel HBM/L2 -> LDS without passing through VGPRs (global_load_lds_dwordx4,
    // 1 KiB per instruction and wave): the copy of chunk c+1 is issued at the start of chunk c and is
    // complete long before the barrier at its end (vmcnt is in-order and the loop waits for younger loads).
#define H3_TABLE_ASYNC(chunk, buf)                                                                             \
    do {                                                                                                       \
        const char *src_ = reinterpret_cast<const char *>(lut) + (int64_t)(chunk) * (CHE * 8) + wave * (CHE * 2) + lane * 16; \
        char *dst_ = reinterpret_cast<char *>(&slut[buf][0]) + wave * (CHE * 2);                              \
        _Pragma("unroll") for (int t_ = 0; t_ < CHE * 2 / 1024; t_++)                                                       \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src_ + 1024 * t_), \
                                             (__attribute__((address_space(3))) void *)(dst_ + 1024 * t_), 16, 0, 0); \
    } while (0)
    H3_TABLE_ASYNC(c_beg, c_beg & 1);
    H3_LOAD_WORDS(c_beg * QCH, wa, wb);
    H3_LOAD_WORDS(c_beg * QCH + 1, wa2, wb2);
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
    __syncthreads();

    // fp32 partial sums go to the fp64 panel every promote_chunks table chunks (launch_syrk_h3: 4096 SNPs for the
    // three-product kernel, one whole 16 384-SNP feed block for the exact-row kernel)
    for (int c = c_beg; c < c_end; c++) {
        const int cur = c & 1;
        const int q0 = c * QCH;
        const int q_cnt = (q0 + QCH <= n_q) ? QCH : (n_q - q0);      // multiple of 4 (blocks are padded to 64 SNPs)
        const bool more = (c + 1 < c_end);
        // byte address of the tables of this lane-half's 4 SNP pairs of group 0 of the chunk
        const char *tb = reinterpret_cast<const char *>(&slut[cur][0]) + 4 * PST * kh;
        for (int q = 0; q < q_cnt; q += 2) {        // q_cnt is even; words are loaded two groups ahead
            H3_DECODE(0, tb, wa, wb);
            tb += 8 * PST;
            H3_LOAD_WORDS(q0 + q + 2, wa, wb);      // W8 has spare rows: reading ahead is always legal
            // next chunk's table: issued together with a word load (same latency, in-order return), so the
            // vmcnt wait of a later decode does not stall on it
            if (q == 0 && more) H3_TABLE_ASYNC(c + 1, cur ^ 1);
            H3_MFMAS(0);
            H3_DECODE(0, tb, wa2, wb2);
            tb += 8 * PST;
            H3_LOAD_WORDS(q0 + q + 3, wa2, wb2);
            H3_MFMAS(0);
        }
        // every MM_PROMOTE SNPs (and at the end of the part) flush the fp32 partial into the fp64 panel
        if (!more || ((c + 1) % promote_chunks) == 0) {
            double *pflush = pacc;                  // opaque: keeps the 32 row addresses out of the main loop's
            asm volatile("" : "+v"(pflush));        // live ranges (the compiler would precompute and spill them)
            // rows of real samples at / below this lane's first row; padding rows are never written (they stay 0)
            const int64_t rows_left = (n_rows_real > 0 ? n_rows_real : ((int64_t)1 << 40)) -
                                      ((int64_t)item.x * H3_TILE_R + wr * (32 * TM) + 4 * kh);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                    double *__restrict__ pr = pflush + (int64_t)row * rs;
                    const bool real_row = (row < rows_left);
#pragma unroll
                    for (int j = 0; j < TN; j++) {
                        // fp32 partials, exactly representable in fp64: the panel sums do not depend on the order in
                        // which the K parts of a tile arrive
                        if (real_row) unsafeAtomicAdd(pr + 32 * j, (double)c32[i][j][r]);
                        c32[i][j][r] = 0.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        if (more) {                                 // the next chunk's table is in place for every wave
            // vmcnt is in-order: the table copy was issued in the chunk's first iteration, the only loads that may
            // still be in flight behind it are the words of the next two groups (2 (TM + TN) = 12; a chunk that is
            // followed by another one is full, so at least that many were issued after the copy) -- wait for
            // everything older, not for them
            static_assert(2 * (TM + TN) == 12, "s_waitcnt immediate below");
            __builtin_amdgcn_s_waitcnt(0x0F7C);     // vmcnt(12)
            __syncthreads();
        }
    }
#undef H3_TABLE_ASYNC
#undef H3_LOAD_WORDS
#undef H3_DECODE
#undef H3_MFMAS
}

// ---------------------------------------------------------------------------
// syrk_x1_kernel: the exact-row SYRK (syrk_h3_kernel<2, true>: same tables, same words, same arithmetic) with ONE wave
// per SIMD.  Each wave owns 128 x 128 = 4 x 4 accumulators (256 AGPRs), the workgroup (2 x 2 waves) a 256 x 256 tile: 32
// table lookups per 32 MFMAs instead of 24 per 16, and a third fewer sample words per flop.  With the SIMD to itself the
// wave hides its own latencies: two operand register sets -- the lookups of group g + 1 are interleaved one by one with the
// MFMAs of group g (source order pinned by sched_barrier) -- and a ring of four word sets (the words of group g + 4 are
// requested during group g and first used during group g + 3).  Table chunks as in syrk_h3_kernel (256 SNPs, 32 KiB, double
// buffered, HBM/L2 -> LDS without VGPRs); the barrier of a chunk boundary sits before the chunk's LAST group, whose MFMAs
// then cover the first lookups out of the next chunk's table.  Table entries are 12 bytes {hi pair, lo pair, row pair}: the
// dword banks 3 c + {0, 1, 2} (mod 32) of the 16 entries of a pair are distinct, so every lookup is a conflict-free
// ds_read_b32 straight into its slot of an MFMA operand (an 8-byte read needs a v_mov -- and a wait -- per value).
// LDS addresses are carried as 32-bit byte offsets (a generic pointer that passes through an opaque asm loses its address
// space and comes back as 64-bit arithmetic plus null checks)
typedef __attribute__((address_space(3))) const char x1_lds_char;
typedef __attribute__((address_space(3))) const volatile uint32_t x1_lds_u32;
__device__ __forceinline__ uint32_t x1_lds_off(const void *shared_ptr)
{
    return (uint32_t)(uintptr_t)(x1_lds_char *)shared_ptr;
}
__device__ __forceinline__ uint32_t x1_lds32(uint32_t off)
{
    return *(x1_lds_u32 *)(uintptr_t)off;
}
// 16 bytes per lane HBM / L2 -> LDS without VGPRs (lane l lands at lds_base + 16 l).  Written as inline asm on purpose: the
// compiler models __builtin_amdgcn_global_load_lds as a FLAT access that may touch LDS *and* memory, and while one is
// pending every wait it inserts becomes vmcnt(0) / lgkmcnt(0) -- with a table copy in flight for most of a chunk that
// turned all the counted waits of the word loads and lookups into full drains.  An instruction the waitcnt pass does not
// see only makes its vmcnt(N) waits conservative (the counter is in-order and the copy adds outstanding requests); the
// kernels wait for the copy explicitly (s_waitcnt vmcnt + barrier) before the first lookup in the new table.
__device__ __forceinline__ void x1_lds_dma16(const void *gsrc, uint32_t lds_base)
{
    const uint32_t b = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(b) : "memory");
}

__global__ __launch_bounds__(256, 1) void syrk_x1_kernel(
    const uint32_t *__restrict__ w8, int64_t ncols_pad, const uint2 *__restrict__ lut, int n_q,
    double *__restrict__ acc, int64_t ld, int64_t tiles_c, const int4 *__restrict__ work,
    const unsigned long long *__restrict__ d_skip_if_zero, int64_t n_rows_real, int chunk_lo, int chunk_hi, int n_runs, int run_chunks,
    int run_group, int n_items8, const unsigned long long *__restrict__ d_short_runs, int short_div)
{
    if (d_skip_if_zero && *d_skip_if_zero == 0ull) return;
    constexpr int TM = 4, TN = 4, D = 4;
    constexpr int CHS = X1_CHS;                    // SNPs per table chunk
    constexpr int PST = 192;                       // bytes of table per SNP pair: 16 entries of 12 bytes {hi pair, lo pair, row pair}
    constexpr int CHE = (CHS / 2) * PST / 8;       // 8-byte units per chunk: 24 KiB
    constexpr int QCH = CHS / 16;                  // 16-SNP groups per chunk
    static_assert(QCH % (2 * D) == 0, "whole double rounds of the word banks per chunk");
    __shared__ uint2 slut[2][CHE];

    // n_runs > 1: one launch for all fp32 runs of the block, work items (tile, run), run fastest per XCD (see syrk_uv_kernel)
    int wi = blockIdx.x;
    if (n_runs > 1) {
        const int kpos = (int)blockIdx.x >> 3, span = run_group * n_runs;
        const int grp = kpos / span, within = kpos - grp * span, run = within / run_group, ti = grp * run_group + (within - run * run_group);
        if (ti >= n_items8) return;
        wi = ti * 8 + ((int)blockIdx.x & 7);
        // the launch is laid out for the SHORT runs (run_chunks / short_div table chunks each: blocks that hold rare variants next
        // to missing calls, flag set by build_lut_kernel); any other block uses the first runs of it at the full length
        const int rc = (d_short_runs && *d_short_runs != 0ull) ? run_chunks / short_div : run_chunks;
        chunk_lo = run * rc;
        if (chunk_lo >= chunk_hi) return;
        chunk_hi = (chunk_lo + rc < chunk_hi) ? (chunk_lo + rc) : chunk_hi;
    } else if (d_short_runs) {
        // one launch per run (SNPGPU_RUN_INNER=0), two launches per full-length run [chunk_lo, chunk_hi): `run_group` = which half.
        // A flagged block runs both halves as fp32 runs of their own -- the same numerics as the fused launch (this
        // A/B switch used to change the accuracy as well) --, any other block does the whole run in launch 0 and launch 1 exits
        if (*d_short_runs != 0ull) {
            chunk_lo += run_group * (run_chunks / short_div);
            if (chunk_lo >= chunk_hi) return;
            if (run_group == 0) chunk_hi = (chunk_lo + run_chunks / short_div < chunk_hi) ? (chunk_lo + run_chunks / short_div) : chunk_hi;
        } else if (run_group != 0) return;
    }
    const int4 item = work[wi];
    if (item.w == 0) return;
    // this launch covers the table chunks [chunk_lo, chunk_hi) -- one fp32 run: the accumulators are flushed ONCE, after the
    // K loop (a flush inside the loop writes them with VALU instructions and the compiler then keeps all 256 in VGPRs,
    // shuttling each through AGPRs around its MFMA: 1140 bytes of scratch per lane)
    const int per = (chunk_hi - chunk_lo + item.w - 1) / item.w;
    const int c_beg = chunk_lo + item.z * per;
    const int c_end = (c_beg + per < chunk_hi) ? (c_beg + per) : chunk_hi;
    if (c_beg >= c_end) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, kh = lane >> 5;
    const int64_t row_w = (int64_t)item.x * X1_TILE + wr * (32 * TM), col_w = (int64_t)item.y * X1_TILE + wc * (32 * TN);
    const uint32_t *__restrict__ pa = w8 + (int64_t)kh * ncols_pad + row_w + li;
    const uint32_t *__restrict__ pb = w8 + (int64_t)kh * ncols_pad + col_w + li;
    double *__restrict__ pacc = acc + acc_off(ld, tiles_c, row_w + 4 * kh, col_w + li);
    const int64_t rs = tiles_c ? ACC_TILE : ld;    // row stride inside this wave's part of the accumulator

    f32x16 c32[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) c32[i][j][r] = 0.f;

    u32x4 Ah[2][TM], Bh[2][TN], Bl[2][TN];        // two operand sets: MFMAs read one, the lookups fill the other
    // two banks of four word sets: a round of four groups looks its words up in one bank while ALL 32 word loads of the
    // next round go out during its first group into the other bank -- whatever s_waitcnt vmcnt the compiler places later in
    // the round (it is conservative across the loop edge) then finds them three groups old
    uint32_t W0a[D][TM], W0b[D][TN], W1a[D][TM], W1b[D][TN];

#define X1_LOOKUP(t, WA_, WB_, wset, m, tb)  /* lookup number m of a group: 16 row lookups, then 16 column lookups */ \
    do {                                                                                     \
        if ((m) < 16) {                                                                      \
            constexpr int i_ = ((m) < 16 ? (m) : 0) >> 2, p_ = (m) & 3;                       \
            Ah[t][i_][p_] = x1_lds32((tb) + ((WA_[wset][i_] >> (8 * p_)) & 0xFFu) + PST * p_ + 8); \
        } else {                                                                             \
            constexpr int j_ = ((m) >= 16 ? (m) - 16 : 0) >> 2, p_ = (m) & 3;                 \
            const uint32_t e_ = (tb) + ((WB_[wset][j_] >> (8 * p_)) & 0xFFu) + PST * p_;      \
            Bh[t][j_][p_] = x1_lds32(e_);      /* volatile: not to be merged into ds_read2_b32 (pair result + moves) */ \
            Bl[t][j_][p_] = x1_lds32(e_ + 4);                                                \
        }                                                                                    \
    } while (0)
#define X1_TABLE_ASYNC(chunk, buf)                                                                             \
    do {                                                                                                       \
        const char *src_ = reinterpret_cast<const char *>(lut) + (int64_t)(chunk) * (CHE * 8) + wave * (CHE * 2) + lane * 16; \
        char *dst_ = reinterpret_cast<char *>(&slut[buf][0]) + wave * (CHE * 2);                              \
        _Pragma("unroll") for (int t_ = 0; t_ < CHE * 2 / 1024; t_++)                                          \
            x1_lds_dma16(src_ + 1024 * t_, x1_lds_off(dst_ + 1024 * t_));                                      \
    } while (0)
    // word load number m (0..31) of a round: set m >> 3, sample group m & 7 (four row groups, four column groups)
#define X1_LOAD(YA_, YB_, g_first, m)                                                         \
    do {                                                                                      \
        const int64_t off_ = (int64_t)((g_first) + ((m) >> 3)) * 2 * ncols_pad;               \
        if (((m) & 7) < TM) YA_[(m) >> 3][((m) & 7) < TM ? ((m) & 7) : 0] = pa[off_ + 32 * ((m) & 7)]; \
        else YB_[(m) >> 3][((m) & 7) >= TM ? ((m) & 7) - TM : 0] = pb[off_ + 32 * (((m) & 7) - TM)];  \
    } while (0)
    // one group: 32 MFMAs out of operand set S_.  The 48 LDS reads of the NEXT group's lookups (word set NS_ of bank LA_ / LB_,
    // into operand set T_) ride behind the first 24 MFMAs, two lookups behind each of the first 8 and one behind the next 16, so
    // that the last returns while the final eight MFMAs run; LOAD_ != 0 (first group of a round): behind every MFMA one word
    // load of the next round (bank YA_ / YB_).  Plain macros: every register index is a literal.
    // address of lookup L of the next group / its LDS read(s), split so that an address is computed one MFMA slot before it is used
    // (a ds_read right behind its v_add_u32_sdwa waits for the VALU result: two such pairs filled a 32-cycle MFMA slot)
#define X1_AD(LA_, LB_, NS_, L, tb)   /* table base + the pair's byte; the constant part of the address stays an immediate of the read */ \
    ((L) < 16 ? (tb) + ((LA_[NS_][((L) < 16 ? (L) : 0) >> 2] >> (8 * ((L) & 3))) & 0xFFu)                            \
              : (tb) + ((LB_[NS_][((L) >= 16 ? (L) - 16 : 0) >> 2] >> (8 * ((L) & 3))) & 0xFFu))
#define X1_RD(T_, L, a)                                                                                              \
    do {                                                                                                            \
        if ((L) < 16) Ah[T_][((L) < 16 ? (L) : 0) >> 2][(L) & 3] = x1_lds32((a) + PST * ((L) & 3) + 8);              \
        else { Bh[T_][((L) >= 16 ? (L) - 16 : 0) >> 2][(L) & 3] = x1_lds32((a) + PST * ((L) & 3));                   \
               Bl[T_][((L) >= 16 ? (L) - 16 : 0) >> 2][(L) & 3] = x1_lds32((a) + PST * ((L) & 3) + 4); }             \
    } while (0)
#define X1_STEP(m, S_, T_, LA_, LB_, NS_, LOAD_, YA_, YB_, g_load, tb)                                              \
    do {                                                                                                            \
        c32[((m) & 15) >> 2][(m) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(                                      \
            (f16x8)Ah[S_][((m) & 15) >> 2], (f16x8)(((m) >> 4) ? Bl[S_][(m) & 3] : Bh[S_][(m) & 3]),                 \
            c32[((m) & 15) >> 2][(m) & 3], 0, 0, 0);                                                                \
        if ((m) < 8) {                                                                                              \
            X1_RD(T_, 2 * (m), a0_); X1_RD(T_, 2 * (m) + 1, a1_);                                                    \
            if ((m) < 7) { a0_ = X1_AD(LA_, LB_, NS_, 2 * (m) + 2, tb); a1_ = X1_AD(LA_, LB_, NS_, 2 * (m) + 3, tb); \
                           asm volatile("" : "+v"(a0_), "+v"(a1_)); }   /* pins the additions HERE, not next to their reads */ \
            else { a0_ = X1_AD(LA_, LB_, NS_, 16, tb); asm volatile("" : "+v"(a0_)); }                              \
        } else if ((m) < 24) {                                                                                      \
            X1_RD(T_, (m) + 8, a0_);                                                                                \
            if ((m) < 23) { a0_ = X1_AD(LA_, LB_, NS_, (m) + 9, tb); asm volatile("" : "+v"(a0_)); }                 \
        }                                                                                                           \
        if (LOAD_) X1_LOAD(YA_, YB_, g_load, m);                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    } while (0)
#define X1_STEP4(m, ...) X1_STEP(m, __VA_ARGS__); X1_STEP((m) + 1, __VA_ARGS__); X1_STEP((m) + 2, __VA_ARGS__); X1_STEP((m) + 3, __VA_ARGS__)
#define X1_GROUP_ADDR0(S_, T_, LA_, LB_, NS_, LOAD_, YA_, YB_, g_load, tb)                                           \
    uint32_t a0_ = X1_AD(LA_, LB_, NS_, 0, tb), a1_ = X1_AD(LA_, LB_, NS_, 1, tb)
#define X1_GROUP(...)                                                                                               \
    do {                                                                                                            \
        X1_GROUP_ADDR0(__VA_ARGS__);                                                                                \
        X1_STEP4(0, __VA_ARGS__); X1_STEP4(4, __VA_ARGS__); X1_STEP4(8, __VA_ARGS__); X1_STEP4(12, __VA_ARGS__);    \
        X1_STEP4(16, __VA_ARGS__); X1_STEP4(20, __VA_ARGS__); X1_STEP4(24, __VA_ARGS__); X1_STEP4(28, __VA_ARGS__); \
    } while (0)

    // prologue: table of the first chunk, the words of the first round, the lookups of group 0
    X1_TABLE_ASYNC(c_beg, c_beg & 1);
#define X1_L8(m) X1_LOAD(W0a, W0b, c_beg * QCH, m); X1_LOAD(W0a, W0b, c_beg * QCH, (m) + 1); X1_LOAD(W0a, W0b, c_beg * QCH, (m) + 2); X1_LOAD(W0a, W0b, c_beg * QCH, (m) + 3); \
                 X1_LOAD(W0a, W0b, c_beg * QCH, (m) + 4); X1_LOAD(W0a, W0b, c_beg * QCH, (m) + 5); X1_LOAD(W0a, W0b, c_beg * QCH, (m) + 6); X1_LOAD(W0a, W0b, c_beg * QCH, (m) + 7)
    X1_L8(0); X1_L8(8); X1_L8(16); X1_L8(24);
#undef X1_L8
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
    __syncthreads();
    uint32_t tbn = x1_lds_off(&slut[c_beg & 1][0]) + 4 * PST * kh;
#define X1_L4(m) X1_LOOKUP(0, W0a, W0b, 0, m, tbn); X1_LOOKUP(0, W0a, W0b, 0, (m) + 1, tbn); X1_LOOKUP(0, W0a, W0b, 0, (m) + 2, tbn); X1_LOOKUP(0, W0a, W0b, 0, (m) + 3, tbn)
    X1_L4(0); X1_L4(4); X1_L4(8); X1_L4(12); X1_L4(16); X1_L4(20); X1_L4(24); X1_L4(28);
#undef X1_L4
    tbn += 8 * PST;

    for (int c = c_beg; c < c_end; c++) {
        const int cur = c & 1;
        const int q0 = c * QCH;
        const int q_cnt = (q0 + QCH <= n_q) ? QCH : (n_q - q0);      // multiple of 8 (blocks are padded to 128 SNPs)
        const bool more = (c + 1 < c_end);
        if (more) X1_TABLE_ASYNC(c + 1, cur ^ 1);   // every wave is past the barrier that freed this buffer
        for (int q = 0; q < q_cnt; q += 2 * D) {
            const int g = q0 + q;
            // round A: words of bank 0, loads into bank 1
            X1_GROUP(0, 1, W0a, W0b, 1, 1, W1a, W1b, g + 4, tbn); tbn += 8 * PST;
            X1_GROUP(1, 0, W0a, W0b, 2, 0, W1a, W1b, g + 4, tbn); tbn += 8 * PST;
            X1_GROUP(0, 1, W0a, W0b, 3, 0, W1a, W1b, g + 4, tbn); tbn += 8 * PST;
            X1_GROUP(1, 0, W1a, W1b, 0, 0, W1a, W1b, g + 4, tbn); tbn += 8 * PST;
            // round B: words of bank 1, loads into bank 0
            X1_GROUP(0, 1, W1a, W1b, 1, 1, W0a, W0b, g + 8, tbn); tbn += 8 * PST;
            X1_GROUP(1, 0, W1a, W1b, 2, 0, W0a, W0b, g + 8, tbn); tbn += 8 * PST;
            X1_GROUP(0, 1, W1a, W1b, 3, 0, W0a, W0b, g + 8, tbn); tbn += 8 * PST;
            // the chunk's last group looks up the NEXT chunk's table (or, at the very end, harmlessly re-reads this one:
            // ONE straight-line body -- a second variant without lookups made the register allocator give the sixteen
            // accumulator tiles different AGPRs on the two paths and shuffle them through scratch every round)
            if (q + 2 * D >= q_cnt) {
                if (more) {
                    // vmcnt is in-order: the table copy was issued at the start of this (full) chunk, behind it 64 word
                    // loads, the last 32 of them a whole round ago -- wait for everything older than those
                    __builtin_amdgcn_s_waitcnt(0x8F70); // vmcnt(32)
                    __syncthreads();
                    tbn = x1_lds_off(&slut[cur ^ 1][0]) + 4 * PST * kh;
                } else {
                    tbn = x1_lds_off(&slut[cur][0]) + 4 * PST * kh;
                }
            }
            X1_GROUP(1, 0, W0a, W0b, 0, 0, W0a, W0b, g + 8, tbn); tbn += 8 * PST;
        }
    }
    {
        double *pflush = pacc;
        asm volatile("" : "+v"(pflush));
        const int64_t rows_left = (n_rows_real > 0 ? n_rows_real : ((int64_t)1 << 40)) - (row_w + 4 * kh);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                double *__restrict__ pr = pflush + (int64_t)row * rs;
                if (row < rows_left) {
#pragma unroll
                    for (int j = 0; j < TN; j++) unsafeAtomicAdd(pr + 32 * j, (double)c32[i][j][r]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
#undef X1_GROUP
#undef X1_GROUP_ADDR0
#undef X1_STEP4
#undef X1_STEP
#undef X1_RD
#undef X1_AD
#undef X1_LOAD
#undef X1_LOOKUP
#undef X1_TABLE_ASYNC
}

// ---------------------------------------------------------------------------
// syrk_uv_kernel: ONE fp16 product per SNP for blocks without missing calls.  The per-SNP weight y^2 = 1 / (p (1 - p)) is
// factorised as u v with u, v BOTH fp16 (uv_factor_kernel searches the 1024 mantissas of u for the one whose quotient rounds
// best: |u v / y^2 - 1| ~ 1e-6 rms, <= 4.2e-6) and the genotypes are centred at INTEGERS c_a, c_b in {0, 1, 2}:
//     row operand  (g_i - c_a) u   and   column operand  (g_j - c_b) v   are exact fp16 numbers (+-u, +-2u, 0),
// their products exact in fp32, and      u v (g_i - avg)(g_j - avg)
//     = [(g_i - c_a) u] [(g_j - c_b) v]  -  d_b u v (g_i - c_a)  -  d_a u v (g_j - c_b)  +  d_a d_b u v,   d = avg - c,
// where the last three terms are a per-row sum, a per-column sum and a constant (uvcorr_kernel, fp64; settled with the
// column term of the exact-row kernel).  The centres are picked per SNP so that the running mean of the products,
// sum d_a d_b u v, stays near zero (c_a = c_b = nearest integer gives + d^2, nearest / other neighbour gives - |d_a d_b|):
// the fp32 accumulators then carry a centred random walk as with exactly centred operands.
// Same skeleton as syrk_x1_kernel (one wave per SIMD, 4 x 4 accumulators in AGPRs, two operand sets, lookups of group g + 1
// behind the MFMAs of group g) with 8-byte table entries {row pair, column pair} (banks 2 c, 2 c + 1: conflict-free
// ds_read_b32), 16 MFMAs and 32 lookups per 16-SNP group, table chunks of 1024 SNPs (2 x 64 KiB) and two banks of EIGHT
// word sets: the groups take half the time, so the word loads run twice as many groups ahead.
// (the lookup macros index operand arrays in BOTH arms of a constant conditional; inside a template clang warns about the arm that
// is never evaluated)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Warray-bounds"
__global__ __launch_bounds__(256, 1) void syrk_uv_kernel(
    const uint32_t *__restrict__ w8, int64_t ncols_pad, const uint2 *__restrict__ lut, int n_q,
    double *__restrict__ acc, int64_t ld, int64_t tiles_c, const int4 *__restrict__ work,
    const unsigned long long *__restrict__ d_missing, int64_t n_rows_real, int chunk_lo, int chunk_hi, double fscale,
    int n_runs, int run_chunks, int n_target, int run_group, int n_items8, int run_if_missing, int64_t copy_lut_bytes,
    int64_t copy_acc_elems)
{
    // GRM / PCA: blocks WITHOUT missing calls (the others take syrk_x1_kernel).  run_if_missing: KING-homo's both-missing weight
    // sums (binary operands x the two fp16 factors of the weight, homo_uv_tables_kernel) -- blocks WITH missing calls only
    if ((*d_missing != 0ull) != (run_if_missing != 0)) return;
    constexpr int TM = 4, TN = 4, D = 8;
    constexpr int CHS = UV_CHS;                    // SNPs per table chunk
    constexpr int PST = 128;                       // bytes of table per SNP pair: 16 entries of 8 bytes
    constexpr int CHE = (CHS / 2) * PST / 8;       // 8-byte units per chunk: 64 KiB
    constexpr int QCH = CHS / 16;                  // 16-SNP groups per chunk
    static_assert(QCH % (2 * D) == 0, "whole double rounds of the word banks per chunk");
    __shared__ uint2 slut[2][CHE];

    // n_runs > 1: ONE launch for all fp32 runs of the block, work items = (tile, run) with the run index fastest inside an
    // XCD's queue (workgroup b: XCD b & 7, position b >> 3 = item * n_runs + run) -- the runs of a tile execute side by side
    // on one XCD and their fp64 flushes meet the tile's 512 KB in the Infinity Cache instead of sweeping the whole panel
    // through HBM once per run (round 5).  chunk_lo / chunk_hi / fscale then come from the run index.
    int wi = blockIdx.x;
    if (n_runs > 1) {
        // an XCD's queue is cut into groups of `run_group` tiles; a group is walked run by run (group, run, tile in group)
        const int kpos = (int)blockIdx.x >> 3, span = run_group * n_runs;
        const int grp = kpos / span, within = kpos - grp * span, run = within / run_group, ti = grp * run_group + (within - run * run_group);
        if (ti >= n_items8) return;
        wi = ti * 8 + ((int)blockIdx.x & 7);
        chunk_lo = run * run_chunks;
        chunk_hi = (chunk_lo + run_chunks < chunk_hi) ? (chunk_lo + run_chunks) : chunk_hi;
        fscale = (n_target > 1) ? uv_run_factor(run % n_target) : 1.0;
    }
    int4 item = work[wi];
    if (item.w == 0) return;
    {
        // work lists with several copies of every tile (build_worklist `copies`): the copy index picks its own tables and plane
        const int copy = item.w >> 16;
        item.w &= 0xFFFF;
        lut = reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(lut) + (int64_t)copy * copy_lut_bytes);
        acc += (int64_t)copy * copy_acc_elems;
    }
    const int per = (chunk_hi - chunk_lo + item.w - 1) / item.w;
    const int c_beg = chunk_lo + item.z * per;
    const int c_end = (c_beg + per < chunk_hi) ? (c_beg + per) : chunk_hi;
    if (c_beg >= c_end) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, kh = lane >> 5;
    const int64_t row_w = (int64_t)item.x * X1_TILE + wr * (32 * TM), col_w = (int64_t)item.y * X1_TILE + wc * (32 * TN);
    const uint32_t *__restrict__ pa = w8 + (int64_t)kh * ncols_pad + row_w + li;
    const uint32_t *__restrict__ pb = w8 + (int64_t)kh * ncols_pad + col_w + li;
    double *__restrict__ pacc = acc + acc_off(ld, tiles_c, row_w + 4 * kh, col_w + li);
    const int64_t rs = tiles_c ? ACC_TILE : ld;    // row stride inside this wave's part of the accumulator

    f32x16 c32[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) c32[i][j][r] = 0.f;

    u32x4 Av[2][TM], Bv[2][TN];                    // two operand sets: MFMAs read one, the lookups fill the other
    uint32_t W0a[D][TM], W0b[D][TN], W1a[D][TM], W1b[D][TN];   // two banks of eight word sets

    // lookup L (0..31) of a group, in the order the MFMAs (row-major over the 4 x 4 tiles) first need the operands:
    // A0, B0, B1, B2, B3, A1, A2, A3 -- four dwords (SNP pairs) each
#define UV_ISROW(L) ((((L) >> 2) == 0) || (((L) >> 2) >= 5))
#define UV_RI(L) ((((L) >> 2) >= 5) ? ((L) >> 2) - 4 : 0)
#define UV_CI(L) (((((L) >> 2) >= 1) && (((L) >> 2) <= 4)) ? ((L) >> 2) - 1 : 0)
#define UV_AD(LA_, LB_, NS_, L, tb)                                                                        \
    (UV_ISROW(L) ? (tb) + ((LA_[NS_][UV_RI(L)] >> (8 * ((L) & 3))) & 0xFFu)                                \
                 : (tb) + ((LB_[NS_][UV_CI(L)] >> (8 * ((L) & 3))) & 0xFFu))
#define UV_RD(T_, L, a)                                                                                    \
    do {                                                                                                   \
        if (UV_ISROW(L)) Av[T_][UV_RI(L)][(L) & 3] = x1_lds32((a) + PST * ((L) & 3));                       \
        else Bv[T_][UV_CI(L)][(L) & 3] = x1_lds32((a) + PST * ((L) & 3) + 4);                               \
    } while (0)
#define UV_TABLE_ASYNC(chunk, buf)                                                                             \
    do {                                                                                                       \
        const char *src_ = reinterpret_cast<const char *>(lut) + (int64_t)(chunk) * (CHE * 8) + wave * (CHE * 2) + lane * 16; \
        char *dst_ = reinterpret_cast<char *>(&slut[buf][0]) + wave * (CHE * 2);                              \
        _Pragma("unroll") for (int t_ = 0; t_ < CHE * 2 / 1024; t_++)                                          \
            x1_lds_dma16(src_ + 1024 * t_, x1_lds_off(dst_ + 1024 * t_));                                      \
    } while (0)
    // word load number m (0..63) of a round: set m >> 3, sample group m & 7 (four row groups, four column groups)
#define UV_LOAD(YA_, YB_, g_first, m)                                                         \
    do {                                                                                      \
        const int64_t off_ = (int64_t)((g_first) + ((m) >> 3)) * 2 * ncols_pad;               \
        if (((m) & 7) < TM) YA_[(m) >> 3][((m) & 7) < TM ? ((m) & 7) : 0] = pa[off_ + 32 * ((m) & 7)]; \
        else YB_[(m) >> 3][((m) & 7) >= TM ? ((m) & 7) - TM : 0] = pb[off_ + 32 * (((m) & 7) - TM)];  \
    } while (0)
    // one group: 16 MFMAs out of operand set S_; behind each, two lookups of the NEXT group (word set NS_ of bank LA_ / LB_,
    // into operand set T_), their addresses computed one slot earlier.  LOAD_ 1 / 2: behind every MFMA two word loads of
    // the next round (bank YA_ / YB_), numbers 2 m, 2 m + 1 (+ 32 for LOAD_ == 2): all 64 go out during the first two
    // groups of a round and are first looked up in its last one.
#define UV_STEP(m, S_, T_, LA_, LB_, NS_, LOAD_, YA_, YB_, g_load, tb)                                              \
    do {                                                                                                            \
        c32[(m) >> 2][(m) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(                                             \
            (f16x8)Av[S_][(m) >> 2], (f16x8)Bv[S_][(m) & 3], c32[(m) >> 2][(m) & 3], 0, 0, 0);                       \
        UV_RD(T_, 2 * (m), a0_); UV_RD(T_, 2 * (m) + 1, a1_);                                                        \
        if ((m) < 15) { a0_ = UV_AD(LA_, LB_, NS_, 2 * (m) + 2, tb); a1_ = UV_AD(LA_, LB_, NS_, 2 * (m) + 3, tb);    \
                        asm volatile("" : "+v"(a0_), "+v"(a1_)); }   /* pins the additions HERE, not next to their reads */ \
        if (LOAD_) { UV_LOAD(YA_, YB_, g_load, 2 * (m) + 32 * ((LOAD_) - 1)); UV_LOAD(YA_, YB_, g_load, 2 * (m) + 1 + 32 * ((LOAD_) - 1)); } \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    } while (0)
#define UV_STEP4(m, ...) UV_STEP(m, __VA_ARGS__); UV_STEP((m) + 1, __VA_ARGS__); UV_STEP((m) + 2, __VA_ARGS__); UV_STEP((m) + 3, __VA_ARGS__)
#define UV_GROUP_ADDR0(S_, T_, LA_, LB_, NS_, LOAD_, YA_, YB_, g_load, tb)                                           \
    uint32_t a0_ = UV_AD(LA_, LB_, NS_, 0, tb), a1_ = UV_AD(LA_, LB_, NS_, 1, tb)
#define UV_GROUP(...)                                                                                               \
    do {                                                                                                            \
        UV_GROUP_ADDR0(__VA_ARGS__);                                                                                \
        UV_STEP4(0, __VA_ARGS__); UV_STEP4(4, __VA_ARGS__); UV_STEP4(8, __VA_ARGS__); UV_STEP4(12, __VA_ARGS__);    \
    } while (0)

    // prologue: table of the first chunk, the words of the first round, the lookups of group 0
    UV_TABLE_ASYNC(c_beg, c_beg & 1);
#define UV_L8(m) UV_LOAD(W0a, W0b, c_beg * QCH, m); UV_LOAD(W0a, W0b, c_beg * QCH, (m) + 1); UV_LOAD(W0a, W0b, c_beg * QCH, (m) + 2); UV_LOAD(W0a, W0b, c_beg * QCH, (m) + 3); \
                 UV_LOAD(W0a, W0b, c_beg * QCH, (m) + 4); UV_LOAD(W0a, W0b, c_beg * QCH, (m) + 5); UV_LOAD(W0a, W0b, c_beg * QCH, (m) + 6); UV_LOAD(W0a, W0b, c_beg * QCH, (m) + 7)
    UV_L8(0); UV_L8(8); UV_L8(16); UV_L8(24); UV_L8(32); UV_L8(40); UV_L8(48); UV_L8(56);
#undef UV_L8
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
    __syncthreads();
    uint32_t tbn = x1_lds_off(&slut[c_beg & 1][0]) + 4 * PST * kh;
    {
        uint32_t a0_, a1_;
#define UV_LK2(L) a0_ = UV_AD(W0a, W0b, 0, L, tbn); a1_ = UV_AD(W0a, W0b, 0, (L) + 1, tbn); UV_RD(0, L, a0_); UV_RD(0, (L) + 1, a1_)
#define UV_LK8(L) UV_LK2(L); UV_LK2((L) + 2); UV_LK2((L) + 4); UV_LK2((L) + 6)
        UV_LK8(0); UV_LK8(8); UV_LK8(16); UV_LK8(24);
#undef UV_LK8
#undef UV_LK2
    }
    tbn += 8 * PST;

    for (int c = c_beg; c < c_end; c++) {
        const int cur = c & 1;
        const int q0 = c * QCH;
        const int q_cnt = (q0 + QCH <= n_q) ? QCH : (n_q - q0);      // multiple of 16 (blocks are padded to 256 SNPs)
        const bool more = (c + 1 < c_end);
        if (more) UV_TABLE_ASYNC(c + 1, cur ^ 1);   // every wave is past the barrier that freed this buffer
        for (int q = 0; q < q_cnt; q += 2 * D) {
            const int g = q0 + q;
            // round A: words of bank 0, loads into bank 1
            UV_GROUP(0, 1, W0a, W0b, 1, 1, W1a, W1b, g + 8, tbn); tbn += 8 * PST;
            UV_GROUP(1, 0, W0a, W0b, 2, 2, W1a, W1b, g + 8, tbn); tbn += 8 * PST;
            UV_GROUP(0, 1, W0a, W0b, 3, 0, W1a, W1b, g + 8, tbn); tbn += 8 * PST;
            UV_GROUP(1, 0, W0a, W0b, 4, 0, W1a, W1b, g + 8, tbn); tbn += 8 * PST;
            UV_GROUP(0, 1, W0a, W0b, 5, 0, W1a, W1b, g + 8, tbn); tbn += 8 * PST;
            UV_GROUP(1, 0, W0a, W0b, 6, 0, W1a, W1b, g + 8, tbn); tbn += 8 * PST;
            UV_GROUP(0, 1, W0a, W0b, 7, 0, W1a, W1b, g + 8, tbn); tbn += 8 * PST;
            UV_GROUP(1, 0, W1a, W1b, 0, 0, W1a, W1b, g + 8, tbn); tbn += 8 * PST;
            // round B: words of bank 1, loads into bank 0
            UV_GROUP(0, 1, W1a, W1b, 1, 1, W0a, W0b, g + 16, tbn); tbn += 8 * PST;
            UV_GROUP(1, 0, W1a, W1b, 2, 2, W0a, W0b, g + 16, tbn); tbn += 8 * PST;
            UV_GROUP(0, 1, W1a, W1b, 3, 0, W0a, W0b, g + 16, tbn); tbn += 8 * PST;
            UV_GROUP(1, 0, W1a, W1b, 4, 0, W0a, W0b, g + 16, tbn); tbn += 8 * PST;
            UV_GROUP(0, 1, W1a, W1b, 5, 0, W0a, W0b, g + 16, tbn); tbn += 8 * PST;
            UV_GROUP(1, 0, W1a, W1b, 6, 0, W0a, W0b, g + 16, tbn); tbn += 8 * PST;
            UV_GROUP(0, 1, W1a, W1b, 7, 0, W0a, W0b, g + 16, tbn); tbn += 8 * PST;
            // the chunk's last group looks up the NEXT chunk's table (or, at the very end, harmlessly re-reads this one);
            // one straight-line body, as in syrk_x1_kernel
            if (q + 2 * D >= q_cnt) {
                if (more) {
                    // vmcnt is in-order: the table copy went out at the start of this (full) chunk, behind it eight rounds of
                    // 64 word loads, the last of them seven groups ago -- all but the newest 62 requests covers it (63 is
                    // the counter's ceiling and waits for nothing)
                    __builtin_amdgcn_s_waitcnt(0xCF7E); // vmcnt(62)
                    __syncthreads();
                    tbn = x1_lds_off(&slut[cur ^ 1][0]) + 4 * PST * kh;
                } else {
                    tbn = x1_lds_off(&slut[cur][0]) + 4 * PST * kh;
                }
            }
            UV_GROUP(1, 0, W0a, W0b, 0, 0, W0a, W0b, g + 16, tbn); tbn += 8 * PST;
        }
    }
    {
        double *pflush = pacc;
        asm volatile("" : "+v"(pflush));
        const int64_t rows_left = (n_rows_real > 0 ? n_rows_real : ((int64_t)1 << 40)) - (row_w + 4 * kh);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                double *__restrict__ pr = pflush + (int64_t)row * rs;
                if (row < rows_left) {
#pragma unroll
                    for (int j = 0; j < TN; j++)      // f_q x fp32 partial: exact in fp64 (13 + 24 bits)
                        (void)__builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double *)(pr + 32 * j),
                                                                      (double)c32[i][j][r] * fscale);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
#undef UV_GROUP
#undef UV_GROUP_ADDR0
#undef UV_STEP4
#undef UV_STEP
#undef UV_LOAD
#undef UV_TABLE_ASYNC
#undef UV_RD
#undef UV_AD
#undef UV_CI
#undef UV_RI
#undef UV_ISROW
}
#pragma clang diagnostic pop

// ---------------------------------------------------------------------------
// syrk_uv16_kernel (round 6): syrk_uv_kernel's arithmetic -- the same tables, words, work list, fp32 runs and fp64 flush -- on
// v_mfma_f32_16x16x32_f16.  Why: the kernel runs against the socket power cap, and what a matrix instruction costs in power is
// dominated by its accumulator traffic.  32x32x16 reads and writes 16 accumulator registers per lane for 32 768 flops, 16x16x32 four
// for 16 384: half the traffic per flop.  A register-only stream with this kernel's operand classes sustains 2100 TFLOP/s through
// 16x16x32 against 1790 through 32x32x16 on the same box (snpgpu_diag_mfma_rate, profiles/r06_probe_shapes.txt); results are
// bit-identical (the same products summed in the same order: tools/ubench/r06_kloop_ubench.hip -- so the hoped-for "one rounding
// per 32 SNPs" does not exist, the power does).
// A wave's 128 x 128 tile is 8 x 8 sub-tiles of 16 x 16 (64 x 4 = the same 256 AGPRs).  Lane l: sample l & 15 of a sub-tile, SNP
// quarter l >> 4 of a 32-SNP group = word row 4 G + (l >> 4).  Per group: 64 MFMAs, 64 lookups (one behind every MFMA), 16 words.
// Registers: sixteen 4-dword operands per group would need 128 VGPRs double-buffered; the ROW operands are therefore refilled
// in place -- row r of the 8 x 8 MFMA order is the last reader of row operand r, so row operand r - 1 of the NEXT group is looked up
// behind the MFMAs of row r (operand 7 behind row 0 of the group that uses it) -- and only the column operands have two sets:
// 96 VGPRs of operands + a ring of four word sets (64): the words of group g + 4 are requested behind the first 16 MFMAs of group g,
// into the set group g has just finished with (its one remaining use, the word of row operand 7, is copied out first), and are
// first looked up in group g + 3.
// LDS banks: a 32-lane pass of a lookup now spans TWO quarters, i.e. two pair tables with the same bank mapping (entry c of
// every table sits in banks 2 c, 2 c + 1).  The table builders therefore swap the halves of the entries of odd quarters
// ({column pair, row pair}; uv_tables_kernel / homo_uv_tables_kernel, `swap_odd`): a row lookup reads bank 2 c in even quarters
// and 2 c + 1 in odd ones, a column lookup the other way round -- conflict-free again.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Warray-bounds"
__global__ __launch_bounds__(256, 1) void syrk_uv16_kernel(
    const uint32_t *__restrict__ w8, int64_t ncols_pad, const uint2 *__restrict__ lut, int n_q,
    double *__restrict__ acc, int64_t ld, int64_t tiles_c, const int4 *__restrict__ work,
    const unsigned long long *__restrict__ d_missing, int64_t n_rows_real, int chunk_lo, int chunk_hi, double fscale,
    int n_runs, int run_chunks, int n_target, int run_group, int n_items8, int run_if_missing, int64_t copy_lut_bytes,
    int64_t copy_acc_elems)
{
    if ((*d_missing != 0ull) != (run_if_missing != 0)) return;
    constexpr int TS = 8, D = 4;
    constexpr int CHS = UV_CHS;                    // SNPs per table chunk
    constexpr int PST = 128;                       // bytes of table per SNP pair: 16 entries of 8 bytes
    constexpr int CHE = (CHS / 2) * PST / 8;       // 8-byte units per chunk: 64 KiB
    constexpr int GCH = CHS / 32;                  // 32-SNP groups per chunk
    constexpr int GST = 16 * PST;                  // bytes of table per group
    static_assert(GCH % (2 * D) == 0, "whole double rounds of the word banks per chunk");
    __shared__ uint2 slut[2][CHE];

    int wi = blockIdx.x;
    if (n_runs > 1) {                              // fused (tile, run) launch: see syrk_uv_kernel
        const int kpos = (int)blockIdx.x >> 3, span = run_group * n_runs;
        const int grp = kpos / span, within = kpos - grp * span, run = within / run_group, ti = grp * run_group + (within - run * run_group);
        if (ti >= n_items8) return;
        wi = ti * 8 + ((int)blockIdx.x & 7);
        chunk_lo = run * run_chunks;
        chunk_hi = (chunk_lo + run_chunks < chunk_hi) ? (chunk_lo + run_chunks) : chunk_hi;
        fscale = (n_target > 1) ? uv_run_factor(run % n_target) : 1.0;
    }
    int4 item = work[wi];
    if (item.w == 0) return;
    {
        const int copy = item.w >> 16;
        item.w &= 0xFFFF;
        lut = reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(lut) + (int64_t)copy * copy_lut_bytes);
        acc += (int64_t)copy * copy_acc_elems;
    }
    const int per = (chunk_hi - chunk_lo + item.w - 1) / item.w;
    const int c_beg = chunk_lo + item.z * per;
    const int c_end = (c_beg + per < chunk_hi) ? (c_beg + per) : chunk_hi;
    if (c_beg >= c_end) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l16 = lane & 15, kq = lane >> 4;
    const int64_t row_w = (int64_t)item.x * X1_TILE + wr * (16 * TS), col_w = (int64_t)item.y * X1_TILE + wc * (16 * TS);
    // word loads: uniform row base (SGPRs, per group) + a 32-bit lane offset + an immediate -- no address arithmetic on the VALU
    const int la = (int)((int64_t)kq * ncols_pad + row_w + l16), lb = (int)((int64_t)kq * ncols_pad + col_w + l16);
    double *__restrict__ pacc = acc + acc_off(ld, tiles_c, row_w + 4 * kq, col_w + l16);
    const int64_t rs = tiles_c ? ACC_TILE : ld;

    f32x4 c16[TS][TS];
#pragma unroll
    for (int i = 0; i < TS; i++)
#pragma unroll
        for (int j = 0; j < TS; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) c16[i][j][r] = 0.f;

    u32x4 Av[TS], Bv[2][TS];                       // row operands: ONE set, refilled in place; column operands: two sets
    uint32_t Wa[D][TS], Wb[D][TS];                 // ring of four word sets (8 row + 8 column words each): group g lives in set g & 3
    uint32_t wa7;                                  // this group's word of row operand 7 (its set is being refilled for group g + 4)
    uint32_t tc_row, tn_row, tn_col;               // table positions: this group's (row half), the next group's (row / column half)

    // lookup L (0..63) of a group, issued behind MFMA L (row r = L >> 3 of the 8 x 8 order, t = L & 7):
    //   t < 4:  dword t of row operand (r == 0 ? 7 of THIS group : r - 1 of the NEXT group)
    //   t >= 4: dword t - 4 of column operand r of the next group (set T_)
#define U16_R(L) ((L) >> 3)
#define U16_ISA(L) (((L) & 7) < 4)
#define U16_AI(L) (U16_R(L) == 0 ? 7 : U16_R(L) - 1)
#define U16_D(L) ((L) & 3)
#define U16_AD(NS_, L)                                                                                       \
    (U16_ISA(L) ? (U16_R(L) == 0 ? tc_row + ((wa7 >> (8 * U16_D(L))) & 0xFFu)                                 \
                                 : tn_row + ((Wa[NS_][U16_AI(L)] >> (8 * U16_D(L))) & 0xFFu))                 \
                : tn_col + ((Wb[NS_][U16_R(L)] >> (8 * U16_D(L))) & 0xFFu))
#define U16_RD(T_, L, a)                                                                                     \
    do {                                                                                                     \
        if (U16_ISA(L)) Av[U16_AI(L)][U16_D(L)] = x1_lds32((a) + PST * U16_D(L));                             \
        else Bv[T_][U16_R(L)][U16_D(L)] = x1_lds32((a) + PST * U16_D(L));                                     \
    } while (0)
#define U16_TABLE_ASYNC(chunk, buf)                                                                            \
    do {                                                                                                       \
        const char *src_ = reinterpret_cast<const char *>(lut) + (int64_t)(chunk) * (CHE * 8) + wave * (CHE * 2) + lane * 16; \
        char *dst_ = reinterpret_cast<char *>(&slut[buf][0]) + wave * (CHE * 2);                              \
        _Pragma("unroll") for (int t_ = 0; t_ < CHE * 2 / 1024; t_++)                                          \
            x1_lds_dma16(src_ + 1024 * t_, x1_lds_off(dst_ + 1024 * t_));                                      \
    } while (0)
    // word load number m (0..15) of a group into set CS_: eight row sub-tiles, eight column sub-tiles
#define U16_LOAD(CS_, g_abs, m)                                                               \
    do {                                                                                      \
        const uint32_t *__restrict__ bs_ = w8 + (int64_t)(g_abs) * 4 * ncols_pad;              \
        if ((m) < TS) Wa[CS_][(m) < TS ? (m) : 0] = bs_[la + 16 * (m)];                        \
        else Wb[CS_][(m) >= TS ? (m) - TS : 0] = bs_[lb + 16 * ((m) - TS)];                    \
    } while (0)
#define U16_MFMA(m, S_)                                                                                             \
        c16[(m) >> 3][(m) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                                             \
            (f16x8)Av[(m) >> 3], (f16x8)Bv[S_][(m) & 7], c16[(m) >> 3][(m) & 7], 0, 0, 0)
    // Issue pattern (measured, profiles/r06_uv16_patterns.txt; ms per 65 536-SNP step at N = 100 000 on one box, the 32x32x16 kernel 456):
    // one lookup + one address op behind every MFMA 489; MFMAs in runs of 4 / 8 with their lookups behind 486 / 540; exactly TWO
    // companions of ONE kind behind every MFMA -- [M dd][M aa] 430, [M aa][M dd] a little better again: a lone wave pays for every
    // switch between the matrix pipe, the LDS and the VALU, and a 16-clock MFMA hides two instructions, not three.  The addresses of
    // a batch of four lookups are computed one batch ahead into the other half of eight address registers.
#define U16_SB() __builtin_amdgcn_sched_barrier(0)
#define U16_M(m, S_) do { U16_MFMA(m, S_); U16_SB(); } while (0)
#define U16_A2(NS_, m, k)      /* addresses of lookups m + 4 + k, + 1 (the NEXT batch) into the other register half */              \
    do {                                                                                                                            \
        if ((m) + 4 + (k) < 64) {                                                                                                   \
            a_[4 * ((((m) >> 2) + 1) & 1) + (k)] = U16_AD(NS_, ((m) + 4 + (k)) & 63);                                               \
            a_[4 * ((((m) >> 2) + 1) & 1) + (k) + 1] = U16_AD(NS_, ((m) + 5 + (k)) & 63);                                           \
            asm volatile("" : "+v"(a_[4 * ((((m) >> 2) + 1) & 1) + (k)]), "+v"(a_[4 * ((((m) >> 2) + 1) & 1) + (k) + 1]));           \
        }                                                                                                                           \
        U16_SB();                                                                                                                   \
    } while (0)
#define U16_D2(T_, m, k)       /* lookups m + k, + 1 of THIS batch */                                                               \
    do {                                                                                                                            \
        U16_RD(T_, (m) + (k), a_[4 * (((m) >> 2) & 1) + (k)]); U16_RD(T_, (m) + (k) + 1, a_[4 * (((m) >> 2) & 1) + (k) + 1]);        \
        U16_SB();                                                                                                                   \
    } while (0)
#define U16_STEP4(m, S_, T_, CS_, NS_, g_abs)                                                                       \
    do {                                                                                                            \
        U16_M(m, S_);       U16_A2(NS_, m, 0);                                                                      \
        U16_M((m) + 1, S_); U16_D2(T_, m, 0);                                                                       \
        U16_M((m) + 2, S_); U16_A2(NS_, m, 2);                                                                      \
        U16_M((m) + 3, S_); U16_D2(T_, m, 2);                                                                       \
        if ((m) < 16) {     /* the words of group g + 4 into this group's set, four behind each of the first four batches */ \
            U16_LOAD(CS_, (g_abs) + D, m); U16_LOAD(CS_, (g_abs) + D, (m) + 1); U16_LOAD(CS_, (g_abs) + D, (m) + 2); U16_LOAD(CS_, (g_abs) + D, (m) + 3); \
            U16_SB();                                                                                               \
        }                                                                                                           \
    } while (0)
#define U16_STEP8(m, ...) U16_STEP4(m, __VA_ARGS__); U16_STEP4((m) + 4, __VA_ARGS__)
    // one 32-SNP group (absolute index g_abs, word set CS_ = g_abs & 3, the next group's NS_); afterwards the table positions move on
#define U16_GROUP(S_, T_, CS_, NS_, g_abs)                                                                          \
    do {                                                                                                            \
        wa7 = Wa[CS_][7];                                                                                           \
        asm volatile("" : "+v"(wa7));                                                                               \
        uint32_t a_[8];                                                                                             \
        a_[0] = U16_AD(NS_, 0); a_[1] = U16_AD(NS_, 1); a_[2] = U16_AD(NS_, 2); a_[3] = U16_AD(NS_, 3);             \
        U16_STEP8(0, S_, T_, CS_, NS_, g_abs);  U16_STEP8(8, S_, T_, CS_, NS_, g_abs);                               \
        U16_STEP8(16, S_, T_, CS_, NS_, g_abs); U16_STEP8(24, S_, T_, CS_, NS_, g_abs);                              \
        U16_STEP8(32, S_, T_, CS_, NS_, g_abs); U16_STEP8(40, S_, T_, CS_, NS_, g_abs);                              \
        U16_STEP8(48, S_, T_, CS_, NS_, g_abs); U16_STEP8(56, S_, T_, CS_, NS_, g_abs);                              \
        tc_row = tn_row; tn_row += GST; tn_col += GST;                                                              \
    } while (0)

    // prologue: table of the first chunk, the words of the first four groups, the lookups of group 0 (row operand 7 comes with row 0)
    U16_TABLE_ASYNC(c_beg, c_beg & 1);
#define U16_L16(S) U16_LOAD(S, c_beg * GCH + S, 0); U16_LOAD(S, c_beg * GCH + S, 1); U16_LOAD(S, c_beg * GCH + S, 2); U16_LOAD(S, c_beg * GCH + S, 3);     \
                   U16_LOAD(S, c_beg * GCH + S, 4); U16_LOAD(S, c_beg * GCH + S, 5); U16_LOAD(S, c_beg * GCH + S, 6); U16_LOAD(S, c_beg * GCH + S, 7);     \
                   U16_LOAD(S, c_beg * GCH + S, 8); U16_LOAD(S, c_beg * GCH + S, 9); U16_LOAD(S, c_beg * GCH + S, 10); U16_LOAD(S, c_beg * GCH + S, 11);   \
                   U16_LOAD(S, c_beg * GCH + S, 12); U16_LOAD(S, c_beg * GCH + S, 13); U16_LOAD(S, c_beg * GCH + S, 14); U16_LOAD(S, c_beg * GCH + S, 15)
    U16_L16(0); U16_L16(1); U16_L16(2); U16_L16(3);
#undef U16_L16
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
    __syncthreads();
    {
        // odd quarters read the row half of an entry at + 4 and the column half at + 0 (swapped entries, see the header)
        const uint32_t base = x1_lds_off(&slut[c_beg & 1][0]) + 4 * PST * kq;
        tn_row = base + 4 * (kq & 1);
        tn_col = base + 4 - 4 * (kq & 1);
        tc_row = tn_row;
    }
    {
        uint32_t a_;
        // group 0: row operands 0..6 and the eight column operands (set 0) from word set 0
#define U16_PA(i, d) a_ = tn_row + ((Wa[0][i] >> (8 * (d))) & 0xFFu); Av[i][d] = x1_lds32(a_ + PST * (d))
#define U16_PB(j, d) a_ = tn_col + ((Wb[0][j] >> (8 * (d))) & 0xFFu); Bv[0][j][d] = x1_lds32(a_ + PST * (d))
#define U16_P4(M, i) M(i, 0); M(i, 1); M(i, 2); M(i, 3)
        U16_P4(U16_PA, 0); U16_P4(U16_PA, 1); U16_P4(U16_PA, 2); U16_P4(U16_PA, 3); U16_P4(U16_PA, 4); U16_P4(U16_PA, 5); U16_P4(U16_PA, 6);
        U16_P4(U16_PB, 0); U16_P4(U16_PB, 1); U16_P4(U16_PB, 2); U16_P4(U16_PB, 3); U16_P4(U16_PB, 4); U16_P4(U16_PB, 5); U16_P4(U16_PB, 6); U16_P4(U16_PB, 7);
#undef U16_P4
#undef U16_PB
#undef U16_PA
    }
    tn_row += GST; tn_col += GST;                  // (tc_row stays on group 0: its row operand 7 is looked up behind row 0)

    for (int c = c_beg; c < c_end; c++) {
        const int cur = c & 1;
        const int q_cnt = (c * (CHS / 16) + CHS / 16 <= n_q) ? GCH : (n_q - c * (CHS / 16)) / 2;   // 32-SNP groups: a multiple of 8
        const bool more = (c + 1 < c_end);
        if (more) U16_TABLE_ASYNC(c + 1, cur ^ 1);  // every wave is past the barrier that freed this buffer
        for (int q = 0; q < q_cnt; q += 2 * D) {
            const int g = c * GCH + q;
            U16_GROUP(0, 1, 0, 1, g);
            U16_GROUP(1, 0, 1, 2, g + 1);
            U16_GROUP(0, 1, 2, 3, g + 2);
            U16_GROUP(1, 0, 3, 0, g + 3);
            U16_GROUP(0, 1, 0, 1, g + 4);
            U16_GROUP(1, 0, 1, 2, g + 5);
            U16_GROUP(0, 1, 2, 3, g + 6);
            // the chunk's last group looks up the NEXT chunk's first group (or, at the very end, harmlessly re-reads this chunk);
            // its own row operand 7 still comes from this chunk (tc_row)
            if (q + 2 * D >= q_cnt) {
                uint32_t base;
                if (more) {
                    // vmcnt is in-order: the table copy went out at the start of this (full) chunk, behind it 31 groups of 16 word
                    // loads -- all but the newest 62 requests covers it
                    __builtin_amdgcn_s_waitcnt(0xCF7E); // vmcnt(62)
                    __syncthreads();
                    base = x1_lds_off(&slut[cur ^ 1][0]) + 4 * PST * kq;
                } else {
                    base = x1_lds_off(&slut[cur][0]) + 4 * PST * kq;
                }
                tn_row = base + 4 * (kq & 1);
                tn_col = base + 4 - 4 * (kq & 1);
            }
            U16_GROUP(1, 0, 3, 0, g + 7);
        }
    }
    {
        double *pflush = pacc;
        asm volatile("" : "+v"(pflush));
        const int64_t rows_left = (n_rows_real > 0 ? n_rows_real : ((int64_t)1 << 40)) - (row_w + 4 * kq);
#pragma unroll
        for (int i = 0; i < TS; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = i * 16 + r;
                double *__restrict__ pr = pflush + (int64_t)row * rs;
                if (row < rows_left) {
#pragma unroll
                    for (int j = 0; j < TS; j++)      // f_q x fp32 partial: exact in fp64 (13 + 24 bits)
                        (void)__builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double *)(pr + 16 * j),
                                                                      (double)c16[i][j][r] * fscale);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
#undef U16_GROUP
#undef U16_STEP8
#undef U16_STEP4
#undef U16_D2
#undef U16_A2
#undef U16_M
#undef U16_SB
#undef U16_MFMA
#undef U16_LOAD
#undef U16_TABLE_ASYNC
#undef U16_RD
#undef U16_AD
#undef U16_D
#undef U16_AI
#undef U16_ISA
#undef U16_R
}
#pragma clang diagnostic pop

// ---------------------------------------------------------------------------
// syrk_uv16c_kernel (round 6, SNPGPU_SYRK_UV16=2): syrk_uv16_kernel with the operands CONVERTED instead of looked up.  The pair bytes
// of a block without missing calls hold two e2m1 nibbles c0 | c1 << 4 (value c / 2; transpose8_kernel, nibble_nomiss), ONE
// v_cvt_scalef32_pk_f16_fp4 (byte select by op_sel) turns a byte into the fp16 pair (c0 / 2, c1 / 2) and ONE v_pk_fma_f16 with the
// lane's factor pairs makes (c / 2)(2 u) - c_a u = (c - c_a) u: exact at every step, the same operand values as the tables'.  Per
// operand dword two vector ops instead of an address op + a ds_read_b32; per 32-SNP group four ds_read_b128 of factors (256 bytes per
// group: uv_tables_kernel, swap_odd == 2) instead of 64 table reads; LDS 16 KiB instead of 128.  K-loop model
// (tools/ubench/r06_kloop_ubench.hip, E against F): 20.5 against 22.6 us per 1024 SNPs of a wave tile.
// Same MFMA order, register plan (row operands refilled in place, two column sets, ring of four word sets), work list, runs, flush.
// MEASURED (configs[2], interleaved on one box, profiles/r06_uvc_ab.txt): the kernel is bound by the socket power cap, not by issue slots --
// the converted operands alone (SNPGPU_SYRK_UV16=2) need 5 % fewer cycles and run at a 5 % lower clock: 432 against 431 ms per step.  With the
// runs walked inside and 35 of a wave's 64 sub-tile sums carried in LDS (=3) the panel writes fall from 242 to 131 GB per step (32: 144).
// THE PACE-MAKER.  L2 -> fabric reads (TCC_EA0_RDREQ x 128 B; the fp64 atomics leave as EA atomic writes and fetch nothing) are all genotype
// word lines: the workgroups demand 1.28 TB of them per step from their L2s, and what they fetch depends on whether the 32 workgroups of an
// XCD stream the word rows they share IN STEP.  The lookup kernel's do (391 GB: a line serves ~3.3 workgroups, the 4 x 4 super-tiles'
// sharing); this kernel's, left alone, drift apart (727 GB, 947 GB with the runs walked inside).  What keeps the lookup kernel's in step is
// its table: every workgroup fetches the same 64 KiB per chunk; the first to arrive misses, and because vmcnt counts in order its word loads
// wait behind that fetch, while the followers' fetches hit -- the leader is held back one memory latency per chunk.  This kernel therefore
// issues the same fetch as a PACE-MAKER: 16 x 1 KiB per wave and chunk (= the table's size) from a zero-filled region common to all
// workgroups (uvpace), into an LDS slot nobody reads.  That brings the reads to 345 - 407 GB; 4 or 1 KiB per wave do nothing (1030 / 946 GB).
// With it this form takes 414 - 416 against 424 - 426 ms of kernel time per step (-2.4 %, at 2158 against 2136 MHz under the same 1370 W;
// -2.0 ... -2.8 % on a second box) and moves 489 instead of 633 GB: the default since the end of round 6 (SNPGPU_SYRK_UV16=1: the lookup
// kernel; SNPGPU_UVC_PACE=0: no pace-maker).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const volatile u32x4 x1_lds_u128;
__device__ __forceinline__ u32x4 x1_lds128(uint32_t off)
{
    return *(x1_lds_u128 *)(uintptr_t)off;
}
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Warray-bounds"
__global__ __launch_bounds__(256, 1) void syrk_uv16c_kernel(
    const uint32_t *__restrict__ w8, int64_t ncols_pad, const uint2 *__restrict__ lut, int n_q,
    double *__restrict__ acc, int64_t ld, int64_t tiles_c, const int4 *__restrict__ work,
    const unsigned long long *__restrict__ d_missing, int64_t n_rows_real, int chunk_lo, int chunk_hi, double fscale,
    int n_runs, int run_chunks, int n_target, int run_group, int n_items8, int run_if_missing, const char *__restrict__ pace_src, int pace)
{
    if ((*d_missing != 0ull) != (run_if_missing != 0)) return;
    constexpr int TS = 8, D = 4;
    constexpr int CHS = UV_CHS;                    // slots per factor chunk
    constexpr int GCH = CHS / 32;                  // 32-SNP groups per chunk
    constexpr int GST = 256;                       // bytes of factors per group: {row 2u, row -c u, column 2v, column -c v} x 4 quarters x 4 pairs
    constexpr int CHB = GCH * GST;                 // 8 KiB per chunk
    static_assert(GCH % (2 * D) == 0, "whole double rounds of the word banks per chunk");
    __shared__ u32x4 sfac[2][CHB / 16];
    // run_group == 0 with n_runs > 1 (SNPGPU_SYRK_UV16=3): a work item is a TILE and walks its fp32 runs itself.  With the tables gone
    // 144 KiB of LDS are free: the sums of CARRY_SUB of a wave's 64 sub-tiles stay there between runs as fp32 (carry += f_q x partial;
    // six additions of 24-bit numbers: 3e-7 of a run's scale against the 5e-6 of the run itself) and meet the fp64 panel ONCE per block;
    // the other sub-tiles flush after every run as before.  Half the fp64 read-modify-writes of the 40 GB panel per run go away.
    constexpr int CARRY_SUB = 35;                  // sub-tiles 0 .. 34 in (i, j) order: 35 KiB per wave = all the LDS there is (16 + 4 + 140 KiB)
    __shared__ f32x4 scar[4][CARRY_SUB * 64];
    __shared__ u32x4 space[4][64];                 // 1 KiB per wave: where the pace-maker fetches land (never read)
    const bool inner = (n_runs > 1 && run_group == 0);

    int wi = blockIdx.x;
    if (n_runs > 1 && !inner) {                    // fused (tile, run) launch: see syrk_uv_kernel
        const int kpos = (int)blockIdx.x >> 3, span = run_group * n_runs;
        const int grp = kpos / span, within = kpos - grp * span, run = within / run_group, ti = grp * run_group + (within - run * run_group);
        if (ti >= n_items8) return;
        wi = ti * 8 + ((int)blockIdx.x & 7);
        chunk_lo = run * run_chunks;
        chunk_hi = (chunk_lo + run_chunks < chunk_hi) ? (chunk_lo + run_chunks) : chunk_hi;
        fscale = (n_target > 1) ? uv_run_factor(run % n_target) : 1.0;
    }
    int4 item = work[wi];
    if (item.w == 0) return;
    item.w &= 0xFFFF;                              // (no table copies in this form: GRM / PCA contexts only)
    const int runs_here = inner ? n_runs : 1;
    const bool carry_on = inner && item.w == 1;    // (a tile whose K range is split over several workgroups flushes every run)
    const int all_lo = chunk_lo, all_hi = chunk_hi;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l16 = lane & 15, kq = lane >> 4;
    const int64_t row_w = (int64_t)item.x * X1_TILE + wr * (16 * TS), col_w = (int64_t)item.y * X1_TILE + wc * (16 * TS);
    const int la = (int)((int64_t)kq * ncols_pad + row_w + l16), lb = (int)((int64_t)kq * ncols_pad + col_w + l16);
    double *__restrict__ pacc = acc + acc_off(ld, tiles_c, row_w + 4 * kq, col_w + l16);
    const int64_t rs = tiles_c ? ACC_TILE : ld;

    f32x4 c16[TS][TS];
#pragma unroll
    for (int i = 0; i < TS; i++)
#pragma unroll
        for (int j = 0; j < TS; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) c16[i][j][r] = 0.f;

    for (int run = 0; run < runs_here; run++) {
    if (inner) {
        chunk_lo = run * run_chunks;
        chunk_hi = (chunk_lo + run_chunks < all_hi) ? (chunk_lo + run_chunks) : all_hi;
        fscale = (n_target > 1) ? uv_run_factor(run % n_target) : 1.0;
    } else { chunk_lo = all_lo; chunk_hi = all_hi; }
    const int per = (chunk_hi - chunk_lo + item.w - 1) / item.w;
    const int c_beg = chunk_lo + item.z * per;
    const int c_end = (c_beg + per < chunk_hi) ? (c_beg + per) : chunk_hi;
    if (c_beg >= c_end) continue;                  // (uniform over the workgroup; never with carry_on)

    u32x4 Av[TS], Bv[2][TS];                       // row operands: ONE set, refilled in place; column operands: two sets
    uint32_t Wa[D][TS], Wb[D][TS];                 // ring of four word sets (8 row + 8 column words each): group g lives in set g & 3
    uint32_t wa7;                                  // this group's word of row operand 7 (its set is being refilled for group g + 4)
    u32x4 RF1[2], RF0[2];                          // row factors {2 u}, {-c_a u} of the lane's four pairs: group parity g & 1
    u32x4 CF1, CF0;                                // column factors of the NEXT group
    uint32_t fn;                                   // LDS position of the next group's factors (this lane's quarter)

    // conversion L (0..63) of a group, issued around MFMA L (row r = L >> 3 of the 8 x 8 order, t = L & 7):
    //   t < 4:  dword t of row operand (r == 0 ? 7 of THIS group : r - 1 of the NEXT group)
    //   t >= 4: dword t - 4 of column operand r of the next group (set T_)
#define C16_R(L) ((L) >> 3)
#define C16_ISA(L) (((L) & 7) < 4)
#define C16_AI(L) (C16_R(L) == 0 ? 7 : C16_R(L) - 1)
#define C16_D(L) ((L) & 3)
#define C16_WORD(NS_, L) (C16_ISA(L) ? (C16_R(L) == 0 ? wa7 : Wa[NS_][C16_AI(L)]) : Wb[NS_][C16_R(L)])
#define C16_CVT(NS_, L) __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(C16_WORD(NS_, L), 1.0f, C16_D(L)))
    // P_ = parity of THIS group: its own row factors RF[P_] serve operand 7, the next group's RF[P_ ^ 1] operands 0..6
#define C16_F1(P_, L) (C16_ISA(L) ? RF1[C16_R(L) == 0 ? (P_) : (P_) ^ 1][C16_D(L)] : CF1[C16_D(L)])
#define C16_F0(P_, L) (C16_ISA(L) ? RF0[C16_R(L) == 0 ? (P_) : (P_) ^ 1][C16_D(L)] : CF0[C16_D(L)])
#define C16_FMA(T_, P_, L, x)                                                                                \
    do {                                                                                                     \
        const f16x2 y_ = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, (uint32_t)(x)), __builtin_bit_cast(f16x2, (uint32_t)(C16_F1(P_, L))), \
                                                   __builtin_bit_cast(f16x2, (uint32_t)(C16_F0(P_, L))));     \
        if (C16_ISA(L)) Av[C16_AI(L)][C16_D(L)] = __builtin_bit_cast(uint32_t, y_);                           \
        else Bv[T_][C16_R(L)][C16_D(L)] = __builtin_bit_cast(uint32_t, y_);                                   \
    } while (0)
#define C16_TABLE_ASYNC(chunk, buf)                                                                            \
    do {                                                                                                       \
        const char *src_ = reinterpret_cast<const char *>(lut) + (int64_t)(chunk) * CHB + wave * (CHB / 4) + lane * 16; \
        char *dst_ = reinterpret_cast<char *>(&sfac[buf][0]) + wave * (CHB / 4);                              \
        _Pragma("unroll") for (int t_ = 0; t_ < CHB / 4 / 1024; t_++)                                          \
            x1_lds_dma16(src_ + 1024 * t_, x1_lds_off(dst_ + 1024 * t_));                                      \
        /* the pace-maker: 16 more KiB per wave from a region every workgroup reads for this chunk.  (Unrolled on purpose: as a loop  \
           with a run-time count the compiler drains vmcnt at its back edge, every iteration waits for all word loads in flight, and  \
           the workgroups drift as if there were no pace-maker: 1043 against 407 GB of word fetches, + 2 % instead of - 2 %.) */    \
        if (pace) {                                                                                            \
            const char *ps_ = pace_src + (int64_t)(chunk) * 65536 + wave * 16384 + lane * 16;                 \
            _Pragma("unroll") for (int t_ = 0; t_ < 16; t_++) x1_lds_dma16(ps_ + 1024 * t_, x1_lds_off(&space[wave][0])); \
        }                                                                                                      \
    } while (0)
#define C16_LOAD(CS_, g_abs, m)                                                               \
    do {                                                                                      \
        const uint32_t *__restrict__ bs_ = w8 + (int64_t)(g_abs) * 4 * ncols_pad;              \
        if ((m) < TS) Wa[CS_][(m) < TS ? (m) : 0] = bs_[la + 16 * (m)];                        \
        else Wb[CS_][(m) >= TS ? (m) - TS : 0] = bs_[lb + 16 * ((m) - TS)];                    \
    } while (0)
#define C16_MFMA(m, S_)                                                                                             \
        c16[(m) >> 3][(m) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                                             \
            (f16x8)Av[(m) >> 3], (f16x8)Bv[S_][(m) & 7], c16[(m) >> 3][(m) & 7], 0, 0, 0)
    // issue pattern: as syrk_uv16_kernel's -- two companions of ONE kind behind every MFMA: [M cc][M ff], the conversions of a batch of
    // four one batch ahead of their fmas
#define C16_SB() __builtin_amdgcn_sched_barrier(0)
#define C16_M(m, S_) do { C16_MFMA(m, S_); C16_SB(); } while (0)
#define C16_C2(NS_, m, k)      /* conversions m + 4 + k, + 1 (the NEXT batch) into the other register half */                       \
    do {                                                                                                                            \
        if ((m) + 4 + (k) < 64) {                                                                                                   \
            x_[4 * ((((m) >> 2) + 1) & 1) + (k)] = C16_CVT(NS_, ((m) + 4 + (k)) & 63);                                              \
            x_[4 * ((((m) >> 2) + 1) & 1) + (k) + 1] = C16_CVT(NS_, ((m) + 5 + (k)) & 63);                                          \
            asm volatile("" : "+v"(x_[4 * ((((m) >> 2) + 1) & 1) + (k)]), "+v"(x_[4 * ((((m) >> 2) + 1) & 1) + (k) + 1]));           \
        }                                                                                                                           \
        C16_SB();                                                                                                                   \
    } while (0)
#define C16_F2(T_, P_, m, k)   /* fmas m + k, + 1 of THIS batch */                                                                  \
    do {                                                                                                                            \
        C16_FMA(T_, P_, (m) + (k), x_[4 * (((m) >> 2) & 1) + (k)]); C16_FMA(T_, P_, (m) + (k) + 1, x_[4 * (((m) >> 2) & 1) + (k) + 1]); \
        C16_SB();                                                                                                                   \
    } while (0)
#define C16_STEP4(m, S_, T_, CS_, NS_, P_, g_abs)                                                                   \
    do {                                                                                                            \
        C16_M(m, S_);       C16_C2(NS_, m, 0);                                                                      \
        C16_M((m) + 1, S_); C16_F2(T_, P_, m, 0);                                                                   \
        C16_M((m) + 2, S_); C16_C2(NS_, m, 2);                                                                      \
        C16_M((m) + 3, S_); C16_F2(T_, P_, m, 2);                                                                   \
        if ((m) < 16) {     /* the words of group g + 4 into this group's set, four behind each of the first four batches */ \
            C16_LOAD(CS_, (g_abs) + D, m); C16_LOAD(CS_, (g_abs) + D, (m) + 1); C16_LOAD(CS_, (g_abs) + D, (m) + 2); C16_LOAD(CS_, (g_abs) + D, (m) + 3); \
            C16_SB();                                                                                               \
        }                                                                                                           \
    } while (0)
#define C16_STEP8(m, ...) C16_STEP4(m, __VA_ARGS__); C16_STEP4((m) + 4, __VA_ARGS__)
    // one 32-SNP group (absolute index g_abs, parity P_, word set CS_ = g_abs & 3, the next group's NS_): the next group's factors are
    // requested first (row pairs first used behind MFMA 8, column pairs behind MFMA 4)
#define C16_GROUP(S_, T_, CS_, NS_, P_, g_abs)                                                                      \
    do {                                                                                                            \
        wa7 = Wa[CS_][7];                                                                                           \
        asm volatile("" : "+v"(wa7));                                                                               \
        CF1 = x1_lds128(fn + 128); CF0 = x1_lds128(fn + 192);                                                       \
        RF1[(P_) ^ 1] = x1_lds128(fn); RF0[(P_) ^ 1] = x1_lds128(fn + 64);                                          \
        uint32_t x_[8];                                                                                             \
        x_[0] = C16_CVT(NS_, 0); x_[1] = C16_CVT(NS_, 1); x_[2] = C16_CVT(NS_, 2); x_[3] = C16_CVT(NS_, 3);         \
        C16_STEP8(0, S_, T_, CS_, NS_, P_, g_abs);  C16_STEP8(8, S_, T_, CS_, NS_, P_, g_abs);                       \
        C16_STEP8(16, S_, T_, CS_, NS_, P_, g_abs); C16_STEP8(24, S_, T_, CS_, NS_, P_, g_abs);                      \
        C16_STEP8(32, S_, T_, CS_, NS_, P_, g_abs); C16_STEP8(40, S_, T_, CS_, NS_, P_, g_abs);                      \
        C16_STEP8(48, S_, T_, CS_, NS_, P_, g_abs); C16_STEP8(56, S_, T_, CS_, NS_, P_, g_abs);                      \
        fn += GST;                                                                                                  \
    } while (0)

    // prologue: factors of the first chunk, the words of the first four groups, the operands of group 0 (row operand 7 comes with row 0)
    C16_TABLE_ASYNC(c_beg, c_beg & 1);
#define C16_L16(S) C16_LOAD(S, c_beg * GCH + S, 0); C16_LOAD(S, c_beg * GCH + S, 1); C16_LOAD(S, c_beg * GCH + S, 2); C16_LOAD(S, c_beg * GCH + S, 3);     \
                   C16_LOAD(S, c_beg * GCH + S, 4); C16_LOAD(S, c_beg * GCH + S, 5); C16_LOAD(S, c_beg * GCH + S, 6); C16_LOAD(S, c_beg * GCH + S, 7);     \
                   C16_LOAD(S, c_beg * GCH + S, 8); C16_LOAD(S, c_beg * GCH + S, 9); C16_LOAD(S, c_beg * GCH + S, 10); C16_LOAD(S, c_beg * GCH + S, 11);   \
                   C16_LOAD(S, c_beg * GCH + S, 12); C16_LOAD(S, c_beg * GCH + S, 13); C16_LOAD(S, c_beg * GCH + S, 14); C16_LOAD(S, c_beg * GCH + S, 15)
    C16_L16(0); C16_L16(1); C16_L16(2); C16_L16(3);
#undef C16_L16
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
    __syncthreads();
    fn = x1_lds_off(&sfac[c_beg & 1][0]) + 16 * kq;
    {
        // group 0: row operands 0..6 and the eight column operands (set 0) from word set 0 with group 0's factors
        RF1[0] = x1_lds128(fn); RF0[0] = x1_lds128(fn + 64);
        CF1 = x1_lds128(fn + 128); CF0 = x1_lds128(fn + 192);      // (group 0's columns: the loop's first group replaces them with group 1's)
#define C16_PO(W, d, F1_, F0_) __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(W, 1.0f, d)), \
                                   __builtin_bit_cast(f16x2, (uint32_t)F1_[d]), __builtin_bit_cast(f16x2, (uint32_t)F0_[d])))      /* (the casts matter: __builtin_bit_cast of a vector-ELEMENT lvalue reads element 0) */
#define C16_PA(i) Av[i][0] = C16_PO(Wa[0][i], 0, RF1[0], RF0[0]); Av[i][1] = C16_PO(Wa[0][i], 1, RF1[0], RF0[0]); \
                  Av[i][2] = C16_PO(Wa[0][i], 2, RF1[0], RF0[0]); Av[i][3] = C16_PO(Wa[0][i], 3, RF1[0], RF0[0])
#define C16_PB(j) Bv[0][j][0] = C16_PO(Wb[0][j], 0, CF1, CF0); Bv[0][j][1] = C16_PO(Wb[0][j], 1, CF1, CF0); \
                  Bv[0][j][2] = C16_PO(Wb[0][j], 2, CF1, CF0); Bv[0][j][3] = C16_PO(Wb[0][j], 3, CF1, CF0)
        C16_PA(0); C16_PA(1); C16_PA(2); C16_PA(3); C16_PA(4); C16_PA(5); C16_PA(6);
        C16_PB(0); C16_PB(1); C16_PB(2); C16_PB(3); C16_PB(4); C16_PB(5); C16_PB(6); C16_PB(7);
#undef C16_PB
#undef C16_PA
#undef C16_PO
    }
    fn += GST;                                     // (RF[0] stays group 0's: its row operand 7 is made behind row 0)

    for (int c = c_beg; c < c_end; c++) {
        const int cur = c & 1;
        const int q_cnt = (c * (CHS / 16) + CHS / 16 <= n_q) ? GCH : (n_q - c * (CHS / 16)) / 2;   // 32-SNP groups: a multiple of 8
        const bool more = (c + 1 < c_end);
        if (more) C16_TABLE_ASYNC(c + 1, cur ^ 1);  // every wave is past the barrier that freed this buffer
        for (int q = 0; q < q_cnt; q += 2 * D) {
            const int g = c * GCH + q;
            C16_GROUP(0, 1, 0, 1, 0, g);
            C16_GROUP(1, 0, 1, 2, 1, g + 1);
            C16_GROUP(0, 1, 2, 3, 0, g + 2);
            C16_GROUP(1, 0, 3, 0, 1, g + 3);
            C16_GROUP(0, 1, 0, 1, 0, g + 4);
            C16_GROUP(1, 0, 1, 2, 1, g + 5);
            C16_GROUP(0, 1, 2, 3, 0, g + 6);
            // the chunk's last group prepares the NEXT chunk's first group (or, at the very end, harmlessly re-reads this chunk)
            if (q + 2 * D >= q_cnt) {
                if (more) {
                    // vmcnt is in-order: the factor copy went out at the start of this chunk, behind it at least seven groups of 16 word
                    // loads -- all but the newest 62 requests covers it
                    __builtin_amdgcn_s_waitcnt(0xCF7E); // vmcnt(62)
                    __syncthreads();
                    fn = x1_lds_off(&sfac[cur ^ 1][0]) + 16 * kq;
                } else {
                    fn = x1_lds_off(&sfac[cur][0]) + 16 * kq;
                }
            }
            C16_GROUP(1, 0, 3, 0, 1, g + 7);
        }
    }
    {
        double *pflush = pacc;
        asm volatile("" : "+v"(pflush));
        const int64_t rows_left = (n_rows_real > 0 ? n_rows_real : ((int64_t)1 << 40)) - (row_w + 4 * kq);
        const bool first_run = (run == 0), last_run = (run + 1 == runs_here);
        const float fs32 = (float)fscale;          // 1 - q / 4096: exact in fp32
#pragma unroll
        for (int i = 0; i < TS; i++) {
            const int nc = carry_on ? ((CARRY_SUB - i * TS) < 0 ? 0 : (CARRY_SUB - i * TS) > TS ? TS : (CARRY_SUB - i * TS)) : 0;   // carried: j < nc
#pragma unroll
            for (int j = 0; j < TS; j++)
                if (j < nc) {                      // carried sub-tiles: fp32 sums in LDS until the block's last run
                    f32x4 *cp = &scar[wave][(i * TS + j) * 64 + lane];
                    f32x4 t = c16[i][j] * fs32;
                    if (!first_run) t += *cp;
                    if (!last_run) *cp = t;
                    c16[i][j] = t;                 // (what the last run flushes below; every other run clears it)
                }
            if (nc == TS && !last_run) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = i * 16 + r;
                double *__restrict__ pr = pflush + (int64_t)row * rs;
                if (row < rows_left) {
#pragma unroll
                    for (int j = 0; j < TS; j++)      // f_q x fp32 partial: exact in fp64 (13 + 24 bits); carried sums carry their factors already
                        if (j >= nc || last_run)
                            (void)__builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double *)(pr + 16 * j),
                                                                          (double)c16[i][j][r] * (j < nc ? 1.0 : fscale));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < TS; i++)
#pragma unroll
            for (int j = 0; j < TS; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) c16[i][j][r] = 0.f;
    }
    }   // run
#undef C16_GROUP
#undef C16_STEP8
#undef C16_STEP4
#undef C16_F2
#undef C16_C2
#undef C16_M
#undef C16_SB
#undef C16_MFMA
#undef C16_LOAD
#undef C16_TABLE_ASYNC
#undef C16_FMA
#undef C16_F0
#undef C16_F1
#undef C16_CVT
#undef C16_WORD
#undef C16_D
#undef C16_AI
#undef C16_ISA
#undef C16_R
}
#pragma clang diagnostic pop

// One launch for ALL fp32 runs of a block (round 5), work items (tile, run): an XCD's queue is walked in groups of G tiles, run by
// run inside a group -- G = 32 = the XCD's CUs: the same 32 tiles are up again one round (~250 us) later, their 512 KB fp64 regions
// still in the Infinity Cache (256 tiles x 512 KB = 128 MB per round, chip-wide), and the block has ONE tail round instead of one
// per run.  A/B on one box (ms per 65 536-SNP step, N = 100 000, profiles/r05_run_inner_ab.txt): one launch per run 470.8 / 474.3,
// G = 32 467.9 / 468.9 (-0.9 %; blocks with missing calls 986.2 -> 977.7), G = 64 472.1 / 473.2, G = 4 476.3 / 476.8, G = 1 (the six
// runs of a tile side by side on six CUs, all flushing the same lines at once) 487.7 / 489.0.  SNPGPU_RUN_INNER=0: one launch per run.
static int run_inner_launch()
{
    static const int g = getenv("SNPGPU_RUN_INNER") ? std::max(0, std::min(atoi(getenv("SNPGPU_RUN_INNER")), 1 << 20)) : 32;
    return g;
}
// workgroups of a fused launch: every XCD queue (n_blocks / 8 items) padded to whole groups, times the runs
static unsigned run_inner_grid(int n_blocks, int n_runs, int group)
{
    const int per_xcd = n_blocks / 8, groups = (per_xcd + group - 1) / group;
    return (unsigned)groups * (unsigned)group * (unsigned)n_runs * 8u;
}

// run_chunks: table chunks per fp32 run (0: the whole block is one run); n_target > 1: run q's sums are multiplied by
// uv_run_factor(q) at its flush (the run's SNPs were factorised for the weight target t / f_q, uv_factor_kernel)
int launch_syrk_uv(hipStream_t st, const int4 *work_x1, int n_blocks_x1, const uint32_t *w8, int64_t ncols_pad,
                   const uint2 *lut, int n_q, double *acc, int64_t ld, int64_t tiles_c, const unsigned long long *d_missing,
                   int64_t n_rows_real, int run_chunks, int n_target, int run_if_missing, int64_t copy_lut_bytes, int64_t copy_acc_elems,
                   int uv16, const void *pace_src, int pace)
{
    if (n_q <= 0 || n_blocks_x1 <= 0) return 0;
    const int n_chunk = (n_q + (UV_CHS / 16) - 1) / (UV_CHS / 16);           // table chunks of the block; one launch per fp32 run
    const int run = run_chunks > 0 ? run_chunks : n_chunk;
    const int n_runs = (n_chunk + run - 1) / run;
    // (round 6: a non-atomic read-modify-write flush for tiles with one owner per launch was measured -- 465.8 against 456.8 ms per
    // step in the one-launch-per-run form, profiles/r06_flush_rmw_ab.txt -- and removed)
    // uv16: the same launch geometry and arguments, the 16x16x32 form of the kernel (its tables carry swapped odd quarters)
    if (uv16 >= 2) {                              // syrk_uv16c_kernel: `lut` = the slots' factor arrays; pace-maker arguments instead of table copies
        if (uv16 == 3 && n_runs > 1)              // work items = tiles, the runs walked inside, half the sub-tiles carried in LDS
            hipLaunchKernelGGL(syrk_uv16c_kernel, dim3((unsigned)n_blocks_x1), dim3(256), 0, st, w8, ncols_pad, lut, n_q, acc, ld, tiles_c, work_x1,
                               d_missing, n_rows_real, 0, n_chunk, 1.0, n_runs, run, n_target, 0, 0, run_if_missing, (const char *)pace_src, pace);
        else if (n_runs > 1 && run_inner_launch())
            hipLaunchKernelGGL(syrk_uv16c_kernel, dim3(run_inner_grid(n_blocks_x1, n_runs, run_inner_launch())), dim3(256), 0, st, w8, ncols_pad,
                               lut, n_q, acc, ld, tiles_c, work_x1, d_missing, n_rows_real, 0, n_chunk, 1.0, n_runs, run, n_target,
                               run_inner_launch(), n_blocks_x1 / 8, run_if_missing, (const char *)pace_src, pace);
        else
            for (int lo = 0, q = 0; lo < n_chunk; lo += run, q++)
                hipLaunchKernelGGL(syrk_uv16c_kernel, dim3((unsigned)n_blocks_x1), dim3(256), 0, st, w8, ncols_pad, lut, n_q, acc, ld, tiles_c,
                                   work_x1, d_missing, n_rows_real, lo, std::min(lo + run, n_chunk),
                                   n_target > 1 ? uv_run_factor(q % n_target) : 1.0, 1, 0, 1, 1, 0, run_if_missing, (const char *)pace_src, pace);
        SNPGPU_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const auto kern = uv16 ? syrk_uv16_kernel : syrk_uv_kernel;
    if (n_runs > 1 && run_inner_launch())
        hipLaunchKernelGGL(kern, dim3(run_inner_grid(n_blocks_x1, n_runs, run_inner_launch())), dim3(256), 0, st,
                           w8, ncols_pad, lut, n_q, acc, ld, tiles_c, work_x1, d_missing, n_rows_real, 0, n_chunk, 1.0, n_runs, run, n_target,
                           run_inner_launch(), n_blocks_x1 / 8, run_if_missing, copy_lut_bytes, copy_acc_elems);
    else
        for (int lo = 0, q = 0; lo < n_chunk; lo += run, q++)
            hipLaunchKernelGGL(kern, dim3((unsigned)n_blocks_x1), dim3(256), 0, st, w8, ncols_pad, lut, n_q, acc, ld,
                               tiles_c, work_x1, d_missing, n_rows_real, lo, std::min(lo + run, n_chunk),
                               n_target > 1 ? uv_run_factor(q % n_target) : 1.0, 1, 0, 1, 1, 0, run_if_missing, copy_lut_bytes, copy_acc_elems);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// a_kind < 0: three-product kernel for every block.  a_kind 0: exact-row kernel (16-byte table entries); with
// d_missing != nullptr the table was built for the block's missing flag (build_lut_kernel) and exactly one of the two
// launches does the work (blocks with missing calls: three products), with d_missing == nullptr the exact-row kernel
// takes every block (the row value of a missing call is the fp16 residual avg - c_s).  a_kind 1 / 2: two-product kernel
// with a constant row table, always.  promote_snps: fp32 run length of the exact-row kernel (0 = default).
int launch_syrk_h3(hipStream_t st, const int4 *work, int n_blocks, const uint32_t *w8, int64_t ncols_pad,
                   const uint2 *lut, int n_q, double *acc, int64_t ld, int64_t tiles_c,
                   const unsigned long long *d_skip_if_zero, int a_kind, const unsigned long long *d_missing, int64_t n_rows_real,
                   int promote_snps,
                   const int4 *work_x1, int n_blocks_x1, const unsigned long long *d_short_runs)
{
    if (n_q <= 0 || n_blocks <= 0) return 0;
    const int p3 = H3_PROMOTE / H3_LUTCH;                                    // three products: 512-SNP chunks
    const int p2e = (promote_snps > 0 ? promote_snps : H3_PROMOTE_EXACT) / (H3_LUTCH / 2);   // exact rows: 256-SNP chunks
    const int p2c = H3_PROMOTE / H3_LUTCH;                                   // constant row table: 512-SNP chunks
    if (a_kind < 0 || (a_kind == 0 && d_missing))
        hipLaunchKernelGGL((syrk_h3_kernel<3, false>), dim3((unsigned)n_blocks), dim3(256), 0, st, w8, ncols_pad, lut, n_q, acc, ld, tiles_c, work,
                           d_skip_if_zero, a_kind == 0 ? d_missing : nullptr, n_rows_real, 0, p3);
    if (a_kind == 0 && work_x1 && !d_missing) {
        const int n_chunk = (n_q + (X1_CHS / 16) - 1) / (X1_CHS / 16);       // table chunks of the block; one launch per fp32 run
        const int run = std::max(1, (promote_snps > 0 ? promote_snps : H3_PROMOTE_EXACT) / X1_CHS);
        // (fused launch only) blocks flagged by build_lut_kernel run as half-length fp32 runs: the grid is laid out for those
        const int short_div = (d_short_runs && run >= 2 && (run % 2) == 0) ? 2 : 1;
        const int n_runs = (n_chunk + run / short_div - 1) / (run / short_div);
        // (round 6: a 16x16x32 form of this kernel -- syrk_uv16_kernel's skeleton, column operands looked up just in time, a padded
        // table layout against the bank conflicts of two quarters per LDS pass -- was built, passed every parity test and ran configs[2]
        // with 2 % missing calls in 883 - 902 ms per step against 896 here, depending on the issue pattern: not kept,
        // profiles/r06_x116_patterns.txt)
        if (n_runs > 1 && run_inner_launch())
            hipLaunchKernelGGL(syrk_x1_kernel, dim3(run_inner_grid(n_blocks_x1, n_runs, run_inner_launch())), dim3(256), 0, st, w8, ncols_pad,
                               lut, n_q, acc, ld, tiles_c, work_x1, d_skip_if_zero, n_rows_real, 0, n_chunk, n_runs, run, run_inner_launch(),
                               n_blocks_x1 / 8, short_div > 1 ? d_short_runs : nullptr, short_div);
        else
            for (int lo = 0; lo < n_chunk; lo += run)
                for (int half = 0; half < short_div; half++)      // (short_div = 2: see the kernel's one-launch-per-run branch)
                    hipLaunchKernelGGL(syrk_x1_kernel, dim3((unsigned)n_blocks_x1), dim3(256), 0, st, w8, ncols_pad, lut, n_q, acc, ld, tiles_c,
                                       work_x1, d_skip_if_zero, n_rows_real, lo, std::min(lo + run, n_chunk), 1, run, half, 0,
                                       short_div > 1 ? d_short_runs : nullptr, short_div);
    } else if (a_kind == 0)
        hipLaunchKernelGGL((syrk_h3_kernel<2, true>), dim3((unsigned)n_blocks), dim3(256), 0, st, w8, ncols_pad, lut, n_q, acc, ld, tiles_c,
                           work, d_skip_if_zero, d_missing, n_rows_real, a_kind, p2e > 0 ? p2e : 1);
    else if (a_kind > 0)
        hipLaunchKernelGGL((syrk_h3_kernel<2, false>), dim3((unsigned)n_blocks), dim3(256), 0, st, w8, ncols_pad, lut, n_q, acc, ld, tiles_c,
                           work, d_skip_if_zero, nullptr, n_rows_real, a_kind, p2c);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// pair_sparse_miss_kernel: the GCTA both-missing counts of a 256 x 256 tile from the SETS of samples with a missing call
// (missmask256_kernel: MM[group][snp] = 256 bits).  One workgroup of 1024 threads per work item {tile row, tile column, K part,
// K parts}; the tile's 65 536 counters sit in LDS as 16-bit halves of 32 768 dwords (counter (r, c) = half c & 1 of dword
// 128 r + c / 2; at most 32 768 SNPs between two flushes, so a half cannot overflow).  A thread takes one SNP at a time: the two
// sets (2 x 32 bytes, consecutive SNPs adjacent), and for every pair (bit of the row set, bit of the column set) one LDS atomic
// add.  Work ~ f^2 N^2 B / 2 for a missing rate f against N^2 B / 2 int8 products of the dense form: 1 / 2500 of the products at
// f = 2 %.  The flush adds the non-zero dwords to the counter panel (atomic: parts share a tile).
__global__ __launch_bounds__(1024) void pair_sparse_miss_kernel(const uint4 *__restrict__ mm, int64_t snp_stride, int n_snp,
                                                                uint32_t *__restrict__ acc, int64_t ncols_pad,
                                                                const int4 *__restrict__ work,
                                                                const unsigned long long *__restrict__ d_run)
{
    if (*d_run == 0ull) return;
    __shared__ uint32_t cnt[256 * 128];
    const int4 item = work[blockIdx.x];
    if (item.w == 0) return;
    const int per = (((n_snp + item.w - 1) / item.w) + 1023) / 1024 * 1024;
    const int s_beg = item.z * per, s_end = (s_beg + per < n_snp) ? (s_beg + per) : n_snp;
    if (s_beg >= s_end) return;
    const int tid = threadIdx.x;
    const uint4 *__restrict__ mi = mm + (int64_t)item.x * snp_stride * 2;
    const uint4 *__restrict__ mj = mm + (int64_t)item.y * snp_stride * 2;
    for (int sb = s_beg; sb < s_end; sb += 32768) {
        const int se = (sb + 32768 < s_end) ? (sb + 32768) : s_end;
        for (int e = tid; e < 256 * 128; e += 1024) cnt[e] = 0u;
        __syncthreads();
        for (int s = sb + tid; s < se; s += 1024) {
            const uint4 a0 = mi[2 * (int64_t)s], a1 = mi[2 * (int64_t)s + 1];
            if ((a0.x | a0.y | a0.z | a0.w | a1.x | a1.y | a1.z | a1.w) == 0u) continue;
            const uint4 b0 = mj[2 * (int64_t)s], b1 = mj[2 * (int64_t)s + 1];
            const uint32_t ra[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const uint32_t cb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            if ((b0.x | b0.y | b0.z | b0.w | b1.x | b1.y | b1.z | b1.w) == 0u) continue;
            for (int a = 0; a < 8; a++) {
                uint32_t x = ra[a];
                while (x) {
                    const int r = 32 * a + __builtin_ctz(x);
                    x &= x - 1u;
                    uint32_t *row = cnt + r * 128;
                    for (int b = 0; b < 8; b++) {
                        uint32_t y = cb[b];
                        while (y) {
                            const int c = __builtin_ctz(y);
                            y &= y - 1u;
                            atomicAdd(row + 16 * b + (c >> 1), 1u << (16 * (c & 1)));
                        }
                    }
                }
            }
        }
        __syncthreads();
        uint32_t *__restrict__ dst = acc + (int64_t)item.x * 256 * ncols_pad + (int64_t)item.y * 256;
        for (int e = tid; e < 256 * 128; e += 1024) {
            const uint32_t v = cnt[e];
            if (v) {
                uint32_t *p = dst + (int64_t)(e >> 7) * ncols_pad + 2 * (e & 127);
                if (v & 0xFFFFu) atomicAdd(p, v & 0xFFFFu);
                if (v >> 16) atomicAdd(p + 1, v >> 16);
            }
        }
        __syncthreads();
    }
}

int launch_pair_sparse_miss(hipStream_t st, const uint4 *mm, int64_t snp_stride, int n_snp, uint32_t *acc, int64_t ncols_pad,
                            const int4 *work, int n_blocks, const unsigned long long *d_run)
{
    if (n_snp <= 0 || n_blocks <= 0) return 0;
    hipLaunchKernelGGL(pair_sparse_miss_kernel, dim3((unsigned)n_blocks), dim3(1024), 0, st, mm, snp_stride, n_snp, acc, ncols_pad, work, d_run);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// The same counters as exact int8 contractions on the matrix cores.
// Per genotype code (0,1,2 = allele count, 3 = missing) define int8 values
//     v = called   h = het   y = hom   s = v - 2h   x = [g==0] - [g==2]      (all 0 for missing)
// then for a pair of samples, summed over SNPs,
//     v.v' = both called            s.s' = both called - 2 * (exactly one het)
//     h.v' = row het & col called   h.h' = both het      y.y' - x.x' = 2 * (opposite homozygotes)
//     y.y' + x.x' = 2 * (equal homozygotes)             h.y' + y.h' = exactly one het
// i.e. every counter of PairOps<> is an exact integer combination of a few int8 dot products, which
// v_mfma_i32_32x32x32_i8 evaluates 32 SNPs x 1024 pairs at a time with int32 accumulation (bit-exact).
// Measured on MI355X (tools/ubench/i8_ubench.hip): SIMD time = 36 cycles per MFMA + ~4 cycles per
// VALU op (they do not overlap), so the operand decode is kept to shift/and (four clean codes per
// dword: (w >> 2u) & 0x03030303) plus ONE v_perm_b32 per operand dword (the code bytes select from a
// 4-byte value table).  Lane l holds sample (l & 31) and 16 of the 32 SNPs of a k-step in one dword
// of the sample-major 2-bit words W2[d][sample] (d = 16-SNP group, half h = l >> 5 reads d = 2q + h);
// any SNP order inside a k-step is legal because both operands use the same one.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

#define I8T_V 0x00010101u     /* byte c of the table = value for code c */
#define I8T_H 0x00000100u
#define I8T_NH 0x0000FF00u
#define I8T_S 0x0001FF01u
#define I8T_Y 0x00010001u
#define I8T_X 0x00FF0001u
#define I8T_NX 0x000100FFu
#define I8T_M 0x01000000u     /* code 3 only */
#define I8T_E0 0x00000001u    /* g == 0 */
#define I8T_E2 0x00010000u    /* g == 2 */
#define I8T_G 0x00020100u     /* g itself (0 for missing) */
#define I8T_G2 0x00000102u    /* 2 - g (0 for missing) */
#define I8T_CODE 0xFFFFFFFFu  /* the code byte itself as operand (0, 1, 2; 3 for missing / padding): no table lookup */

template <int MODE> struct I8Scheme;
template <> struct I8Scheme<PM_IBS> {            // 4 MFMA slots, 3 accumulators, 64 x 64 per wave
    static constexpr int NS = 4, NA = 3, TM = 2, TN = 2, C = 3, WPS = 2;
    // ibs0 as e0.e2' + e2.e0' (binary operands) rather than (y.y' - x.x') / 2 (x = +-1): same four products and
    // value types, fewer toggling multiplier bits -- the kernel runs at the socket power cap (HISTORY.md 4.5)
    static __device__ __forceinline__ constexpr uint32_t ta(int s) { return s == 0 ? I8T_V : s == 1 ? I8T_S : s == 2 ? I8T_E0 : I8T_E2; }
    static __device__ __forceinline__ constexpr uint32_t tb(int s) { return s == 0 ? I8T_V : s == 1 ? I8T_S : s == 2 ? I8T_E2 : I8T_E0; }
    static __device__ __forceinline__ constexpr int acc(int s) { return s == 0 ? 0 : s == 1 ? 1 : 2; }
    static __device__ __forceinline__ void emit(const int *a, int, uint32_t *cnt)   // {nvalid, ibs1, 2 ibs0}
    {
        cnt[0] = (uint32_t)a[0]; cnt[1] = (uint32_t)(a[0] - a[1]) >> 1; cnt[2] = 2u * (uint32_t)a[2];
    }
};
// IBS / KING-robust for blocks WITHOUT missing calls (imputed data): TWO products, h.h' and g.g' (h = het, g = the
// genotype itself).  With the per-sample counts H = #het and T = #(g == 2) of the block:
//     ibs1 = H_i + H_j - 2 h.h'                (exactly one het)
//     sum (g - g')^2 = ibs1 + 4 ibs0 = (H_i + 4 T_i) + (H_j + 4 T_j) - 2 g.g'      =>      2 ibs0 = 2 (T_i + T_j) + h.h' - g.g'
//     KING: N1_Aa = H_i, N2_Aa = H_j
// (with the per-sample margins known, ibs0 and ibs1 span two dimensions modulo separable terms: two products is the
// minimum; the three-product binary form h.h', e0.e2', e2.e0' was 1.5x the MFMA work).  g.g' needs no operand table at all:
// the extracted code bytes ARE g (padding SNPs hold code 3 for every sample: 9 per padding SNP and pair, a constant the
// flush puts back) -- 24 fewer v_perm_b32 per k-step, 4.3 instead of 5.8 decode instructions per MFMA.  Operands {0, 1} and
// {0, 1, 2, (3)}: the first two-product form used x = [g==0] - [g==2], whose -1 bytes put the kernel back at the power cap
// (2.2 GHz, 1364 W).  The kernel adds {n, -2 h.h', h.h' - g.g'}; the rank-one terms H_i + H_j, 2 (T_i + T_j) (and N1, N2)
// are added once, when a result is asked for (counts from the transposition kernel per block, het_settle_kernel at the
// end).  Plane 2 of the IBS / KING-robust counters therefore carries 2 ibs0 for EVERY block and backend (the halves do not
// separate per block); the finalisers shift.  Selected per block on the device (missing-call flag).
// Per-wave tile 128 x 64 with ONE wave per SIMD: the 2 x 8 x 16 = 256 accumulators live in AGPRs, 203 VGPRs hold the
// pipeline.  Against 64 x 64 at two waves per SIMD the decode drops from 6.3 to 4.75 VALU per MFMA (114 per 24 MFMAs), and
// with four word sets in flight (I8PipeSpread::D) the lone wave never waits for its loads: 5.19 -> 4.70 ms per 65 536-SNP
// block at N = 10 000 (A/B on one box; the same tile with two word sets: 5.05 ms).
#ifndef I8_NOMISS_TM
#define I8_NOMISS_TM 4
#define I8_NOMISS_WPS 1
#endif
// GCTA both-missing product: 128 x 128 per wave, one wave per SIMD (256 AGPR accumulators), operands straight from the
// masked words (I8ExtractMask): 7 VALU per operand dword, 3.5 per MFMA.  54 -> 37 ms per 16 384-SNP block at N = 100 000
// with 2 % missing calls (A/B on one box; the 128 x 64 / two-waves form with the same extraction: 70 ms -- its single code set
// aliases the operand registers the queued MFMAs still read).
#ifndef I8_GCTA_TN
#define I8_GCTA_TN 4
#define I8_GCTA_WPS 1
#endif
#ifndef I8_KING_TN
#define I8_KING_TN 2
#define I8_KING_WPS 2
#endif
template <> struct I8Scheme<PM_IBS_NOMISS> {
    static constexpr int NS = 2, NA = 2, TM = I8_NOMISS_TM, TN = 2, C = 3, WPS = I8_NOMISS_WPS;
    // slot 0: g.g' straight from the code bytes (no v_perm_b32: a block without missing calls holds the codes 0, 1, 2, and
    // 3 only as SNP / sample padding); slot 1: h.h'.  The code product comes FIRST: its MFMAs read the code registers of this
    // k-step, which the extraction of the k-step after next overwrites a whole product later
    static __device__ __forceinline__ constexpr uint32_t ta(int s) { return s == 0 ? I8T_CODE : I8T_H; }
    static __device__ __forceinline__ constexpr uint32_t tb(int s) { return s == 0 ? I8T_CODE : I8T_H; }
    static __device__ __forceinline__ constexpr int acc(int s) { return s; }
    // a[0] = g.g' + 9 (padding SNPs of this K part), a[1] = h.h'.  {nvalid, ibs1 - H_i - H_j}; plane 2 gets h.h' + 9 pad
    // (atomic add) and - a[0] (global_atomic_sub) from the flush -- any arithmetic on the first accumulator there (a sum of
    // both, a negation) tipped the register allocator into spilling 85 registers, some inside the K loop
    static __device__ __forceinline__ void emit(const int *a, int nv, uint32_t *cnt)
    {
        cnt[0] = (uint32_t)nv; cnt[1] = 0u - 2u * (uint32_t)a[1]; cnt[2] = 0u;
    }
};
// GCTA denominators: both-missing counts over the masked words (code 3 = missing call at a polymorphic SNP
// of a real sample, launch_transpose2_missmask) -- one product, one accumulator.
template <> struct I8Scheme<PM_GCTA_MISS> {
    static constexpr int NS = 1, NA = 1, TM = 4, TN = I8_GCTA_TN, C = 1, WPS = I8_GCTA_WPS;
    static __device__ __forceinline__ constexpr uint32_t ta(int) { return I8T_M; }
    static __device__ __forceinline__ constexpr uint32_t tb(int) { return I8T_M; }
    static __device__ __forceinline__ constexpr int acc(int) { return 0; }
    static __device__ __forceinline__ void emit(const int *a, int, uint32_t *cnt) { cnt[0] = (uint32_t)a[0]; }
};
// KING-robust in the basis {y, h, x} (v = y + h): five products / five accumulators / three value types
//   a0 = y.y'  a1 = x.x'  a2 = y.h'  a3 = h.y'  a4 = h.h'
//   nLoci = a0 + a2 + a3 + a4   N1_Aa (row het, column called) = a3 + a4   N2_Aa = a2 + a4
//   ibs1 = a2 + a3   ibs0 = (a0 - a1) / 2
// (the direct form {v.v', h.v', v.h', h.h', y.y' - x.x'} needs six products and four value types)
template <> struct I8Scheme<PM_KING_ROBUST> {
    static constexpr int NS = 5, NA = 5, TM = 1, TN = I8_KING_TN, C = 5, WPS = I8_KING_WPS;
    static __device__ __forceinline__ constexpr uint32_t ta(int s) { return s == 0 ? I8T_Y : s == 1 ? I8T_X : s == 2 ? I8T_Y : I8T_H; }
    static __device__ __forceinline__ constexpr uint32_t tb(int s) { return s == 0 ? I8T_Y : s == 1 ? I8T_X : s == 2 ? I8T_H : s == 3 ? I8T_Y : I8T_H; }
    static __device__ __forceinline__ constexpr int acc(int s) { return s; }
    static __device__ __forceinline__ void emit(const int *a, int, uint32_t *cnt)   // {nLoci, ibs1, 2 ibs0, N1_Aa, N2_Aa}
    {
        cnt[0] = (uint32_t)(a[0] + a[2] + a[3] + a[4]); cnt[1] = (uint32_t)(a[2] + a[3]);
        cnt[2] = (uint32_t)(a[0] - a[1]) /* 2 ibs0 */; cnt[3] = (uint32_t)(a[3] + a[4]); cnt[4] = (uint32_t)(a[2] + a[4]);
    }
};
template <> struct I8Scheme<PM_KING_HOMO> {      // 4 slots, 2 accumulators: 128 x 64 per wave at one wave per SIMD, as the binary kernel
    static constexpr int NS = 4, NA = 2, TM = 4, TN = 2, C = 2, WPS = 1;
    // ibs0 = e0.e2' + e2.e0' (binary operands, as in I8Scheme<PM_IBS>)
    static __device__ __forceinline__ constexpr uint32_t ta(int s) { return s == 0 ? I8T_H : s == 1 ? I8T_Y : s == 2 ? I8T_E0 : I8T_E2; }
    static __device__ __forceinline__ constexpr uint32_t tb(int s) { return s == 0 ? I8T_Y : s == 1 ? I8T_H : s == 2 ? I8T_E2 : I8T_E0; }
    static __device__ __forceinline__ constexpr int acc(int s) { return s < 2 ? 0 : 1; }
    static __device__ __forceinline__ void emit(const int *a, int, uint32_t *cnt)   // {ibs1, 2 ibs0}
    {
        cnt[0] = (uint32_t)a[0]; cnt[1] = 2u * (uint32_t)a[1];
    }
};
// KING-homo, blocks without missing calls: the two products of I8Scheme<PM_IBS_NOMISS> into the planes {ibs1, 2 ibs0}
template <> struct I8Scheme<PM_HOMO_NOMISS> {
    static constexpr int NS = 2, NA = 2, TM = I8_NOMISS_TM, TN = 2, C = 2, WPS = I8_NOMISS_WPS;
    static __device__ __forceinline__ constexpr uint32_t ta(int s) { return s == 0 ? I8T_CODE : I8T_H; }
    static __device__ __forceinline__ constexpr uint32_t tb(int s) { return s == 0 ? I8T_CODE : I8T_H; }
    static __device__ __forceinline__ constexpr int acc(int s) { return s; }
    static __device__ __forceinline__ void emit(const int *a, int, uint32_t *cnt)   // {ibs1 - H_i - H_j, 0} (+ h.h' + 9 pad - a[0] in the flush)
    {
        cnt[0] = 0u - 2u * (uint32_t)a[1]; cnt[1] = 0u;
    }
};
// individual beta: the three counters lie in the span of three symmetric rank-one products,
//   a0 = y.y' (both homozygous)   a1 = x.x' (equal - opposite homozygotes)   a2 = v.v' (both called)
//   num = a2   at least one het (both called) = a2 - a0   equal homozygotes = (a0 + a1) / 2
// (round 1 built the middle one from y.h' + h.y' + h.h': five products)
template <> struct I8Scheme<PM_BETA> {
    static constexpr int NS = 3, NA = 3, TM = 2, TN = 2, C = 3, WPS = 1;
    static __device__ __forceinline__ constexpr uint32_t ta(int s) { return s == 0 ? I8T_Y : s == 1 ? I8T_X : I8T_V; }
    static __device__ __forceinline__ constexpr uint32_t tb(int s) { return s == 0 ? I8T_Y : s == 1 ? I8T_X : I8T_V; }
    static __device__ __forceinline__ constexpr int acc(int s) { return s; }
    static __device__ __forceinline__ void emit(const int *a, int, uint32_t *cnt)   // {num, >= one het, equal homozygotes}
    {
        cnt[0] = (uint32_t)a[2]; cnt[1] = (uint32_t)(a[2] - a[0]); cnt[2] = (uint32_t)(a[0] + a[1]) >> 1;
    }
};

__device__ __forceinline__ i32x4 i8_decode(uint32_t tbl, const uint32_t *e)
{
    i32x4 r;
#pragma unroll
    for (int u = 0; u < 4; u++) r[u] = (int)__builtin_amdgcn_perm(0u, tbl, e[u]);
    return r;
}

// The masked words of the GCTA denominators hold only the codes 0 and 3 ("missing call at a polymorphic SNP"): bit 0 of a
// code IS the int8 operand, so (w >> 2u) & 0x01010101 delivers four operand bytes and the v_perm_b32 table lookup falls
// away -- 7 instead of 11 VALU instructions per 16 SNPs (the kernel is VALU-issue-bound: 9.85 VALU per MFMA measured).
template <int MODE> struct I8ExtractMask { static constexpr uint32_t value = 0x03030303u; };
template <> struct I8ExtractMask<PM_GCTA_MISS> { static constexpr uint32_t value = 0x01010101u; };
template <int MODE> __device__ __forceinline__ i32x4 i8_decode_mode(uint32_t tbl, const uint32_t *e)
{
    if (MODE == PM_GCTA_MISS || tbl == I8T_CODE) {
        i32x4 r;
#pragma unroll
        for (int u = 0; u < 4; u++) r[u] = (int)e[u];
        return r;
    }
    return i8_decode(tbl, e);
}

// Software pipeline of one wave.  int8 MFMAs and VALU ops overlap on gfx950 (about 6 VALU ops hide
// behind one 32x32x32 MFMA, tools/ubench/coissue_ubench.hip), but only if the decode of the NEXT slot
// writes other registers than the queued MFMAs read: two operand register sets; slot s+1 is decoded
// while the MFMAs of slot s run, and the last slot of a k-step extracts the next k-step's codes.
// W2 has spare rows, so the words two k-steps ahead are loaded unconditionally.
template <int MODE> struct I8Pipe {
    typedef I8Scheme<MODE> S;
    static constexpr int TM = S::TM, TN = S::TN, NA = S::NA;
    // the operand register sets alternate per slot: an odd number of products is walked two k-steps at a time
    static constexpr int STEPS = (S::NS % 2 == 0) ? 1 : 2;
    const uint32_t *pa, *pb;
    int64_t kstride;
    uint32_t cw[TM + TN], e[TM + TN][4];
    i32x4 A[2][TM], B[2][TN];

    __device__ __forceinline__ void load_words()
    {
#pragma unroll
        for (int i = 0; i < TM; i++) cw[i] = pa[64 * i];
#pragma unroll
        for (int j = 0; j < TN; j++) cw[TM + j] = pb[64 * j];
        pa += kstride; pb += kstride;
    }
    __device__ __forceinline__ void extract()
    {
#pragma unroll
        for (int g = 0; g < TM + TN; g++)
#pragma unroll
            for (int u = 0; u < 4; u++) e[g][u] = (cw[g] >> (2 * u)) & I8ExtractMask<MODE>::value;
    }
    template <int SLOT, int SET> __device__ __forceinline__ void decode()
    {
#pragma unroll
        for (int i = 0; i < TM; i++) A[SET][i] = i8_decode_mode<MODE>(S::ta(SLOT), e[i]);
#pragma unroll
        for (int j = 0; j < TN; j++) B[SET][j] = i8_decode_mode<MODE>(S::tb(SLOT), e[TM + j]);
    }
    template <int P> __device__ __forceinline__ void phase(i32x16 (&c)[NA][TM][TN])
    {
        constexpr int s = P % S::NS;                     // product of this phase
        constexpr int cur = P & 1, nxt = cur ^ 1;
        constexpr bool last = (s == S::NS - 1);          // last product of a k-step
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
                c[S::acc(s)][i][j] =
                    __builtin_amdgcn_mfma_i32_32x32x32_i8(A[cur][i], B[cur][j], c[S::acc(s)][i][j], 0, 0, 0);
        if (last) {
            extract();
            load_words();
            decode<0, nxt>();
        } else {
            decode<last ? 0 : s + 1, nxt>();
        }
        // one MFMA, then a share of this phase's VALU work
        constexpr int dv = (MODE == PM_GCTA_MISS) ? 0 : 4;         // VALU ops per operand dword of a decode
        constexpr int nv = last ? (7 + dv) * (TM + TN) : dv * (TM + TN);
        constexpr int per = (nv + TM * TN - 1) / (TM * TN);
#pragma unroll
        for (int m = 0; m < TM * TN; m++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int... Is> __device__ __forceinline__ void kstep(i32x16 (&c)[NA][TM][TN], std::integer_sequence<int, Is...>)
    {
        (phase<Is>(c), ...);
    }
    __device__ __forceinline__ void prologue()
    {
        load_words();
        extract();
        load_words();
        decode<0, 0>();
    }
};

// The same pipeline with the code extraction of the NEXT k-step spread over the phases (row group g in phase g)
// instead of sitting in the last one: needs a second set of words and codes (4 (TM + TN) + (TM + TN) VGPRs) and
// at least TM + TN products.  Used where the register budget allows (KING-robust: 5 products, 3 row groups).
template <int MODE> struct I8PipeSpread {
    typedef I8Scheme<MODE> S;
    static constexpr int TM = S::TM, TN = S::TN, NA = S::NA, R = TM + TN;
    // D word sets: the words of k-step j + D are requested at the start of k-step j and first used (extracted) during
    // k-step j + D - 1.  Four sets: a wave that has its SIMD to itself (WPS == 1) must never wait for its loads (binary kernel
    // 5.05 -> 4.70 ms against two sets); with a second wave on the SIMD they still bought 1.6 % (KING-robust 9.11 -> 8.96 ms).
#ifndef I8_D2
#define I8_D2 4
#endif
    static constexpr int D = (S::WPS == 1) ? 4 : I8_D2;
    static constexpr int STEPS = D;                 // k-steps per loop round (even: the code sets alternate per k-step)
    // row group g is extracted in phase (g * NS) / R of the previous k-step
    const uint32_t *pa, *pb;
    int64_t kstride;
    uint32_t cw[D][R], e[2][R][4];
    i32x4 A[2][TM], B[2][TN];

    template <int K> __device__ __forceinline__ void load_words()
    {
#pragma unroll
        for (int i = 0; i < TM; i++) cw[K][i] = pa[64 * i];
#pragma unroll
        for (int j = 0; j < TN; j++) cw[K][TM + j] = pb[64 * j];
        pa += kstride; pb += kstride;
    }
    template <int W, int K, int G> __device__ __forceinline__ void extract_group()      // word set W -> code set K
    {
#pragma unroll
        for (int u = 0; u < 4; u++) e[K][G][u] = (cw[W][G] >> (2 * u)) & I8ExtractMask<MODE>::value;
    }
    template <int K, int SLOT, int SET> __device__ __forceinline__ void decode()
    {
#pragma unroll
        for (int i = 0; i < TM; i++) A[SET][i] = i8_decode_mode<MODE>(S::ta(SLOT), e[K][i]);
#pragma unroll
        for (int j = 0; j < TN; j++) B[SET][j] = i8_decode_mode<MODE>(S::tb(SLOT), e[K][TM + j]);
    }
    static constexpr int groups_in_phase(int s)
    {
        int n = 0;
        for (int g = 0; g < R; g++) n += ((g * S::NS) / R == s);
        return n;
    }
    template <int W, int K, int G, int PH> __device__ __forceinline__ void extract_if()
    {
        if ((G * S::NS) / R == PH) extract_group<W, K, G>();
    }
    template <int W, int K, int PH, int... Gs> __device__ __forceinline__ void extract_for_phase(std::integer_sequence<int, Gs...>)
    {
        (extract_if<W, K, Gs, PH>(), ...);
    }
    template <int P> __device__ __forceinline__ void phase(i32x16 (&c)[NA][TM][TN])
    {
        constexpr int s = P % S::NS, kj = P / S::NS;            // product, k-step of this loop round
        constexpr int kp = kj & 1, ws = kj % D;                 // code set / word set of this k-step
        constexpr int cur = P & 1, nxt = cur ^ 1;
        constexpr bool last = (s == S::NS - 1);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
                c[S::acc(s)][i][j] =
                    __builtin_amdgcn_mfma_i32_32x32x32_i8(A[cur][i], B[cur][j], c[S::acc(s)][i][j], 0, 0, 0);
        if (s == 0) load_words<ws>();                           // words D k-steps ahead (this k-step's are consumed)
        extract_for_phase<(ws + 1) % D, kp ^ 1, s>(std::make_integer_sequence<int, R>{});   // codes of the next k-step
        if (last) decode<kp ^ 1, 0, nxt>();
        else decode<kp, last ? 0 : s + 1, nxt>();
        constexpr int n_ext = groups_in_phase(s);
        constexpr int nv = 4 * R + 7 * n_ext;
        constexpr int per = (nv + TM * TN - 1) / (TM * TN);
#pragma unroll
        for (int m = 0; m < TM * TN; m++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int... Is> __device__ __forceinline__ void kstep(i32x16 (&c)[NA][TM][TN], std::integer_sequence<int, Is...>)
    {
        (phase<Is>(c), ...);
    }
    template <int... Gs> __device__ __forceinline__ void extract_all0(std::integer_sequence<int, Gs...>)
    {
        (extract_group<0, 0, Gs>(), ...);
    }
    template <int... Ks> __device__ __forceinline__ void load_rest(std::integer_sequence<int, Ks...>)
    {
        (load_words<Ks + 1>(), ...);
    }
    __device__ __forceinline__ void prologue()
    {
        load_words<0>();
        extract_all0(std::make_integer_sequence<int, R>{});
        load_rest(std::make_integer_sequence<int, D - 1>{});
        decode<0, 0, 0>();
    }
};
template <int MODE, bool SPREAD> struct I8PipeSel { typedef I8Pipe<MODE> type; };
template <int MODE> struct I8PipeSel<MODE, true> { typedef I8PipeSpread<MODE> type; };

// Workgroup = 4 waves as 2 x 2, tile (64 TM) x (64 TN).  Workgroup b executes work item b of a host-built
// list (api.hip: build_i8_worklist): {tile row, tile col, K part, K parts}; the list is interleaved so
// that the items of one XCD (b % 8) walk neighbouring tiles, and its tail holds K-split items so that the
// last, partially filled round of workgroups is short.  The flush is atomic, parts may share a tile.
template <int MODE>
__global__ __launch_bounds__(256, I8Scheme<MODE>::WPS) void pair_mfma_i8_kernel(
    const uint32_t *__restrict__ w2, int64_t ncols_pad, int n_q, int n_snp, uint32_t *__restrict__ acc, int64_t acc_plane,
    const int4 *__restrict__ work, const unsigned long long *__restrict__ d_missing, int run_if_missing)
{
    typedef I8Scheme<MODE> S;
    constexpr int TM = S::TM, TN = S::TN, NA = S::NA;
    if (d_missing && ((*d_missing != 0ull) != (run_if_missing != 0))) return;   // the other variant handles this block
    const int4 item = work[blockIdx.x];
    if (item.w == 0) return;
    struct { int tr, tc; } t = {item.x, item.y};
    typedef typename I8PipeSel<MODE, (MODE == PM_KING_ROBUST || MODE == PM_KING_HOMO || MODE == PM_IBS_NOMISS || MODE == PM_HOMO_NOMISS || MODE == PM_BETA ||
                                      (MODE == PM_GCTA_MISS && I8Scheme<MODE>::WPS == 1))>::type Pipe;
    constexpr int KR = Pipe::STEPS > 2 ? Pipe::STEPS : 2;            // k-steps per loop round (n_q is a multiple of 4: blocks are padded to 128 SNPs)
    const int per = (((n_q + item.w - 1) / item.w) + KR - 1) / KR * KR;
    const int q_beg = item.z * per;
    const int q_end = (q_beg + per < n_q) ? (q_beg + per) : n_q;
    if (q_beg >= q_end) return;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, kh = lane >> 5;
    const int row_base = t.tr * (64 * TM) + wr * (32 * TM);
    const int64_t col_base = (int64_t)t.tc * (64 * TN) + wc * (32 * TN);
    // W2 = uint2[row pair = k-step][sample]: the halves of a k-step lie side by side (lane half kh takes element kh)
    const uint32_t *__restrict__ pa = w2 + 2 * ((int64_t)q_beg * ncols_pad + row_base + li) + kh;
    const uint32_t *__restrict__ pb = w2 + 2 * ((int64_t)q_beg * ncols_pad + col_base + li) + kh;
    const int64_t kstride = 2 * ncols_pad;

    i32x16 c[NA][TM][TN];
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) c[a][i][j][r] = 0;

    // Software pipeline (I8Pipe): slot s+1 is decoded while the MFMAs of slot s run.
    Pipe pipe;
    pipe.pa = pa; pipe.pb = pb; pipe.kstride = kstride;
    pipe.prologue();
    for (int q = q_beg; q < q_end; q += Pipe::STEPS)                // q_end - q_beg is a multiple of KR
        pipe.kstep(c, std::make_integer_sequence<int, S::NS * Pipe::STEPS>{});
    // real SNPs of this K part (both-called count of a block without missing calls)
    const int nv_lo = 32 * q_beg, nv_hi = (32 * q_end < n_snp) ? 32 * q_end : n_snp;
    const int nv = (nv_hi > nv_lo) ? (nv_hi - nv_lo) : 0;
    const int pad9 = 9 * (32 * (q_end - q_beg) - nv);      // code product of the padding SNPs of this K part (code 3 x code 3)
    (void)pad9;
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // One owner per element and K slice: fire-and-forget atomic adds (measured against streaming
    // load/add/store updates of the HBM-resident counters, tools/ubench/i8_ubench.hip: atomics cost 4.5 %
    // of an IBS launch at N = 10 000, B = 16 384, the load/store form 6.5 %).
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
            uint32_t *p0 = acc + (int64_t)(row_base + 32 * i + 4 * kh) * ncols_pad + col_base + 32 * j + li;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int a[NA];
                uint32_t cnt[S::C];
#pragma unroll
                for (int k = 0; k < NA; k++) a[k] = c[k][i][j][r];
                S::emit(a, nv, cnt);
                uint32_t *p = p0 + (int64_t)((r & 3) + 8 * (r >> 2)) * ncols_pad;
#pragma unroll
                for (int k = 0; k < S::C; k++) {
                    if ((MODE == PM_IBS_NOMISS && k == 2) || (MODE == PM_HOMO_NOMISS && k == 1)) continue;   // below
                    atomicAdd(p + (int64_t)k * acc_plane, cnt[k]);
                }
                if (MODE == PM_IBS_NOMISS || MODE == PM_HOMO_NOMISS) {    // 2 ibs0 += h.h' - g.g' (+ the rank-one terms at settle time)
                    uint32_t *p0 = p + (MODE == PM_IBS_NOMISS ? 2 : 1) * acc_plane;
                    atomicAdd(p0, (uint32_t)(a[1] + pad9));
                    asm volatile("global_atomic_sub %0, %1, off" : : "v"(p0), "v"(a[0]) : "memory");
                }
            }
        }
}

template <int MODE>
static int launch_i8(hipStream_t st, const int4 *work, int n_blocks, const uint32_t *w2, int64_t ncols_pad, int n_q,
                     int n_snp, uint32_t *acc, int64_t acc_plane, const unsigned long long *d_missing, int run_if_missing)
{
    hipLaunchKernelGGL(pair_mfma_i8_kernel<MODE>, dim3((unsigned)n_blocks), dim3(256), 0, st, w2, ncols_pad, n_q, n_snp,
                       acc, acc_plane, work, d_missing, run_if_missing);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// GCTA both-missing counts on the MX-fp4 matrix instruction (round 4).  The masked words hold only the codes 0 and 3, so bit 1 of
// a code, left where it is, IS an e2m1 nibble: 0b0010 = 1.0 (0b0000 = 0).  One 16-code word gives two operand dwords of eight
// nibbles -- w & 0x22222222 (even SNPs) and (w >> 2) & 0x22222222 (odd SNPs); any SNP order inside a k-step is legal, both
// operands use the same one -- three VALU per two dwords.  v_mfma_scale_f32_32x32x64_f8f6f4 with both formats fp4 (cbsz = blgp = 4)
// and unit E8M0 scales (0x7F) takes 64 SNPs x 1024 pairs per instruction at twice the int8 rate (MI355X_MICROARCH.md: 9099 TF against
// 4404 TOP/s); the products are 0 or 1 and the fp32 sums exact (a launch holds <= 2^16 SNPs < 2^24).  Same tile, work list
// and flush as pair_mfma_i8_kernel<PM_GCTA_MISS>: 128 x 128 per wave in AGPRs, one wave per SIMD, four word sets in flight.
// Lane l: sample l & 31, SNPs 32 (l >> 5) ... + 32 of the k-step = the word rows 4 s + 2 (l >> 5) + {0, 1}.
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
struct Fp4MissPipe {
    static constexpr int TM = 4, TN = 4, R = TM + TN, D = 4;
    // the word PAIRS of a sample lie side by side (launch_transpose2_missmask, paired): one 8-byte load per 32 SNPs, 32 loads in
    // flight with four sets (64 single-word loads overran the 6-bit vmcnt counter: the compiler then waited for loads it had just
    // issued).  Uniform base (SGPRs, advanced per k-step) + one constant byte offset per lane and operand.
    const char *base;
    uint32_t offa, offb;
    int64_t kstride;
    uint2 cw[D][R];
    i32x4 A[2][TM], B[2][TN];

    template <int K> __device__ __forceinline__ void load_words()
    {
        // raw buffer loads: descriptor = the uniform row address (SGPRs, rebuilt per k-step with scalar adds), lane offset in ONE
        // VGPR.  (Flat loads from base + offset kept 64-bit lane addresses alive across the loop; the allocator spilled them and
        // every reload -- scratch counts in vmcnt -- drained the four word sets in flight.)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(offa + 256 * i), 0, 0);
            cw[K][i] = make_uint2(v[0], v[1]);
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(offb + 256 * j), 0, 0);
            cw[K][TM + j] = make_uint2(v[0], v[1]);
        }
        base += kstride;
    }
    static __device__ __forceinline__ i32x4 nibbles(const uint2 w)
    {
        i32x4 r;
        r[0] = (int)(w.x & 0x22222222u); r[1] = (int)((w.x >> 2) & 0x22222222u);
        r[2] = (int)(w.y & 0x22222222u); r[3] = (int)((w.y >> 2) & 0x22222222u);
        return r;
    }
    template <int K, int SET> __device__ __forceinline__ void decode()
    {
#pragma unroll
        for (int i = 0; i < TM; i++) A[SET][i] = nibbles(cw[K][i]);
#pragma unroll
        for (int j = 0; j < TN; j++) B[SET][j] = nibbles(cw[K][TM + j]);
    }
    template <int J> __device__ __forceinline__ void step(f32x16 (&c)[TM][TN])
    {
        constexpr int cur = J & 1, nxt = cur ^ 1;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const i32x8 a = __builtin_shufflevector(A[cur][i], A[cur][i], 0, 1, 2, 3, -1, -1, -1, -1);
                const i32x8 b = __builtin_shufflevector(B[cur][j], B[cur][j], 0, 1, 2, 3, -1, -1, -1, -1);
                c[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[i][j], 4, 4, 0, 0, 0, 0);   // scale operands 0: the unscaled, single instruction
            }
        decode<(J + 1) % D, nxt>();        // the next k-step's operands while this one's MFMAs run
        load_words<J % D>();               // words D k-steps ahead (this k-step's were decoded a step ago)
#pragma unroll
        for (int m = 0; m < TM * TN; m++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            if (m % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void prologue()
    {
        load_words<0>(); load_words<1>(); load_words<2>(); load_words<3>();
        decode<0, 0>();
    }
};

__global__ __launch_bounds__(256, 1) void pair_mfma_fp4_miss_kernel(
    const uint32_t *__restrict__ w2, int64_t ncols_pad, int n_s, uint32_t *__restrict__ acc, const int4 *__restrict__ work,
    const unsigned long long *__restrict__ d_missing, int run_if_missing)
{
    typedef Fp4MissPipe P;
    if (d_missing && ((*d_missing != 0ull) != (run_if_missing != 0))) return;
    const int4 item = work[blockIdx.x];
    if (item.w == 0) return;
    const int per = (((n_s + item.w - 1) / item.w) + P::D - 1) / P::D * P::D;      // n_s is a multiple of D (blocks padded to 256 SNPs)
    // (the division runs on the VALU: without readfirstlane the uniform row address -- the buffer descriptor of the word loads --
    // sits in VGPRs and every load becomes a waterfall loop)
    const int s_beg = __builtin_amdgcn_readfirstlane(item.z * per);
    const int s_end = __builtin_amdgcn_readfirstlane((s_beg + per < n_s) ? (s_beg + per) : n_s);
    if (s_beg >= s_end) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, kh = lane >> 5;
    const int row_base = item.x * (64 * P::TM) + wr * (32 * P::TM);
    const int64_t col_base = (int64_t)item.y * (64 * P::TN) + wc * (32 * P::TN);
    P pipe;
    // pair row 2 s + kh of k-step s; a row of pairs is 8 ncols_pad bytes (< 2^31 for every panel that fits a GPU)
    pipe.base = reinterpret_cast<const char *>(w2) + (int64_t)(2 * s_beg) * ncols_pad * 8;
    pipe.offa = (uint32_t)(((int64_t)kh * ncols_pad + row_base + li) * 8);
    pipe.offb = (uint32_t)(((int64_t)kh * ncols_pad + col_base + li) * 8);
    pipe.kstride = 2 * ncols_pad * 8;
    f32x16 c[P::TM][P::TN];
#pragma unroll
    for (int i = 0; i < P::TM; i++)
#pragma unroll
        for (int j = 0; j < P::TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) c[i][j][r] = 0.f;
    pipe.prologue();
    for (int s = s_beg; s < s_end; s += P::D) {
        pipe.step<0>(c); pipe.step<1>(c); pipe.step<2>(c); pipe.step<3>(c);
    }
    // the sums leave the loop IN the accumulation registers: without this the allocator split their live ranges at the loop
    // latch and copied 85 of them to scratch in every round
#pragma unroll
    for (int i = 0; i < P::TM; i++)
#pragma unroll
        for (int j = 0; j < P::TN; j++) asm volatile("" : "+a"(c[i][j]));
#pragma unroll
    for (int i = 0; i < P::TM; i++)
#pragma unroll
        for (int j = 0; j < P::TN; j++) {
            uint32_t *p0 = acc + (int64_t)(row_base + 32 * i + 4 * kh) * ncols_pad + col_base + 32 * j + li;
#pragma unroll
            for (int r = 0; r < 16; r++)
                atomicAdd(p0 + (int64_t)((r & 3) + 8 * (r >> 2)) * ncols_pad, (uint32_t)c[i][j][r]);
        }
}

// ---------------------------------------------------------------------------
// IBS / KING counters of blocks WITHOUT missing calls on the MX-fp4 MFMA (round 4): the two products of I8Scheme<PM_IBS_NOMISS>,
// g.g' and h.h', with e2m1 operands.  A 2-bit code, left where it is, IS the nibble of g / 2 (0b0000 = 0, 0b0001 = 0.5,
// 0b0010 = 1, 0b0011 = 1.5), and in a block without missing calls bit 0 of a code is the het indicator (codes 0, 1, 2; 3 only as
// SNP / sample padding), i.e. the nibble 0.5 h.  Round 6: the UNSCALED instruction (scale operands 0: one issue slot instead of the v_mfma_ld_scale + v_mfma
// pair; +2 ... 5 % on every fp4 kernel) sums g g' / 4 and h h' / 4 -- multiples of 1/4 below 2^18, exact in fp32 -- and the flush multiplies by 4.
// Decode per 16-code word: w & 0x33333333, (w >> 2) & 0x33333333 (g, even / odd SNPs), w & 0x11111111, (w >> 2) & 0x11111111
// (h) -- five VALU per word for both products, 3.75 per MFMA (int8 form: 4.75), and every MFMA takes 64 SNPs instead of 32.
// Padding SNPs (code 3 for every sample) add 9 to g.g' (as in the int8 form) and 1 to h.h': constants of the K part, put
// back by the flush.  Sums exact in fp32 (<= 9 x 2^16 per launch).  Tile, work list, planes and rank-one terms as I8Scheme<PM_IBS_NOMISS>.
struct Fp4NomissPipe {
    static constexpr int TM = 4, TN = 2, R = TM + TN, D = 4;
    const char *base;
    uint32_t offa, offb;
    int64_t kstride;
    uint2 cw[D][R];
    i32x4 G[2][R], H[2][R];

    template <int K> __device__ __forceinline__ void load_words()
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(offa + 256 * i), 0, 0);
            cw[K][i] = make_uint2(v[0], v[1]);
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(offb + 256 * j), 0, 0);
            cw[K][TM + j] = make_uint2(v[0], v[1]);
        }
        base += kstride;
    }
    template <int K, int SET> __device__ __forceinline__ void decode()
    {
#pragma unroll
        for (int g = 0; g < R; g++) {
            const uint32_t x = cw[K][g].x, y = cw[K][g].y, xs = x >> 2, ys = y >> 2;
            G[SET][g][0] = (int)(x & 0x33333333u); G[SET][g][1] = (int)(xs & 0x33333333u);
            G[SET][g][2] = (int)(y & 0x33333333u); G[SET][g][3] = (int)(ys & 0x33333333u);
            H[SET][g][0] = (int)(x & 0x11111111u); H[SET][g][1] = (int)(xs & 0x11111111u);
            H[SET][g][2] = (int)(y & 0x11111111u); H[SET][g][3] = (int)(ys & 0x11111111u);
        }
    }
    static __device__ __forceinline__ i32x8 wide(const i32x4 v) { return __builtin_shufflevector(v, v, 0, 1, 2, 3, -1, -1, -1, -1); }
    template <int J> __device__ __forceinline__ void step(f32x16 (&cg)[TM][TN], f32x16 (&ch)[TM][TN])
    {
        constexpr int cur = J & 1, nxt = cur ^ 1;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
                cg[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wide(G[cur][i]), wide(G[cur][TM + j]), cg[i][j], 4, 4, 0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
                ch[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wide(H[cur][i]), wide(H[cur][TM + j]), ch[i][j], 4, 4, 0, 0, 0, 0);
        decode<(J + 1) % D, nxt>();
        load_words<J % D>();
#pragma unroll
        for (int m = 0; m < 2 * TM * TN; m++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            if (m % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void prologue()
    {
        load_words<0>(); load_words<1>(); load_words<2>(); load_words<3>();
        decode<0, 0>();
    }
};

template <int MODE>
__global__ __launch_bounds__(256, 1) void pair_mfma_fp4_nomiss_kernel(
    const uint32_t *__restrict__ w2, int64_t ncols_pad, int n_s, int n_snp, uint32_t *__restrict__ acc, int64_t acc_plane,
    const int4 *__restrict__ work, const unsigned long long *__restrict__ d_missing)
{
    typedef Fp4NomissPipe P;
    if (*d_missing != 0ull) return;                    // the general kernel takes blocks with missing calls
    const int4 item = work[blockIdx.x];
    if (item.w == 0) return;
    const int per = (((n_s + item.w - 1) / item.w) + P::D - 1) / P::D * P::D;      // n_s is a multiple of D (blocks padded to 256 SNPs)
    // (the division runs on the VALU: without readfirstlane the uniform row address -- the buffer descriptor of the word loads --
    // sits in VGPRs and every load becomes a waterfall loop)
    const int s_beg = __builtin_amdgcn_readfirstlane(item.z * per);
    const int s_end = __builtin_amdgcn_readfirstlane((s_beg + per < n_s) ? (s_beg + per) : n_s);
    if (s_beg >= s_end) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, kh = lane >> 5;
    const int row_base = item.x * (64 * P::TM) + wr * (32 * P::TM);
    const int64_t col_base = (int64_t)item.y * (64 * P::TN) + wc * (32 * P::TN);
    P pipe;
    pipe.base = reinterpret_cast<const char *>(w2) + (int64_t)(2 * s_beg) * ncols_pad * 8;
    pipe.offa = (uint32_t)(((int64_t)kh * ncols_pad + row_base + li) * 8);
    pipe.offb = (uint32_t)(((int64_t)kh * ncols_pad + col_base + li) * 8);
    pipe.kstride = 2 * ncols_pad * 8;
    f32x16 cg[P::TM][P::TN], ch[P::TM][P::TN];
#pragma unroll
    for (int i = 0; i < P::TM; i++)
#pragma unroll
        for (int j = 0; j < P::TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) { cg[i][j][r] = 0.f; ch[i][j][r] = 0.f; }
    pipe.prologue();
    for (int s = s_beg; s < s_end; s += P::D) {
        pipe.step<0>(cg, ch); pipe.step<1>(cg, ch); pipe.step<2>(cg, ch); pipe.step<3>(cg, ch);
    }
#pragma unroll
    for (int i = 0; i < P::TM; i++)
#pragma unroll
        for (int j = 0; j < P::TN; j++) { asm volatile("" : "+a"(cg[i][j])); asm volatile("" : "+a"(ch[i][j])); }
    const int nv_lo = 64 * s_beg, nv_hi = (64 * s_end < n_snp) ? 64 * s_end : n_snp;
    const int nv = (nv_hi > nv_lo) ? (nv_hi - nv_lo) : 0;         // real SNPs of this K part = both-called count of every pair
    const int npad = 64 * (s_end - s_beg) - nv;                   // padding SNPs: 9 each in g.g', 1 each in h.h'
#pragma unroll
    for (int i = 0; i < P::TM; i++)
#pragma unroll
        for (int j = 0; j < P::TN; j++) {
            uint32_t *p0 = acc + (int64_t)(row_base + 32 * i + 4 * kh) * ncols_pad + col_base + 32 * j + li;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                uint32_t *p = p0 + (int64_t)((r & 3) + 8 * (r >> 2)) * ncols_pad;
                const int gg = (int)(4.0f * cg[i][j][r]);         // g.g' + 9 npad  (unscaled products of g / 2, h / 2: x 4, exact)
                const int hh = (int)(4.0f * ch[i][j][r]) - npad;  // h.h'
                if (MODE == PM_IBS_NOMISS) {                      // {n, ibs1 - H_i - H_j, 2 ibs0 - 2 (T_i + T_j)} (+ rank-one terms at settle time)
                    atomicAdd(p, (uint32_t)nv);
                    atomicAdd(p + acc_plane, 0u - 2u * (uint32_t)hh);
                    atomicAdd(p + 2 * acc_plane, (uint32_t)(hh + 9 * npad - gg));
                } else {                                          // PM_HOMO_NOMISS: {ibs1 - H_i - H_j, 2 ibs0 - ...}
                    atomicAdd(p, 0u - 2u * (uint32_t)hh);
                    atomicAdd(p + acc_plane, (uint32_t)(hh + 9 * npad - gg));
                }
            }
        }
}

template <int MODE>
static int launch_fp4_nomiss(hipStream_t st, const int4 *work, int n_blocks, const uint32_t *w2, int64_t ncols_pad, int n_s, int n_snp,
                             uint32_t *acc, int64_t acc_plane, const unsigned long long *d_missing)
{
    if (n_s <= 0 || n_blocks <= 0) return 0;
    // (round 6: a 16x16x128 form of this kernel -- 8 x 4 sub-tiles, het operands made in place, two VALU behind every MFMA -- was built,
    // bit-exact, and measured: K loop 1.70 ms per 65 536-SNP block at N = 10 000 against 1.68, and a flush of 4 x 64-byte pieces per
    // atomic instruction that costs 0.56 ms against 0.23; not kept -- profiles/r06_fp4_16x16x128_ab.txt)
    hipLaunchKernelGGL(pair_mfma_fp4_nomiss_kernel<MODE>, dim3((unsigned)n_blocks), dim3(256), 0, st, w2, ncols_pad, n_s, n_snp, acc,
                       acc_plane, work, d_missing);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// The general IBS / KING-robust counters (blocks WITH missing calls) on the MX-fp4 MFMA (round 4): the products of
// I8Scheme<PM_IBS> / <PM_KING_ROBUST> with e2m1 operands built by bit logic instead of v_perm_b32 table lookups.  Per nibble (one
// SNP; code bits b1 b0, x = the word or the word >> 2, t = x >> 1, M = 0x11111111):
//     P = x & M (b0)    m = P & t (missing)    v = M ^ m (called)    h = P ^ m (het)    y = M ^ P (homozygous, called)
//     e2 = t & y (g == 2)     s = v | h << 3  (= +-1/2: v - 2 h)      x = y | e2 << 3  (= +-1/2: [g == 0] - [g == 2])
// (bit 0 of a nibble = 1/2, bit 3 = the sign; unscaled instruction, the flush multiplies the sums of quarter products by 4), 7 / 9 VALU per eight SNPs for KING's three /
// IBS's four value types.  One wave per SIMD, operand sets double-buffered, four word sets in flight; every MFMA takes 64 SNPs.
//   IBS   64 x 64 per wave: v.v', s.s', y.y', x.x' -> {nvalid, ibs1 = (nvalid - s.s') / 2, 2 ibs0 = y.y' - x.x'}
//   KING  32 x 64 per wave: y.y', x.x', y.h', h.y', h.h' -> the five counters as I8Scheme<PM_KING_ROBUST>::emit
// Sums exact in fp32 (|sum| <= 2^16 per launch).
template <int MODE> struct Fp4Scheme;
// types(x, t, x2, x3, o): x = the word or the word >> 2 (code bits b1 b0 at the nibble's bits 1 0), t = x >> 1 (b1 at bit 0),
// x2 = x << 2 (b1 at bit 3), x3 = x << 3 (b0 at bit 3); M = bit 0, M8 = bit 3 of every nibble.  The sign bit of a nibble whose
// magnitude is 0 is free (-0 = 0), so s takes b0 and x takes b1 as sign without masking them by "called" / "homozygous":
// every value type is ONE three-input boolean instruction (v_bitop3_b32) on top of the shifts.
template <> struct Fp4Scheme<PM_IBS> {
    static constexpr int NS = 4, NA = 4, NT = 4, TM = 2, TN = 2, C = 3, WPS = 1;
    static constexpr bool NEED_X3 = true;
    static __device__ __forceinline__ constexpr int ta(int s) { return s; }
    static __device__ __forceinline__ constexpr int tb(int s) { return s; }
    static __device__ __forceinline__ constexpr int acc(int s) { return s; }
    static __device__ __forceinline__ void types(uint32_t x, uint32_t t, uint32_t x2, uint32_t x3, int (&o)[NT])   // {v, s, y, x}
    {
        const uint32_t M = 0x11111111u, M8 = 0x88888888u, v = M & ~(x & t), y = M & ~x;
        o[0] = (int)v; o[1] = (int)(v | (x3 & M8)); o[2] = (int)y; o[3] = (int)(y | (x2 & M8));
    }
    static __device__ __forceinline__ void emit(const int *a, uint32_t *cnt)         // {nvalid, ibs1, 2 ibs0}
    {
        cnt[0] = (uint32_t)a[0]; cnt[1] = (uint32_t)(a[0] - a[1]) >> 1; cnt[2] = (uint32_t)(a[2] - a[3]);
    }
};
template <> struct Fp4Scheme<PM_KING_ROBUST> {
    static constexpr int NS = 5, NA = 5, NT = 3, TM = 1, TN = 2, C = 5;
#ifndef FP4_KING_WPS
#define FP4_KING_WPS 1     /* 2: 128 + 128 registers, 296 bytes of scratch, 62 instead of 5.5 ms */
#endif
    static constexpr int WPS = FP4_KING_WPS;
    static constexpr bool NEED_X3 = false;
    static __device__ __forceinline__ constexpr int ta(int s) { return s == 0 ? 0 : s == 1 ? 1 : s == 2 ? 0 : 2; }   // y x y h h
    static __device__ __forceinline__ constexpr int tb(int s) { return s == 0 ? 0 : s == 1 ? 1 : s == 2 ? 2 : s == 3 ? 0 : 2; }   // y x h y h
    static __device__ __forceinline__ constexpr int acc(int s) { return s; }
    static __device__ __forceinline__ void types(uint32_t x, uint32_t t, uint32_t x2, uint32_t, int (&o)[NT])        // {y, x, h}
    {
        const uint32_t M = 0x11111111u, M8 = 0x88888888u, y = M & ~x;
        o[0] = (int)y; o[1] = (int)(y | (x2 & M8)); o[2] = (int)(M & x & ~t);
    }
    static __device__ __forceinline__ void emit(const int *a, uint32_t *cnt)         // {nLoci, ibs1, 2 ibs0, N1_Aa, N2_Aa}
    {
        cnt[0] = (uint32_t)(a[0] + a[2] + a[3] + a[4]); cnt[1] = (uint32_t)(a[2] + a[3]);
        cnt[2] = (uint32_t)(a[0] - a[1]); cnt[3] = (uint32_t)(a[3] + a[4]); cnt[4] = (uint32_t)(a[2] + a[4]);
    }
};

// individual beta: y.y', x.x', v.v' -> {num = v.v', at least one het = v.v' - y.y', equal homozygotes = (y.y' + x.x') / 2}
template <> struct Fp4Scheme<PM_BETA> {
    static constexpr int NS = 3, NA = 3, NT = 3, TM = 2, TN = 2, C = 3, WPS = 1;
    static constexpr bool NEED_X3 = false;
    static __device__ __forceinline__ constexpr int ta(int s) { return s; }
    static __device__ __forceinline__ constexpr int tb(int s) { return s; }
    static __device__ __forceinline__ constexpr int acc(int s) { return s; }
    static __device__ __forceinline__ void types(uint32_t x, uint32_t t, uint32_t x2, uint32_t, int (&o)[NT])        // {y, x, v}
    {
        const uint32_t M = 0x11111111u, M8 = 0x88888888u, y = M & ~x;
        o[0] = (int)y; o[1] = (int)(y | (x2 & M8)); o[2] = (int)(M & ~(x & t));
    }
    static __device__ __forceinline__ void emit(const int *a, uint32_t *cnt)
    {
        cnt[0] = (uint32_t)a[2]; cnt[1] = (uint32_t)(a[2] - a[0]); cnt[2] = (uint32_t)(a[0] + a[1]) >> 1;
    }
};

// KING-homo (blocks with missing calls): h.y' + y.h' = ibs1 in ONE accumulator, y.y' and x.x' -> 2 ibs0 = y.y' - x.x'
template <> struct Fp4Scheme<PM_KING_HOMO> {
    static constexpr int NS = 4, NA = 3, NT = 3, TM = 2, TN = 2, C = 2, WPS = 1;
    static constexpr bool NEED_X3 = false;
    static __device__ __forceinline__ constexpr int ta(int s) { return s == 0 ? 2 : s == 1 ? 0 : s == 2 ? 0 : 1; }   // h y y x
    static __device__ __forceinline__ constexpr int tb(int s) { return s == 0 ? 0 : s == 1 ? 2 : s == 2 ? 0 : 1; }   // y h y x
    static __device__ __forceinline__ constexpr int acc(int s) { return s < 2 ? 0 : s - 1; }
    static __device__ __forceinline__ void types(uint32_t x, uint32_t t, uint32_t x2, uint32_t, int (&o)[NT])        // {y, x, h}
    {
        const uint32_t M = 0x11111111u, M8 = 0x88888888u, y = M & ~x;
        o[0] = (int)y; o[1] = (int)(y | (x2 & M8)); o[2] = (int)(M & x & ~t);
    }
    static __device__ __forceinline__ void emit(const int *a, uint32_t *cnt)         // {ibs1, 2 ibs0}
    {
        cnt[0] = (uint32_t)a[0]; cnt[1] = (uint32_t)(a[1] - a[2]);
    }
};

template <int MODE> struct Fp4GenPipe {
    typedef Fp4Scheme<MODE> S;
    static constexpr int TM = S::TM, TN = S::TN, R = TM + TN, NT = S::NT, NS = S::NS, NA = S::NA, D = 4;
    const char *base;
    uint32_t offa, offb;
    int64_t kstride;
    uint2 cw[D][R];
    i32x4 V[2][R][NT];

    template <int K> __device__ __forceinline__ void load_words()
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(offa + 256 * i), 0, 0);
            cw[K][i] = make_uint2(v[0], v[1]);
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(offb + 256 * j), 0, 0);
            cw[K][TM + j] = make_uint2(v[0], v[1]);
        }
        base += kstride;
    }
    // decode unit U = (row group U / 2, word U % 2): 16 SNPs -> dwords 2 (U % 2) and 2 (U % 2) + 1 of every value type
    template <int K, int SET, int U> __device__ __forceinline__ void decode_unit()
    {
        constexpr int g = U / 2, hw = U % 2;
        const uint32_t w = hw ? cw[K][g].y : cw[K][g].x;
        int o[NT];
        S::types(w, w >> 1, w << 2, S::NEED_X3 ? (w << 3) : 0u, o);            // even SNPs of the word
#pragma unroll
        for (int t = 0; t < NT; t++) V[SET][g][t][2 * hw] = o[t];
        S::types(w >> 2, w >> 3, w, S::NEED_X3 ? (w << 1) : 0u, o);            // odd SNPs
#pragma unroll
        for (int t = 0; t < NT; t++) V[SET][g][t][2 * hw + 1] = o[t];
    }
    template <int K, int SET, int U0, int U1> __device__ __forceinline__ void decode_units()
    {
        if constexpr (U0 < U1) { decode_unit<K, SET, U0>(); decode_units<K, SET, U0 + 1, U1>(); }
    }
    template <int K, int SET> __device__ __forceinline__ void decode() { decode_units<K, SET, 0, 2 * R>(); }
    static __device__ __forceinline__ i32x8 wide(const i32x4 v) { return __builtin_shufflevector(v, v, 0, 1, 2, 3, -1, -1, -1, -1); }
    // phase PH of k-step J: the MFMAs of product PH, a share of the next k-step's decode units (and, in phase 0, the word loads
    // D k-steps ahead); a scheduling barrier per phase keeps the VALU work spread under the MFMAs
    template <int J, int PH> __device__ __forceinline__ void phase(f32x16 (&c)[NA][TM][TN])
    {
        constexpr int cur = J & 1, nxt = cur ^ 1;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
                c[S::acc(PH)][i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wide(V[cur][i][S::ta(PH)]), wide(V[cur][TM + j][S::tb(PH)]),
                                                                                     c[S::acc(PH)][i][j], 4, 4, 0, 0, 0, 0);
        constexpr int u0 = PH * 2 * R / NS, u1 = (PH + 1) * 2 * R / NS;
        decode_units<(J + 1) % D, nxt, u0, u1>();
        if (PH == 0) load_words<J % D>();
        constexpr int n_valu = (u1 - u0) * (S::NEED_X3 ? 14 : 10);
        constexpr int per = (n_valu + TM * TN - 1) / (TM * TN);
#pragma unroll
        for (int m = 0; m < TM * TN; m++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
            if (PH == 0) __builtin_amdgcn_sched_group_barrier(0x020, (R + TM * TN - 1) / (TM * TN), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int J, int PH> __device__ __forceinline__ void phases(f32x16 (&c)[NA][TM][TN])
    {
        if constexpr (PH < NS) { phase<J, PH>(c); phases<J, PH + 1>(c); }
    }
    template <int J> __device__ __forceinline__ void step(f32x16 (&c)[NA][TM][TN]) { phases<J, 0>(c); }
    __device__ __forceinline__ void prologue()
    {
        load_words<0>(); load_words<1>(); load_words<2>(); load_words<3>();
        decode<0, 0>();
    }
};

template <int MODE>
__global__ __launch_bounds__(256, Fp4Scheme<MODE>::WPS) void pair_mfma_fp4_kernel(
    const uint32_t *__restrict__ w2, int64_t ncols_pad, int n_s, uint32_t *__restrict__ acc, int64_t acc_plane,
    const int4 *__restrict__ work, const unsigned long long *__restrict__ d_missing)
{
    typedef Fp4GenPipe<MODE> P;
    typedef Fp4Scheme<MODE> S;
    if (d_missing && *d_missing == 0ull) return;       // the two-product kernel takes blocks without missing calls
    const int4 item = work[blockIdx.x];
    if (item.w == 0) return;
    const int per = (((n_s + item.w - 1) / item.w) + P::D - 1) / P::D * P::D;
    const int s_beg = __builtin_amdgcn_readfirstlane(item.z * per);
    const int s_end = __builtin_amdgcn_readfirstlane((s_beg + per < n_s) ? (s_beg + per) : n_s);
    if (s_beg >= s_end) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, kh = lane >> 5;
    const int row_base = item.x * (64 * P::TM) + wr * (32 * P::TM);
    const int64_t col_base = (int64_t)item.y * (64 * P::TN) + wc * (32 * P::TN);
    P pipe;
    pipe.base = reinterpret_cast<const char *>(w2) + (int64_t)(2 * s_beg) * ncols_pad * 8;
    pipe.offa = (uint32_t)(((int64_t)kh * ncols_pad + row_base + li) * 8);
    pipe.offb = (uint32_t)(((int64_t)kh * ncols_pad + col_base + li) * 8);
    pipe.kstride = 2 * ncols_pad * 8;
    f32x16 c[P::NA][P::TM][P::TN];
#pragma unroll
    for (int a = 0; a < P::NA; a++)
#pragma unroll
        for (int i = 0; i < P::TM; i++)
#pragma unroll
            for (int j = 0; j < P::TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) c[a][i][j][r] = 0.f;
    pipe.prologue();
    for (int s = s_beg; s < s_end; s += P::D) {
        pipe.template step<0>(c); pipe.template step<1>(c); pipe.template step<2>(c); pipe.template step<3>(c);
    }
#pragma unroll
    for (int a = 0; a < P::NA; a++)
#pragma unroll
        for (int i = 0; i < P::TM; i++)
#pragma unroll
            for (int j = 0; j < P::TN; j++) asm volatile("" : "+a"(c[a][i][j]));
#pragma unroll
    for (int i = 0; i < P::TM; i++)
#pragma unroll
        for (int j = 0; j < P::TN; j++) {
            uint32_t *p0 = acc + (int64_t)(row_base + 32 * i + 4 * kh) * ncols_pad + col_base + 32 * j + li;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int a[P::NA];
                uint32_t cnt[S::C];
#pragma unroll
                for (int k = 0; k < P::NA; k++) a[k] = (int)(4.0f * c[k][i][j][r]);   // unscaled products of halves: x 4, exact
                S::emit(a, cnt);
                uint32_t *p = p0 + (int64_t)((r & 3) + 8 * (r >> 2)) * ncols_pad;
#pragma unroll
                for (int k = 0; k < S::C; k++) atomicAdd(p + (int64_t)k * acc_plane, cnt[k]);
            }
        }
}

template <int MODE>
static int launch_fp4_gen(hipStream_t st, const int4 *work, int n_blocks, const uint32_t *w2, int64_t ncols_pad, int n_s, uint32_t *acc,
                          int64_t acc_plane, const unsigned long long *d_missing)
{
    if (n_s <= 0 || n_blocks <= 0) return 0;
    hipLaunchKernelGGL(pair_mfma_fp4_kernel<MODE>, dim3((unsigned)n_blocks), dim3(256), 0, st, w2, ncols_pad, n_s, acc, acc_plane, work,
                       d_missing);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// tile and workgroups per CU of the fp4 form of a kind's general kernel (0: the kind has none)
bool pair_fp4_tile(int mode, int *tile_r, int *tile_c, int *wg_per_cu)
{
    if (mode == PM_IBS) { *tile_r = 64 * Fp4Scheme<PM_IBS>::TM; *tile_c = 64 * Fp4Scheme<PM_IBS>::TN; }
    else if (mode == PM_KING_ROBUST) { *tile_r = 64 * Fp4Scheme<PM_KING_ROBUST>::TM; *tile_c = 64 * Fp4Scheme<PM_KING_ROBUST>::TN; }
    else if (mode == PM_BETA) { *tile_r = 64 * Fp4Scheme<PM_BETA>::TM; *tile_c = 64 * Fp4Scheme<PM_BETA>::TN; }
    else if (mode == PM_KING_HOMO) { *tile_r = 64 * Fp4Scheme<PM_KING_HOMO>::TM; *tile_c = 64 * Fp4Scheme<PM_KING_HOMO>::TN; }
    else return false;
    if (wg_per_cu) *wg_per_cu = (mode == PM_KING_ROBUST) ? Fp4Scheme<PM_KING_ROBUST>::WPS : 1;
    return true;
}

int launch_pair_fp4_miss(hipStream_t st, const int4 *work, int n_blocks, const uint32_t *w2, int64_t ncols_pad, int n_s,
                         uint32_t *acc, const unsigned long long *d_missing)
{
    if (n_s <= 0 || n_blocks <= 0) return 0;
    hipLaunchKernelGGL(pair_mfma_fp4_miss_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, w2, ncols_pad, n_s, acc, work,
                       d_missing, 1);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int MODE> static void i8_tile_of(int *tile_r, int *tile_c, int *wg_per_cu)
{
    *tile_r = 64 * I8Scheme<MODE>::TM; *tile_c = 64 * I8Scheme<MODE>::TN;
    if (wg_per_cu) *wg_per_cu = I8Scheme<MODE>::WPS;
}

void pair_i8_tile(int mode, int *tile_r, int *tile_c, int *wg_per_cu)
{
    switch (mode) {
    case PM_IBS: return i8_tile_of<PM_IBS>(tile_r, tile_c, wg_per_cu);
    case PM_KING_ROBUST: return i8_tile_of<PM_KING_ROBUST>(tile_r, tile_c, wg_per_cu);
    case PM_KING_HOMO: return i8_tile_of<PM_KING_HOMO>(tile_r, tile_c, wg_per_cu);
    case PM_GCTA_MISS: return i8_tile_of<PM_GCTA_MISS>(tile_r, tile_c, wg_per_cu);
    case PM_BETA: return i8_tile_of<PM_BETA>(tile_r, tile_c, wg_per_cu);
    default: return i8_tile_of<PM_IBS_NOMISS>(tile_r, tile_c, wg_per_cu);
    }
}

// d_missing != nullptr (IBS, KING-robust): two launches, one of them exits at once -- blocks without missing calls
// take the binary 3-product form on its own work list (128 x 128 tiles).
int launch_pair_i8(hipStream_t st, int mode, const int4 *work, int n_blocks, const uint32_t *w2, int64_t ncols_pad,
                   int n_q, int n_snp, uint32_t *acc, int64_t acc_plane, const unsigned long long *d_missing,
                   const int4 *work_nm, int n_blocks_nm, bool fp4_nomiss, bool fp4_general)
{
    // fp4_nomiss: blocks without missing calls take the MX-fp4 form of the two-product kernel (n_q is a multiple of 8 then)
    if (n_q <= 0 || n_blocks <= 0) return 0;
    const unsigned long long *nf = nullptr;
    switch (mode) {
    case PM_IBS:
        if (fp4_general ? launch_fp4_gen<PM_IBS>(st, work, n_blocks, w2, ncols_pad, n_q / 2, acc, acc_plane, d_missing)
                        : launch_i8<PM_IBS>(st, work, n_blocks, w2, ncols_pad, n_q, n_snp, acc, acc_plane, d_missing, 1)) return 1;
        if (d_missing && fp4_nomiss) return launch_fp4_nomiss<PM_IBS_NOMISS>(st, work_nm, n_blocks_nm, w2, ncols_pad, n_q / 2, n_snp, acc, acc_plane, d_missing);
        return d_missing ? launch_i8<PM_IBS_NOMISS>(st, work_nm, n_blocks_nm, w2, ncols_pad, n_q, n_snp, acc, acc_plane, d_missing, 0) : 0;
    case PM_KING_ROBUST:
        if (fp4_general ? launch_fp4_gen<PM_KING_ROBUST>(st, work, n_blocks, w2, ncols_pad, n_q / 2, acc, acc_plane, d_missing)
                        : launch_i8<PM_KING_ROBUST>(st, work, n_blocks, w2, ncols_pad, n_q, n_snp, acc, acc_plane, d_missing, 1)) return 1;
        if (d_missing && fp4_nomiss) return launch_fp4_nomiss<PM_IBS_NOMISS>(st, work_nm, n_blocks_nm, w2, ncols_pad, n_q / 2, n_snp, acc, acc_plane, d_missing);
        return d_missing ? launch_i8<PM_IBS_NOMISS>(st, work_nm, n_blocks_nm, w2, ncols_pad, n_q, n_snp, acc, acc_plane, d_missing, 0) : 0;
    case PM_KING_HOMO:
        if (!d_missing)
            return fp4_general ? launch_fp4_gen<PM_KING_HOMO>(st, work, n_blocks, w2, ncols_pad, n_q / 2, acc, acc_plane, nf)
                               : launch_i8<PM_KING_HOMO>(st, work, n_blocks, w2, ncols_pad, n_q, n_snp, acc, acc_plane, nf, 0);
        if (fp4_general ? launch_fp4_gen<PM_KING_HOMO>(st, work, n_blocks, w2, ncols_pad, n_q / 2, acc, acc_plane, d_missing)
                        : launch_i8<PM_KING_HOMO>(st, work, n_blocks, w2, ncols_pad, n_q, n_snp, acc, acc_plane, d_missing, 1)) return 1;
        if (fp4_nomiss) return launch_fp4_nomiss<PM_HOMO_NOMISS>(st, work_nm, n_blocks_nm, w2, ncols_pad, n_q / 2, n_snp, acc, acc_plane, d_missing);
        return launch_i8<PM_HOMO_NOMISS>(st, work_nm, n_blocks_nm, w2, ncols_pad, n_q, n_snp, acc, acc_plane, d_missing, 0);
    case PM_BETA:
        if (fp4_general) return launch_fp4_gen<PM_BETA>(st, work, n_blocks, w2, ncols_pad, n_q / 2, acc, acc_plane, nf);
        return launch_i8<PM_BETA>(st, work, n_blocks, w2, ncols_pad, n_q, n_snp, acc, acc_plane, nf, 0);
    case PM_GCTA_MISS:   // only for blocks that hold missing calls
        return launch_i8<PM_GCTA_MISS>(st, work, n_blocks, w2, ncols_pad, n_q, n_snp, acc, acc_plane, d_missing, 1);
    }
    set_error("launch_pair_i8: bad mode");
    return 1;
}

}  // namespace snpgpu
