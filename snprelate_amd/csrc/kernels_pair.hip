// The two N x N pairwise accumulators (the hot path):
//
//  pair_popcount_kernel  IBS / KING counters as wavefront bit-ops
//      (replaces CIBSCount::thread_ibs_num src/genIBS.cpp:154-273,
//       CKINGRobust::thread_ibs_num src/genKING.cpp:292-426, the integer half of
//       CKINGHomo::thread_ibs_num src/genKING.cpp:66-200, and the serial missing-denominator
//       loop of CGCTA_AlgArith::Run src/genPCA.cpp:1201-1224)
//  syrk_mfma_kernel      centred/scaled genotype outer products on fp32 MFMA with on-the-fly
//      2-bit decode (replaces CProdMat_AlgArith::MulAdd src/genPCA.cpp:229-312 and the
//      TransposeGenotype/GenoSub/GenoMul preparation src/genPCA.h:93-108, genPCA.cpp:315-368;
//      with the KING-homo tables also the masked p(1-p) sums of src/genKING.cpp:115-154)
#include "snpgpu_internal.h"

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
template <int MODE> struct PairOps;

template <> struct PairOps<PM_IBS> {   // 8 VALU ops / 32 SNP pairs
    typedef uint4 PV;
    static constexpr int C = 3;       // {nvalid, ibs1, ibs0}
    static __device__ __forceinline__ void run(const uint4 &r, const uint4 &c, uint32_t *cnt)
    {
        const uint32_t t0 = r.x & c.x;                     // both called
        cnt[0] += __popc(t0);
        cnt[1] += __popc((r.y ^ c.y) & t0);                // exactly one heterozygous -> IBS1
        cnt[2] += __popc((r.z & c.w) | (r.w & c.z));       // opposite homozygotes     -> IBS0
    }
};
template <> struct PairOps<PM_KING_ROBUST> {   // 11 VALU ops / 32 SNP pairs
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
        cnt[2] += __popc((r.z & c.w) | (r.w & c.z));
    }
};
template <> struct PairOps<PM_KING_HOMO> {     // 7 ops
    typedef uint4 PV;
    static constexpr int C = 2;       // {ibs1, ibs0}
    static __device__ __forceinline__ void run(const uint4 &r, const uint4 &c, uint32_t *cnt)
    {
        cnt[0] += __popc((r.y ^ c.y) & (r.x & c.x));
        cnt[1] += __popc((r.z & c.w) | (r.w & c.z));
    }
};
template <> struct PairOps<PM_GCTA_MISS> {     // 2 ops per 32 SNP pairs, uint2 = 64 SNPs
    typedef uint2 PV;
    static constexpr int C = 1;       // {both missing at a polymorphic SNP}
    static __device__ __forceinline__ void run(const uint2 &r, const uint2 &c, uint32_t *cnt)
    {
        cnt[0] += __popc(r.x & c.x);
        cnt[0] += __popc(r.y & c.y);
    }
};

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
    const int row_base = t.tr * PC_TILE_R + wave * A;          // wave-uniform
    const int64_t col_base = (int64_t)t.tc * PC_TILE_C + lane;

    uint32_t cnt[A][BC][C];
#pragma unroll
    for (int a = 0; a < A; a++)
#pragma unroll
        for (int b = 0; b < BC; b++)
#pragma unroll
            for (int c = 0; c < C; c++) cnt[a][b][c] = 0;

    const PV *__restrict__ rp = rowp + (int64_t)row_base * KWv;
    const PV *__restrict__ cp = colp + col_base;

    PV cv[BC];
#pragma unroll
    for (int b = 0; b < BC; b++) cv[b] = cp[b * 64];
    for (int kw = 0; kw < KWv; kw++) {
        PV cur[BC];
#pragma unroll
        for (int b = 0; b < BC; b++) cur[b] = cv[b];
        if (kw + 1 < KWv) {  // prefetch the next column words while this word is consumed
            const PV *nx = cp + (int64_t)(kw + 1) * ncols_pad;
#pragma unroll
            for (int b = 0; b < BC; b++) cv[b] = nx[b * 64];
        }
#pragma unroll
        for (int a = 0; a < A; a++) {
            const PV rv = rp[(int64_t)a * KWv + kw];          // uniform address -> scalar load
#pragma unroll
            for (int b = 0; b < BC; b++) PairOps<MODE>::run(rv, cur[b], cnt[a][b]);
        }
    }
    // accumulate into the panel's counters (each tile is owned by exactly one workgroup per launch)
#pragma unroll
    for (int a = 0; a < A; a++)
#pragma unroll
        for (int b = 0; b < BC; b++) {
            uint32_t *p = acc + (int64_t)(row_base + a) * ncols_pad + col_base + b * 64;
#pragma unroll
            for (int c = 0; c < C; c++) p[(int64_t)c * acc_plane] += cnt[a][b][c];
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
    }
    set_error("launch_pair_popcount: bad mode");
    return 1;
}

// ---------------------------------------------------------------------------
// SYRK on fp32 MFMA.  Workgroup = 4 waves (2x2), tile 128 x 128, each wave 64 x 64 = 2x2
// v_mfma_f32_32x32x2_f32 tiles (4 independent accumulators keep the matrix pipe issuing).
// Per stage of MM_KC = 32 SNPs each thread fetches ONE 32-bit word (16 samples of one SNP) per
// operand straight from the SNP-major 2-bit rows, decodes it with that SNP's table
// z(g) = x + g*y (missing -> 0) and stores 16 floats to LDS laid out [snp][sample].
// MFMA operand fetch is ds_read_b32: lane l needs Z[sample = l&31][snp = 2*step + (l>>5)].
// fp32 accumulation runs for MM_PROMOTE SNPs, is then promoted into fp64 registers; the fp64
// partial is added to the panel accumulator once per launch.
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void decode16_store(uint32_t w, float x, float y, float *__restrict__ dst, int rot)
{
    // dst -> 16 consecutive floats; the four float4 are written in a lane-rotated order so that the
    // 8 lanes of a ds_write_b128 group hit 8 distinct 16-byte bank slots.
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int qq = (q + rot) & 3;
        const uint32_t sub = (w >> (8 * qq)) & 0xFFu;
        float4 v;
        uint32_t c;
        c = sub & 3u;        v.x = (c == 3u) ? 0.f : fmaf((float)c, y, x);
        c = (sub >> 2) & 3u; v.y = (c == 3u) ? 0.f : fmaf((float)c, y, x);
        c = (sub >> 4) & 3u; v.z = (c == 3u) ? 0.f : fmaf((float)c, y, x);
        c = (sub >> 6) & 3u; v.w = (c == 3u) ? 0.f : fmaf((float)c, y, x);
        *reinterpret_cast<float4 *>(dst + 4 * qq) = v;
    }
}

__global__ __launch_bounds__(256, 2) void syrk_mfma_kernel(
    const uint8_t *__restrict__ packed, int64_t RB, int64_t col0, const float4 *__restrict__ lut, int n_stage,
    double *__restrict__ acc, int64_t ld, const int *__restrict__ prefix, const int *__restrict__ first, int n_sr,
    int n_super, int n_tr, int n_tc)
{
    const TileCoord t = map_tile(prefix, first, n_sr, n_super, MM_SUPER, n_tr, n_tc, MM_TILE, MM_TILE);
    if (!t.valid) return;
    __shared__ float smem[2][2][MM_KC][MM_TILE];  // [buffer][operand A/B][snp][sample]  64 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;       // wave position in the 2x2 grid
    const int li = lane & 31, kh = lane >> 5;

    // staging role: thread -> (snp ks in stage, 16-sample word wi)
    const int ks = tid >> 3, wi = tid & 7;
    const int64_t RBw = RB >> 2;
    const uint32_t *__restrict__ gA = reinterpret_cast<const uint32_t *>(packed) + (int64_t)ks * RBw +
                                      ((col0 + (int64_t)t.tr * MM_TILE) >> 4) + wi;
    const uint32_t *__restrict__ gB = reinterpret_cast<const uint32_t *>(packed) + (int64_t)ks * RBw +
                                      ((col0 + (int64_t)t.tc * MM_TILE) >> 4) + wi;
    const bool diag = (t.tr == t.tc);
    const int rot = wi >> 1;

    f32x16 c32[2][2];
    double c64[2][2][16];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
#pragma unroll
            for (int r = 0; r < 16; r++) { c32[i][j][r] = 0.f; c64[i][j][r] = 0.0; }
        }

    // prologue: stage 0
    uint32_t wa = gA[0], wb = diag ? 0u : gB[0];
    float4 lt = lut[ks];
    decode16_store(wa, lt.x, lt.y, &smem[0][0][ks][wi * 16], rot);
    if (!diag) decode16_store(wb, lt.x, lt.y, &smem[0][1][ks][wi * 16], rot);
    __syncthreads();

    int since_promote = 0;
    for (int s = 0; s < n_stage; s++) {
        const int cur = s & 1;
        const bool more = (s + 1 < n_stage);
        if (more) {  // issue next stage's global loads before the MFMA block
            const int64_t off = (int64_t)(s + 1) * MM_KC * RBw;
            wa = gA[off];
            if (!diag) wb = gB[off];
            lt = lut[(s + 1) * MM_KC + ks];
        }
        const float *__restrict__ As = &smem[cur][0][0][0];
        const float *__restrict__ Bs = diag ? As : &smem[cur][1][0][0];
#pragma unroll
        for (int kk = 0; kk < MM_KC / 2; kk++) {
            const int krow = (2 * kk + kh) * MM_TILE;
            const float a0 = As[krow + wr * 64 + li];
            const float a1 = As[krow + wr * 64 + 32 + li];
            const float b0 = Bs[krow + wc * 64 + li];
            const float b1 = Bs[krow + wc * 64 + 32 + li];
            c32[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c32[0][0], 0, 0, 0);
            c32[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c32[0][1], 0, 0, 0);
            c32[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c32[1][0], 0, 0, 0);
            c32[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c32[1][1], 0, 0, 0);
        }
        since_promote += MM_KC;
        if (since_promote >= MM_PROMOTE || !more) {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        c64[i][j][r] += (double)c32[i][j][r];
                        c32[i][j][r] = 0.f;
                    }
            since_promote = 0;
        }
        if (more) {
            decode16_store(wa, lt.x, lt.y, &smem[cur ^ 1][0][ks][wi * 16], rot);
            if (!diag) decode16_store(wb, lt.x, lt.y, &smem[cur ^ 1][1][ks][wi * 16], rot);
        }
        __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int64_t row_t = (int64_t)t.tr * MM_TILE + wr * 64;
    const int64_t col_t = (int64_t)t.tc * MM_TILE + wc * 64;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                double *p = acc + (row_t + i * 32 + row) * ld + col_t + j * 32 + li;
                *p += c64[i][j][r];
            }
}

int launch_syrk(hipStream_t st, const TileGrid &tg, const uint8_t *packed, int64_t RB, int64_t col0,
                const float4 *lut, int64_t n_snp_pad, double *acc, int64_t ld)
{
    const int n_stage = (int)(n_snp_pad / MM_KC);
    if (n_stage <= 0) return 0;
    hipLaunchKernelGGL(syrk_mfma_kernel, dim3((unsigned)tg.grid), dim3(256), 0, st, packed, RB, col0, lut, n_stage,
                       acc, ld, tg.d_prefix, tg.d_first, tg.n_sr, tg.n_super, tg.n_tr, tg.n_tc);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace snpgpu
