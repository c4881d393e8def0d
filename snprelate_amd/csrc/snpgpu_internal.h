// Internal declarations shared by the HIP translation units of libsnpgpu.
// Device data layout (all per context, see DESIGN.md 3, "Data layout in HBM"):
//
//   packed   uint8  [B][RB]            2-bit genotypes of the current feed block, SNP-major,
//                                      RB = round_up(N,256)/4 bytes per SNP, samples >= N are 3
//   sum,num  int32  [B]                per-SNP genotype sum / non-missing count over all N
//   lut      float2 [nlut][Bpad/2][16] per-SNP-pair decode table: entry c0 + 4*c1 = (z_2p(c0), z_2p+1(c1))
//   wt       uint32 [Bpad/8][ncols_pad] sample-major pair-coded words (byte p = 8*(c0+4*c1) of SNPs 8d+2p, 8d+2p+1)
//   w2       uint32 [Bpad/16][ncols_pad] sample-major 2-bit words (code of SNP 16d+m at bits 2m), int8-MFMA pair kernel
//   rowp     PV     [rows_pad/8][KW][8] bit planes of the panel's row samples, 8 rows of one word adjacent
//   colp     PV     [KW][ncols_pad]    word-major bit planes of the panel's column samples
//   acc_u32  uint32 [C][rows_pad][ld]  pair counters,   rectangular panel, ld = ncols_pad
//   acc_f64  double [S][rows_pad][ld]  pair fp64 sums,  rectangular panel
//
// A panel covers sample rows [row0,row1) and columns [col0,N) with col0 = row0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <utility>
#include <vector>

#include "../../include/snpgpu.h"

struct snpgpu_ctx;

namespace snpgpu {

constexpr int PANEL_ALIGN = 256;   // row0 / padded extents are multiples of this
// Layout of the fp64 accumulator planes (round 3): TILE-MAJOR, 256 x 256 tiles of 512 KB, rows of a tile 2 KB apart -- the
// fp64 flush of a SYRK wave (128 rows x 32-column pieces) and the eigen solver's panel product (64-row strips) then work
// inside a few hundred KB instead of striding through rows 8 * ncols_pad bytes (800 KB at N = 100 000, 4 MB at 500 000)
// apart.  tiles_c = ncols_pad / 256 (0: row-major with leading dimension ld -- SNPGPU_ACC_LAYOUT=row, and the rocBLAS form
// of the panel product).  Panel-relative (r, c) -> element offset:
constexpr int ACC_TILE = 256;
__host__ __device__ __forceinline__ int64_t acc_off(int64_t ld, int64_t tiles_c, int64_t r, int64_t c)
{
    return tiles_c ? ((((r >> 8) * tiles_c + (c >> 8)) << 16) + ((r & 255) << 8) + (c & 255)) : (r * ld + c);
}
constexpr int PC_ROWS_PER_WAVE = 8;
constexpr int PC_WAVES = 4;
constexpr int PC_TILE_R = PC_ROWS_PER_WAVE * PC_WAVES;  // 32 rows per workgroup
constexpr int PC_COLS_PER_LANE = 2;
constexpr int PC_TILE_C = 64 * PC_COLS_PER_LANE;         // 128 columns per workgroup
constexpr int PC_SUPER = 8;                              // 8x8 workgroup tiles per XCD super-tile
constexpr int MM_TILE_R = 128;                           // SYRK workgroup tile: 128 rows x 128 columns
constexpr int MM_TILE_C = 128;                           //   (4 waves as 2x2, each 64 x 64 = 2x2 MFMA 32x32 tiles)
constexpr int MM_PROMOTE = 1024;                         // SNPs accumulated in fp32 (one rounding per 2 SNPs) before the fp64 flush
constexpr int MM_LUTCH = 256;                            // SNPs per LDS-resident decode-table chunk (128 pairs x 128 B)
constexpr int MM_SUPER = 4;                              // 4x4 tiles per XCD super-tile
constexpr int H3_TILE_R = 256;                           // split-fp16 SYRK: 256 x 128 workgroup tile
constexpr int H3_TILE_C = 128;                           //   (4 waves as 2x2, each 128 x 64 = 4x2 MFMA 32x32 tiles)
#ifndef X1_CHS_SNPS
#define X1_CHS_SNPS 512
#endif
constexpr int X1_CHS = X1_CHS_SNPS;                       // ... SNPs per LDS table chunk (12-byte entries: 96 bytes per SNP)
constexpr int UV_CHS = 1024;                             // single-product SYRK (syrk_uv_kernel): SNPs per LDS table chunk (8-byte entries: 64 bytes per SNP)
constexpr int UV_SPARSE_MAC = 128;                       // ... SNPs with at most this many copies of the minor allele are added sparsely in fp64 (uv_sparse_kernel)
constexpr int X1_SPARSE_MAC = 128;                       // ... blocks WITH missing calls: up to this many copies (512: 8.2e-6 instead of 9.3e-6 on the
                                                         // rare-variant spectrum for +10 % of its step; SNPGPU_X1_SPARSE_MAC lowers it),
constexpr int X1_SPARSE_MIN_N = 384;                     // in contexts of at least this many samples (smaller data sets: bit-reproducible runs),
constexpr double X1_SPARSE_MIN_W = 64.0;                  // blocks WITH missing calls: ... and only where the weight 1 / (p (1 - p)) is at least this
constexpr int UV_CHUNK = 64;                             // ... slots per centre-balancing chunk (uv_tables_kernel)
constexpr int X1_TILE = 256;                             // single-wave-per-SIMD exact-row SYRK: 256 x 256 workgroup tile (4 waves of 128 x 128)
constexpr int H3_SUPER = 8;                              // 8 x 8 tiles per XCD super-tile: the 64 workgroups resident on an XCD share rows / columns (L2 word fetches -17 % against 4 x 4)
constexpr int H3_PROMOTE = 4096;                          // SNPs accumulated in fp32 before the fp64 flush (split-fp16 SYRK, three products)
// fp32 run lengths of the exact-row / single-product kernels (one launch and one fp64 flush per run; a flush = 5e9 fp64 atomics
// at N = 100 000: 7.7 ms, bound by the L2 atomic rate -- flat vs global address space, fp32 instead of fp64 operands and the
// order of the 256 instructions of a wave make no difference, tools/scratch A/B of round 4).  The accumulation error grows with
// sqrt(run): measured over ALL 3.7e8 entries of an 8192-row panel at configs[2]'s size 32 768-SNP runs put the maximum of the
// off-diagonal figure at 1.2e-5 (exact-row) / 1.6e-5 (single product, one weight target), 16 384 at 7e-6 / 1.2e-5 -- hence 8192 SNPs
// for the exact-row kernel.  The single-product kernel's runs also set its number of weight targets (kernels_prep.hip,
// uv_factor_kernel): a 32 768-SNP block = 3 runs of <= 11 264 slots, weight error 0.37e-6 rms, no refinement slots.
// SNPGPU_H3_PROMOTE overrides both; SNPGPU_SYRK_FAST=1 restores one 32 768-SNP run (one target).
constexpr int H3_PROMOTE_EXACT = 8192;          // (16 384: 29 of 3.7e8 entries above 1e-5, maximum 1.17e-5, on GCTA with 2 % missing calls)
constexpr int H3_PROMOTE_UV = 11264;            // slots per run of the single-product kernel (11 table chunks)
constexpr int H3_PROMOTE_FAST = 32768;
constexpr int UV_QMAX = 8;                      // runs of a block that may carry their own weight target (more runs: one target)
__host__ __device__ __forceinline__ double uv_run_factor(int q) { return 1.0 - (double)q * (1.0 / 4096.0); }   // flush factor of run q
constexpr int H3_HOMO_SHIFT = 8;                          // KING-homo tables are multiplied by 2^8 for the fp16 split
constexpr int H3_LUTCH = 512;                            // SNPs per LDS table chunk of the split-fp16 SYRK (2 x 32 KiB)
constexpr int I8_SUPER = 4;                              // int8-MFMA pair kernel: 4x4 tiles per XCD super-tile

void set_error(const std::string &msg);

#define SNPGPU_HIP_CHECK(expr)                                                             \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            ::snpgpu::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e)); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

// counter sets of the bit-plane pair kernel
enum PairMode { PM_IBS = 0, PM_KING_ROBUST = 1, PM_KING_HOMO = 2, PM_GCTA_MISS = 3, PM_BETA = 4,
                PM_IBS_NOMISS = 5 /* int8 kernel only: IBS / KING-robust for blocks without missing calls */,
                PM_HOMO_NOMISS = 6 /* ... KING-homo for such blocks: the same two products into its two planes */ };
constexpr int pair_mode_counters(int m) { return (m == PM_IBS || m == PM_BETA) ? 3 : m == PM_KING_ROBUST ? 5 : m == PM_KING_HOMO ? 2 : 1; }

// decode-table flavours of the SYRK kernel (what z(g) is)
enum LutMode {
    LUT_GCTA = 0,      // (g - 2p)/sqrt(p(1-p)), 0 unless 0<p<1        (genPCA.cpp:98-181)
    LUT_BAYES = 1,     // (g - 2p)/sqrt(s(1-s)), s=(sum+1)/(2num+2)    (genPCA.cpp:441-453)
    LUT_HOMO_W1 = 2,   // sqrt(p(1-p))          -> sum_mask p(1-p)      (genKING.cpp:236-248)
    LUT_HOMO_W2 = 3,   // p(1-p)                -> sum_mask (p(1-p))^2
    LUT_EIGMIX_NUM = 4,  // g - 2p (no scaling; missing -> 0)                 (genEIGMIX.cpp:104-110)
    LUT_EIGMIX_MISSW = 5 // sqrt(4p(1-p)) for MISSING calls, 0 otherwise -> weighted both-missing sums
};

struct TileGrid {      // upper-trapezoid tile enumeration with XCD super-tiles
    int tile_r, tile_c;        // tile extents in samples
    int super;                 // super-tile edge in tiles
    int n_tr, n_tc;            // tiles per panel (rows / cols)
    int n_sr;                  // super-tile rows
    int n_super;               // valid super-tiles
    int grid;                  // workgroups to launch
    int *d_prefix;             // [n_sr+1] prefix count of valid super-tiles per super-row
    int *d_first;              // [n_sr]   first valid super-col per super-row
};

// ---- launchers (defined in the .hip files) --------------------------------
int launch_sym_panel_matmul(hipStream_t st, const double *P, int64_t ld, int64_t tiles_c, int64_t nI, int64_t nJ, int64_t col0,
                            int64_t N, double scale, const double *Q, int m, double *Y, double *qt_scratch, bool fp32_products = false,
                            const float *P32 = nullptr);
int launch_panel_to_f32(hipStream_t st, const double *src, float *dst, size_t n_elems);
// PCA projections (kernels_proj.hip)
int launch_proj_snp(hipStream_t st, int corr, const uint32_t *w2, int64_t ncols_pad, int64_t N, int64_t n_snp,
                    const double *et, int kp, int k, const int32_t *sum, const int32_t *num, int bayesian, double *out,
                    double *part, int *cnt, double *out_avg, double *out_scale, const double *ext_avg = nullptr,
                    const double *ext_scale = nullptr);
int launch_proj_samp(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t N, int64_t n_snp, const double *sl,
                     int kp, int k, const double *af, const double *sc, double *out);
int launch_proj_transpose(hipStream_t st, const double *src, int64_t N, int k, double *dst, int64_t n_pad, int kp);
int launch_synth_block(hipStream_t st, uint8_t *dst, int64_t n_samp, int64_t snp_begin, int64_t n_snp, uint32_t seed,
                       uint32_t miss32, int spectrum, int special);
int launch_repack_stats(hipStream_t st, const void *src, int format, int64_t n_snp, int64_t n_samp, uint8_t *packed,
                        int64_t RB, int32_t *sum, int32_t *num, unsigned long long *d_missing);
int launch_repack(hipStream_t st, const void *src, int format, int64_t n_snp, int64_t n_samp,
                  uint8_t *packed, int64_t RB);
int launch_snp_stats(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t n_samp,
                     int32_t *sum, int32_t *num, unsigned long long *d_missing_cells, int32_t *nhet = nullptr);
int launch_build_lut(hipStream_t st, const int32_t *sum, const int32_t *num, int64_t n_snp, int64_t n_snp_pad,
                     int lut_mode, int split16, float2 *lut, unsigned long long *d_nlocus, double *d_sumden,
                     double *dvals, const unsigned long long *d_missing = nullptr, double2 *ccoef = nullptr,
                     int exact_rows_always = 0, int w_shift = 0, int exact_with_missing = 0, int entry12 = 0,
                     double *homo_const = nullptr, double4 *uvsp_miss = nullptr, int x1_sparse_mac = 0,
                     unsigned long long *d_short_runs = nullptr);
int launch_colcorr(hipStream_t st, const uint32_t *w8, int64_t ncols_pad, int n_d, const double2 *ccoef, double *tc,
                   double *colterm, const unsigned long long *d_missing, int always = 0, int entry12 = 0);
int launch_colterm_settle(hipStream_t st, double *acc, int64_t ld, int64_t tiles_c, int64_t n_rows_real, int64_t ncols_pad, int64_t n_cols_real,
                          double *colterm, double *uvterm = nullptr);
int launch_eigmix_samples(hipStream_t st, const uint32_t *w8, int n_d, int64_t ncols_pad, int64_t col0,
                          const double *dvals, uint32_t *het, double *dmiss, double *dsq,
                          const unsigned long long *d_wide16 = nullptr);
int launch_bitplanes4(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t n_samp,
                      int64_t col0, int64_t ncols_pad, int64_t rows_pad, int KW, uint4 *rowp, uint4 *colp);
int launch_bitplanes_miss(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t n_samp,
                          const int32_t *sum, const int32_t *num, int64_t col0, int64_t ncols_pad,
                          int64_t rows_pad, int KW, uint2 *rowp, uint2 *colp,
                          const unsigned long long *d_missing_cells);
int launch_pair_popcount(hipStream_t st, int mode, const TileGrid &tg, const void *rowp, const void *colp,
                         int KW, int64_t ncols_pad, uint32_t *acc, int64_t acc_plane,
                         const unsigned long long *d_skip_if_zero);
// int8-MFMA form of the pair counters (IBS / KING / beta): sample-major 2-bit words + kernel
int launch_transpose2(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t col0,
                      int64_t ncols_pad, int n_d, uint32_t *w2, uint32_t *het = nullptr,
                      const unsigned long long *d_missing = nullptr, bool classic = false);
int launch_transpose2_missmask(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t n_samp,
                               const int32_t *sum, const int32_t *num, int64_t col0, int64_t ncols_pad, int n_d,
                               uint32_t *w2, uint32_t *diag, const unsigned long long *d_skip_if_zero);
int launch_missmask256(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t N, const int32_t *sum,
                       const int32_t *num, int64_t col0, int n_groups, int64_t snp_stride, uint4 *mm,
                       const unsigned long long *d_missing, unsigned long long max_cells, unsigned long long *flags);
int launch_pair_sparse_miss(hipStream_t st, const uint4 *mm, int64_t snp_stride, int n_snp, uint32_t *acc, int64_t ncols_pad,
                            const int4 *work, int n_blocks, const unsigned long long *d_run);
int launch_transpose2_direct(hipStream_t st, const uint8_t *src, int64_t n_samp, int64_t n_snp, int64_t col0,
                             int64_t ncols_pad, int n_d, uint32_t *w2, uint32_t *het, uint32_t *het_blk,
                             unsigned long long *d_missing);
void pair_i8_tile(int mode, int *tile_r, int *tile_c, int *wg_per_cu = nullptr);
int launch_pair_fp4_miss(hipStream_t st, const int4 *work, int n_blocks, const uint32_t *w2, int64_t ncols_pad, int n_s,
                         uint32_t *acc, const unsigned long long *d_missing);
int launch_pair_i8(hipStream_t st, int mode, const int4 *work, int n_blocks, const uint32_t *w2, int64_t ncols_pad,
                   int n_q, int n_snp, uint32_t *acc, int64_t acc_plane, const unsigned long long *d_missing,
                   const int4 *work_nm = nullptr, int n_blocks_nm = 0, bool fp4_nomiss = false, bool fp4_general = false);
bool pair_fp4_tile(int mode, int *tile_r, int *tile_c, int *wg_per_cu);
int launch_het_settle(hipStream_t st, uint32_t *acc, int64_t plane, int64_t rows_pad, int64_t ncols_pad, uint32_t *het,
                      int king, int plane_ibs1 = 1, int plane_ibs0x2 = 2);
int launch_syrk_h3(hipStream_t st, const int4 *work, int n_blocks, const uint32_t *w8, int64_t ncols_pad,
                    const uint2 *lut, int n_q, double *acc, int64_t ld, int64_t tiles_c,
                    const unsigned long long *d_skip_if_zero = nullptr, int a_kind = -1, const unsigned long long *d_missing = nullptr, int64_t n_rows_real = 0,
                    int promote_snps = 0, const int4 *work_x1 = nullptr, int n_blocks_x1 = 0,
                    const unsigned long long *d_short_runs = nullptr);
int launch_syrk_uv(hipStream_t st, const int4 *work_x1, int n_blocks_x1, const uint32_t *w8, int64_t ncols_pad,
                   const uint2 *lut, int n_q, double *acc, int64_t ld, int64_t tiles_c, const unsigned long long *d_missing,
                   int64_t n_rows_real, int run_chunks, int n_target, int run_if_missing = 0, int64_t copy_lut_bytes = 0,
                   int64_t copy_acc_elems = 0, int uv16 = 0,      // uv16 = 2, 3: syrk_uv16c_kernel (lut = factor arrays)
                   const void *pace_src = nullptr, int pace = 0);   // ... and its pace-maker (on / off; 16 x 1 KiB per wave and chunk)
int launch_build_uv(hipStream_t st, const int32_t *sum, const int32_t *num, int64_t n_snp, int64_t n_snp_pad, int lut_mode,
                    uint2 *lut, double4 *uvcoef, double *kpart, double4 *uvsp, float *cand_err, uint32_t *cand_uv,
                    double2 *snp_tavg, int32_t *slot_of, int32_t *slot_src, int n_target, int cpr,
                    const unsigned long long *d_missing, int swap_odd = 0);
int launch_uv_sparse(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t N, int64_t row0, int64_t row1,
                     int64_t col0, const double4 *uvsp, double *acc, int64_t ld, int64_t tiles_c, int64_t ncols_pad, double *uvterm,
                     const unsigned long long *d_missing, int missing_blocks = 0);
int launch_uvcorr(hipStream_t st, const uint32_t *w8, int64_t ncols_pad, int n_d, const double4 *uvcoef, const double *kpart,
                  int n_kpart, double2 *tc, double *uvterm, const unsigned long long *d_missing, int nibble = 0);
int launch_homo_uv(hipStream_t st, const int32_t *sum, const int32_t *num, int64_t n_snp, int64_t n_snp_pad, uint2 *lut1, uint2 *lut2,
                   double2 *wts, double *totals, const uint32_t *w8, int64_t ncols_pad, double2 *tc, double *msum,
                   const unsigned long long *d_missing, int swap_odd = 0);
int launch_transpose8(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t col0,
                      int64_t ncols_pad, int n_d, uint32_t *w8, const unsigned long long *d_wide16 = nullptr,
                      int always_wide = 0, const int32_t *slot_src = nullptr, int nibble_nomiss = 0);
int launch_syrk(hipStream_t st, const TileGrid &tg, const uint32_t *w8, int64_t ncols_pad, const float2 *lut,
                int n_q, double *acc, int64_t ld, int64_t tiles_c, const unsigned long long *d_skip_if_zero = nullptr);

// finalisers: panel accumulators -> caller layout (device buffers)
struct PanelGeom {
    int64_t N, row0, row1, col0, rows_pad, ncols_pad;
    int64_t f64_tiles_c;       // fp64 planes: 0 = row-major [rows_pad][ncols_pad], else tile-major with this many 256-column tiles per row
};
int launch_fin_ibs_num(hipStream_t st, const PanelGeom &g, const uint32_t *acc, int32_t *o0, int32_t *o1,
                       int32_t *o2, int packed);
int launch_fin_ibs_ave(hipStream_t st, const PanelGeom &g, const uint32_t *acc, double *out, int packed);
int launch_fin_king_counts(hipStream_t st, const PanelGeom &g, const uint32_t *acc, uint32_t *out5);
int launch_fin_king_robust(hipStream_t st, const PanelGeom &g, const uint32_t *acc, const int32_t *family,
                           double *ibs0, double *kin, int packed);
int launch_fin_king_homo(hipStream_t st, const PanelGeom &g, const uint32_t *acc, const double *facc, double fscale,
                         double *k0, double *k1, int packed, const double *w_const = nullptr, const double *msum = nullptr);
int launch_fin_gcta(hipStream_t st, const PanelGeom &g, const double *num, const uint32_t *miss,
                    const uint32_t *diag, const unsigned long long *d_nlocus, double *out, int packed,
                    const double *colterm = nullptr, const double *uvterm = nullptr);
int launch_miss_diag(hipStream_t st, const uint2 *colp, int KWv, int64_t ncols_pad, int64_t col0, uint32_t *diag,
                     const unsigned long long *skip);
int launch_fin_cov(hipStream_t st, const PanelGeom &g, const double *num, double scale, double *out, int packed);
int launch_trace(hipStream_t st, const PanelGeom &g, const double *num, double *d_trace);
int launch_fin_mom(hipStream_t st, const PanelGeom &g, const uint32_t *acc, const double *e, int constraint,
                   double *k0, double *k1, int packed);
int launch_fin_eigmix(hipStream_t st, const PanelGeom &g, const double *num, const double *dd, const uint32_t *het,
                      const double *dmiss, const double *dsq, const double *d_sumden, int diagadj, double scale, double *out, int packed);
int launch_beta_reduce(hipStream_t st, const PanelGeom &g, const uint32_t *acc, int diag_inbreeding, double *partial_min,
                       double *partial_sum, int nblocks);
int launch_fin_beta(hipStream_t st, const PanelGeom &g, const uint32_t *acc, int mode, double avg, double mn, double *out,
                    int packed);
int launch_mirror_diag(hipStream_t st, const PanelGeom &g, double *num);
int launch_mirror_diag_tiles(hipStream_t st, const PanelGeom &g, double *num, int T);

// context-level pieces shared between api.hip and eigen.hip / multi.hip
int ctx_settle(snpgpu_ctx *c);                                   // pending column / row terms of the fp16 SYRK -> panel
int ctx_panel_matmul_enqueue(snpgpu_ctx *c, double scale, const double *Q, int m, double *Y, bool fp32_products = false);   // no host synchronisation

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int alloc(size_t n)
    {
        if (n == 0) n = 16;
        SNPGPU_HIP_CHECK(hipMalloc(&p, n));
        bytes = n;
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};


}  // namespace snpgpu

struct snpgpu_proj {       // PCA projector (proj.hip)
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t N = 0, RB = 0, ncols_pad = 0, Bmax = 0, n_pad = 0;
    int k = 0, kp = 0;
    bool have_eig = false;
    int64_t staged_snps = 0;   // SNPs of the block currently held in `packed` (0: none)
    bool staged_words = false; // ... and whether w2 holds its sample-major words
    snpgpu::DevBuf raw, packed, sum, num, w2, et, eig_in, out, part, cnt, avg, scale, sl, af, sc, acc, flag;
};

struct snpgpu_ctx {
    int kind = 0, device = 0, bayesian = 0;
    int64_t N = 0, row0 = 0, row1 = 0, col0 = 0;
    int64_t rows_pad = 0, ncols_pad = 0, RB = 0, Bmax = 0;
    int KWmax = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool full = false;
    int64_t n_snp_total = 0;
    // asynchronous host feeds: second stream + double-buffered raw block + events
    hipStream_t copy_stream = nullptr;
    snpgpu::DevBuf raw2[2];
    hipEvent_t ev_copied[2] = {nullptr, nullptr};    // H2D of raw2[k] finished
    hipEvent_t ev_consumed[2] = {nullptr, nullptr};  // repack of raw2[k] finished (buffer reusable)
    const void *host_src[2] = {nullptr, nullptr};
    int raw_turn = 0;
    int diag_mirrored = 0;        // eigen solver: 1 = diagonal 64 x 64 tiles mirrored, 2 = whole diagonal square
    bool frozen = false;          // snpgpu_finalize_inplace: plane 0 of acc_f64 holds the FINAL matrix (upper trapezoid of the
    int frozen_diagadj = 0;       //   panel rectangle); no feeds may follow, the kind's finaliser copies it out
    double frozen_scale = 1.0;
    void *blas = nullptr;         // rocblas_handle, created on first use
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[2];  // [0] pair popcount, [1] SYRK

    // feed-block scratch
    snpgpu::DevBuf raw, packed, sum, num, nhet, lut[2], rowp, colp, wt, w2, scalars, family, miss_diag, dvals, samp_het, samp_dmiss, samp_dsq;
    snpgpu::DevBuf eig_qt;        // eigen solver: sample-major copy of the current vector block, double [N][48]
    snpgpu::DevBuf acc_f32;       // eigen solver: fp32 copy of plane 0 for the fp32 products (made on first use where memory allows)
    bool acc_f32_valid = false;   //     ... and whether it still mirrors the plane (a feed invalidates it)
    snpgpu::DevBuf het_blk;       // per-block het counts of the one-pass pre-pass (committed to `het` when the block's flag is final)
    snpgpu::DevBuf het, i8_work_nm;   // binary pair kernel for blocks without missing calls: per-sample het counts, its work list
    int i8_blocks_nm = 0;
    bool het_pending = false;
    snpgpu::DevBuf ccoef, tcorr;   // exact-row-side SYRK: per-SNP {u, v} and per-chunk column terms [Bmax / H3_LUTCH + 1][ncols_pad]
    snpgpu::DevBuf colterm;        // ... their running total per column (fp64 [ncols_pad]), subtracted from every row of the
    snpgpu::DevBuf wt12;           // EIGMIX: 12 * code words of a block with missing calls (exact-row kernel of the numerator)
    bool eigmix_x1 = false;        // EIGMIX numerator of blocks with missing calls on syrk_x1_kernel (else the legacy three-product kernel)
    snpgpu::DevBuf uvpace;         // syrk_uv16c_kernel: 64 KiB per table chunk that every workgroup fetches (zeros; see the kernel)
    int uvc_pace = 0;              // ... pace-maker on / off (SNPGPU_UVC_PACE, default on)
    snpgpu::DevBuf uvlut, uvslot;  // ... its own tables (8-byte entries, per SLOT) and the slot -> SNP map of the current block
    snpgpu::DevBuf uvcand;         // ... per SNP and weight target: {relative error, u | v << 16}, {t, avg}, SNP -> slot
    int uv_promote = 0;            // fp32 run of the single-product kernel in slots (h3_promote: of the exact-row kernel, in SNPs)
    snpgpu::DevBuf uvcoef, uvterm, uvkpart, uvsp;   // single-product SYRK (blocks without missing calls): per-SNP {d_b uv, c_a, d_a uv, c_b},
                                   //     the running row / column terms {R[ncols_pad], Q[ncols_pad], K} and per-chunk parts of K
    bool uv_enabled = false;
    bool uv16 = false;            // single-product kernel on v_mfma_f32_16x16x32_f16 (syrk_uv16_kernel; tables with swapped odd quarters)
    bool uvc_carry = false;       // ... syrk_uv16c_kernel walks a tile's runs itself, half the sub-tile sums carried in LDS as fp32 (SNPGPU_SYRK_UV16=3)
    bool uvc = false;             // ... with the operands CONVERTED from nibble words instead of looked up (syrk_uv16c_kernel: `uvlut` holds the
                                  // slots' factors, `wt` bytes c0 | c1 << 4 in blocks without missing calls); GRM / PCA contexts only
    bool uv_targets = false;       // a weight target per fp32 run (uv_factor_kernel)
    int x1_sparse_mac = 0;
    bool x1_short_runs = true;     // blocks with rare variants on the sparse path AND missing calls: half-length fp32 runs (device flag)
    bool sparse_missing = false;            // rare variants of blocks with missing calls: carriers' pairs added in fp64 (uv_sparse_kernel)
    bool uv_eigmix = false;      // ... for the EIGMIX numerator (weight 1: exact)
    bool homo_uv = false;        // KING-homo blocks with missing calls: weight sums = totals - per-sample missing sums + ONE fp16 product each
    snpgpu::DevBuf homo_lut[2], homo_wts, homo_tc, homo_msum, homo_work;   //     ... its tables, effective weights, per-chunk partials, M[2][ncols_pad], work list
    int homo_blocks = 0;
    bool colterm_pending = false;  //     panel once, before a result is read (settle_colterm, api.hip)
    // accumulators
    snpgpu::DevBuf acc_u32, acc_f64;
    int n_u32 = 0, n_f64 = 0;
    snpgpu::TileGrid tg_pc{}, tg_mm{};
    snpgpu::DevBuf tg_pc_tab, tg_mm_tab;
    bool use_pc = false, use_mm = false;
    bool pc_i8 = false;        // pair counters on int8 MFMA (w2 words) instead of bit planes
    bool miss_fp4 = true;      // GCTA both-missing counts on the MX-fp4 MFMA (SNPGPU_GCTA_MISS_FP4=0: the int8 kernel)
    bool nomiss_fp4 = true;    // IBS / KING blocks without missing calls on the MX-fp4 MFMA (SNPGPU_PAIR_FP4=0: the int8 two-product kernel)
    bool general_fp4 = false;  // ... and the general IBS / KING-robust kernels (blocks with missing calls; SNPGPU_PAIR_FP4_GENERAL=0: int8)
    int i8_blocks = 0;         // work items (= workgroups) of the int8 pair kernel, see build_worklist
    snpgpu::DevBuf i8_work;    // int4 {tile row, tile col, K part, K parts} per workgroup, XCD-interleaved
    snpgpu::DevBuf mm256, sp_work;   // GCTA denominators, sparse form: per (256-sample group, SNP) set of missing calls; its 256 x 256 work list
    int sp_blocks = 0;
    double sp_max_rate = 0.0;        // ... taken for blocks whose missing-call rate is at most this (0: never)
    bool mm_h3 = false;        // SYRK on split-fp16 MFMAs (GCTA / Bayesian tables) instead of fp32 MFMAs
    bool h3_exact_rows = false; // two-product kernel with the exact row operand (g - c_s) 2^shift
    bool h3_exact_missing = false; // ... also for blocks WITH missing calls (row value of a missing call = fp16(avg - c_s)); else three products there
    int h3_promote = 0;         // fp32 run length of the exact-row kernel in SNPs (0 = H3_PROMOTE_EXACT; SNPGPU_H3_PROMOTE)
    int h3_a_kind[2] = {-1, -1};
    int h3_w_shift = 0;         // exact-row tables hold w * 2^-shift, the row operand is +-2^shift (fp16 range, |w| <= 4N)
    int h3_blocks = 0;
    snpgpu::DevBuf h3_work;
    int x1_blocks = 0;          // work list of syrk_x1_kernel (256 x 256 tiles, one workgroup per CU); 0: not used
    snpgpu::DevBuf x1_work;
    int pc_mode = 0;
    int lut_mode[2] = {0, 0};
    int n_lut = 0;

    // scalars layout: SCALAR_SLOTS slots of 8 bytes (unsigned long long / double) -- the allocation in snpgpu_create is exactly
    // this many, a ninth scalar needs SCALAR_SLOTS raised with it:
    // [0] missing cells of the current block, [1] nLocus, [2] trace (double), [3] EIGMIX SumDenominator (double),
    // [4..5] KING-homo weight sums of the blocks without missing calls, [6..7] route of this block's both-missing counts
    // [8] this block holds rare variants on the fp64 sparse path next to missing calls: the exact-row kernel runs it as 4096-SNP fp32 runs
    static constexpr int SCALAR_SLOTS = 16;
    unsigned long long *d_missing() { return (unsigned long long *)scalars.p; }
    unsigned long long *d_nlocus() { return (unsigned long long *)scalars.p + 1; }
    double *d_trace() { return (double *)scalars.p + 2; }
    double *d_sumden() { return (double *)scalars.p + 3; }
    double *d_homo_w() { return (double *)scalars.p + 4; }   // [2]: sum p(1-p), sum (p(1-p))^2 over the blocks without missing calls (KING-homo)
    unsigned long long *d_short_runs() { return (unsigned long long *)scalars.p + 8; }
    unsigned long long *d_miss_route() { return (unsigned long long *)scalars.p + 6; }   // [2]: this block's both-missing counts take the sparse / the dense form

    int64_t acc_tiles_c = 0;     // fp64 planes tile-major: ncols_pad / 256 (0 = row-major)
    snpgpu::PanelGeom geom() const { return snpgpu::PanelGeom{N, row0, row1, col0, rows_pad, ncols_pad, acc_tiles_c}; }
    int64_t plane() const { return rows_pad * ncols_pad; }
};

