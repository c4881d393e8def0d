// C ABI of libsnpgpu, level (1): streaming accumulator contexts (include/snpgpu.h).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include <rocblas/rocblas.h>

#include "snpgpu_internal.h"

namespace snpgpu {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

}  // namespace snpgpu

using namespace snpgpu;

static int build_tile_grid(snpgpu_ctx *c, TileGrid &tg, DevBuf &tab, int tile_r, int tile_c, int S)
{
    tg.tile_r = tile_r; tg.tile_c = tile_c; tg.super = S;
    tg.n_tr = (int)((c->row1 - c->row0 + tile_r - 1) / tile_r);
    tg.n_tc = (int)((c->N - c->col0 + tile_c - 1) / tile_c);
    tg.n_sr = (tg.n_tr + S - 1) / S;
    const int n_sc = (tg.n_tc + S - 1) / S;
    std::vector<int> prefix(tg.n_sr + 1, 0), first(tg.n_sr, 0);
    for (int sr = 0; sr < tg.n_sr; sr++) {
        // first super-column whose last sample column reaches the first row of this super-row
        const int64_t row_lo = (int64_t)sr * S * tile_r;
        int f = (int)(row_lo / ((int64_t)S * tile_c));
        if (f > n_sc) f = n_sc;
        first[sr] = f;
        prefix[sr + 1] = prefix[sr] + (n_sc - f);
    }
    tg.n_super = prefix[tg.n_sr];
    tg.grid = 8 * S * S * ((tg.n_super + 7) / 8);
    if (tab.alloc(sizeof(int) * (size_t)(2 * tg.n_sr + 2))) return 1;
    tg.d_prefix = (int *)tab.p;
    tg.d_first = tg.d_prefix + tg.n_sr + 1;
    SNPGPU_HIP_CHECK(hipMemcpy(tg.d_prefix, prefix.data(), sizeof(int) * prefix.size(), hipMemcpyHostToDevice));
    SNPGPU_HIP_CHECK(hipMemcpy(tg.d_first, first.data(), sizeof(int) * first.size(), hipMemcpyHostToDevice));
    return 0;
}

// Work list of the int8 pair kernel and of the split-fp16 SYRK.  Tiles touching the upper trapezoid of the panel are grouped in
// S x S super-tiles; super-tile k goes to XCD k % 8 (workgroup b runs on XCD b % 8), so the tiles that are
// resident together on one XCD share operand rows/columns in its L2.  The chip runs `slots` workgroups
// at a time; the tiles of the last, partially filled round are split along K into the number of parts
// that makes that round shortest (their counters are flushed with atomics, so parts may share a tile).
// copies > 1: every tile appears `copies` times (copy index in bits 16.. of the item's fourth field): several independent
// products over the same tiles in ONE launch (KING-homo's two weight sums), balanced together
static int build_worklist(snpgpu_ctx *c, int tile_r, int tile_c, int S, DevBuf &buf, int &n_blocks, int wg_per_cu = 2, int copies = 1)
{
    const int n_tr = (int)((c->row1 - c->row0 + tile_r - 1) / tile_r);
    const int n_tc = (int)((c->N - c->col0 + tile_c - 1) / tile_c);
    const int n_sr = (n_tr + S - 1) / S, n_sc = (n_tc + S - 1) / S;
    // (tile row, tile column | copy << 20): the copy index travels in the column field until the items are written
    std::vector<std::vector<std::pair<int, int>>> queue(8);
    int k = 0;
    for (int sr = 0; sr < n_sr; sr++)
        for (int sc = 0; sc < n_sc; sc++) {
            std::vector<std::pair<int, int>> tiles;
            for (int a = 0; a < S; a++)
                for (int b = 0; b < S; b++) {
                    const int tr = sr * S + a, tc = sc * S + b;
                    if (tr < n_tr && tc < n_tc && (int64_t)(tc + 1) * tile_c > (int64_t)tr * tile_r)
                        for (int cp = 0; cp < copies; cp++) tiles.push_back({tr, tc | (cp << 20)});
                }
            if (tiles.empty()) continue;
            auto &q = queue[k++ & 7];
            q.insert(q.end(), tiles.begin(), tiles.end());
        }
    // even out the queues (diagonal super-tiles are smaller): move tiles from the longest to the shortest
    for (;;) {
        int lo = 0, hi = 0;
        for (int x = 1; x < 8; x++) {
            if (queue[x].size() < queue[lo].size()) lo = x;
            if (queue[x].size() > queue[hi].size()) hi = x;
        }
        if (queue[hi].size() <= queue[lo].size() + 1) break;
        queue[lo].push_back(queue[hi].back());
        queue[hi].pop_back();
    }
    int64_t T = 0;
    for (auto &q : queue) T += (int64_t)q.size();
    int ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    const int64_t slots = (int64_t)wg_per_cu * ncu;  // 256 threads x <= 256 VGPRs: two workgroups per CU (one with <= 512)
    const int64_t rem = T % slots;
    int parts = 1;
    if (rem > 0) {
        double best = 1.0;                           // duration of the last round in units of a whole tile
        for (int p = 2; p <= 8; p++) {
            const double d = (double)((rem * p + slots - 1) / slots) / p + 0.02 * (p - 1);   // + flush overhead
            if (d < best - 1e-9) { best = d; parts = p; }
        }
    }
    const char *force = getenv("SNPGPU_I8_TAIL_PARTS");
    if (force && atoi(force) >= 1 && atoi(force) <= 64) parts = atoi(force);
    // split the last `rem` tiles, rem/8 from the end of every queue
    std::vector<int4> work;
    size_t longest = 0;
    std::vector<std::vector<int4>> items(8);
    for (int x = 0; x < 8; x++) {
        const auto &q = queue[x];
        const size_t n_split = (parts > 1) ? std::min(q.size(), (size_t)((rem + 7 - x) / 8)) : 0;
        for (size_t i = 0; i < q.size() - n_split; i++)
            items[x].push_back(make_int4(q[i].first, q[i].second & 0xFFFFF, 0, 1 | ((q[i].second >> 20) << 16)));
        for (int p = 0; p < parts; p++)
            for (size_t i = q.size() - n_split; i < q.size(); i++)
                items[x].push_back(make_int4(q[i].first, q[i].second & 0xFFFFF, p, parts | ((q[i].second >> 20) << 16)));
        longest = std::max(longest, items[x].size());
    }
    work.assign(longest * 8, make_int4(0, 0, 0, 0));
    for (int x = 0; x < 8; x++)
        for (size_t i = 0; i < items[x].size(); i++) work[i * 8 + x] = items[x][i];
    n_blocks = (int)work.size();
    if (work.empty()) return 0;
    if (buf.alloc(sizeof(int4) * work.size())) return 1;
    SNPGPU_HIP_CHECK(hipMemcpy(buf.p, work.data(), sizeof(int4) * work.size(), hipMemcpyHostToDevice));
    return 0;
}

static void free_ctx(snpgpu_ctx *c)
{
    (void)hipSetDevice(c->device);
    DevBuf *all[] = {&c->raw, &c->packed, &c->sum, &c->num, &c->lut[0], &c->lut[1], &c->rowp, &c->colp, &c->wt, &c->w2,
                     &c->scalars, &c->family, &c->miss_diag, &c->nhet, &c->dvals, &c->samp_het, &c->samp_dmiss, &c->samp_dsq, &c->acc_u32, &c->acc_f64, &c->i8_work, &c->mm256, &c->sp_work, &c->h3_work, &c->x1_work, &c->eig_qt, &c->acc_f32, &c->ccoef, &c->tcorr, &c->colterm, &c->uvcoef, &c->uvterm, &c->uvkpart, &c->uvsp, &c->uvlut, &c->uvpace, &c->uvslot, &c->uvcand, &c->homo_lut[0], &c->homo_lut[1], &c->homo_wts, &c->homo_tc, &c->homo_msum, &c->homo_work, &c->wt12, &c->het, &c->het_blk, &c->i8_work_nm, &c->tg_pc_tab,
                     &c->tg_mm_tab};
    for (DevBuf *b : all) b->release();
    for (int k = 0; k < 2; k++) {
        c->raw2[k].release();
        if (c->ev_copied[k]) (void)hipEventDestroy(c->ev_copied[k]);
        if (c->ev_consumed[k]) (void)hipEventDestroy(c->ev_consumed[k]);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->blas) (void)rocblas_destroy_handle((rocblas_handle)c->blas);
    for (int w = 0; w < 2; w++)
        for (auto &p : c->ev[w]) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" {

int snpgpu_abi_version(void) { return SNPGPU_ABI_VERSION; }
const char *snpgpu_last_error(void) { return g_err.c_str(); }

int snpgpu_device_count(int *count)
{
    int n = 0;
    SNPGPU_HIP_CHECK(hipGetDeviceCount(&n));
    if (count) *count = n;
    return 0;
}

int snpgpu_synth_block(void *dst, int64_t n_samp, int64_t snp_begin, int64_t n_snp, uint32_t seed, double missing,
                       int spectrum, int special, int device, void *stream)
{
    if (!dst || n_samp <= 0 || n_snp < 0 || snp_begin < 0 || !(missing >= 0.0 && missing < 1.0) || spectrum < 0 || spectrum > 4) {
        set_error("snpgpu_synth_block: invalid arguments");
        return 1;
    }
    SNPGPU_HIP_CHECK(hipSetDevice(device));
    const uint32_t miss32 = (uint32_t)std::floor(missing * 4294967296.0);
    // without a stream the block is written on the NULL stream, which does not order itself against the contexts' own
    // (non-blocking) streams: wait for whatever may still read `dst` (an earlier asynchronous snpgpu_feed of the same buffer)
    if (!stream) SNPGPU_HIP_CHECK(hipDeviceSynchronize());
    if (launch_synth_block((hipStream_t)stream, (uint8_t *)dst, n_samp, snp_begin, n_snp, seed, miss32, spectrum, special)) return 1;
    if (!stream) SNPGPU_HIP_CHECK(hipDeviceSynchronize());
    return 0;
}

int snpgpu_create(int kind, int64_t n_samp, const snpgpu_opts *opts, snpgpu_ctx **out)
{
    if (!out) { set_error("snpgpu_create: out is NULL"); return 1; }
    *out = nullptr;
    if (kind < SNPGPU_IBS || kind > SNPGPU_INDIV_BETA) { set_error("snpgpu_create: invalid kind"); return 1; }
    if (n_samp <= 0 || n_samp > 0x7fffffffLL) { set_error("snpgpu_create: invalid number of samples"); return 1; }
    snpgpu_opts o{};
    if (opts) o = *opts;
    int ndev = 0;
    SNPGPU_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { set_error("snpgpu_create: no HIP device (the GPU path has no CPU fallback)"); return 1; }
    if (o.device < 0 || o.device >= ndev) { set_error("snpgpu_create: invalid device ordinal"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(o.device));
    {
        // the kernels are written for gfx950 (MI355X) and nothing else: MX-fp4 matrix instructions, 160 KiB of LDS per workgroup,
        // 512 registers per lane.  A device of another architecture could not load the code objects; say so here rather than at
        // the first launch.
        hipDeviceProp_t prop;
        SNPGPU_HIP_CHECK(hipGetDeviceProperties(&prop, o.device));
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0 || prop.sharedMemPerBlock < 160 * 1024) {
            set_error(std::string("snpgpu_create: device ") + std::to_string(o.device) + " is " + prop.gcnArchName +
                      "; libsnpgpu is built for gfx950 (MI355X) only");
            return 1;
        }
    }

    snpgpu_ctx *c = new snpgpu_ctx();
    c->kind = kind; c->device = o.device; c->bayesian = o.bayesian; c->N = n_samp;
    c->row0 = o.row_begin; c->row1 = o.row_end;
    if (c->row0 == 0 && c->row1 == 0) c->row1 = n_samp;
    if (c->row0 < 0 || c->row1 > n_samp || c->row0 >= c->row1 || (c->row0 % PANEL_ALIGN) != 0) {
        set_error("snpgpu_create: invalid panel rows (row_begin must be a multiple of 256, < row_end <= n_samp)");
        delete c;
        return 1;
    }
    c->full = (c->row0 == 0 && c->row1 == n_samp);
    c->col0 = c->row0;
    c->rows_pad = round_up(c->row1 - c->row0, PANEL_ALIGN);
    c->ncols_pad = round_up(c->N - c->col0, PANEL_ALIGN);
    c->RB = round_up(c->N, 256) / 4;
    c->Bmax = o.max_block_snps > 0 ? o.max_block_snps : 32768;       // the block bench.py feeds for GRM / PCA
    c->Bmax = round_up(c->Bmax, 64);
    c->KWmax = (int)(c->Bmax / 32);
    // fp64 planes tile-major (snpgpu_internal.h: acc_off); row-major with SNPGPU_ACC_LAYOUT=row and for the rocBLAS form of
    // the eigen solver's panel product (SNPGPU_EIG_BLAS=1), which needs a leading dimension
    {
        const char *lay = getenv("SNPGPU_ACC_LAYOUT");
        c->acc_tiles_c = ((lay && std::string(lay) == "row") || getenv("SNPGPU_EIG_BLAS")) ? 0 : c->ncols_pad / ACC_TILE;
    }
    if (o.stream) {
        c->stream = (hipStream_t)o.stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            set_error("snpgpu_create: hipStreamCreate failed");
            delete c;
            return 1;
        }
        c->own_stream = true;
    }

    switch (kind) {
    case SNPGPU_IBS: c->use_pc = true; c->pc_mode = PM_IBS; break;
    case SNPGPU_KING_ROBUST: c->use_pc = true; c->pc_mode = PM_KING_ROBUST; break;
    case SNPGPU_KING_HOMO:
        c->use_pc = true; c->pc_mode = PM_KING_HOMO;
        c->use_mm = true; c->n_lut = 2; c->lut_mode[0] = LUT_HOMO_W1; c->lut_mode[1] = LUT_HOMO_W2;
        break;
    case SNPGPU_GRM_GCTA:
        c->use_pc = true; c->pc_mode = PM_GCTA_MISS;
        c->use_mm = true; c->n_lut = 1; c->lut_mode[0] = LUT_GCTA;
        break;
    case SNPGPU_PCA_COV:
        c->use_mm = true; c->n_lut = 1; c->lut_mode[0] = o.bayesian ? LUT_BAYES : LUT_GCTA;
        break;
    case SNPGPU_EIGMIX:
        c->use_mm = true; c->n_lut = 2; c->lut_mode[0] = LUT_EIGMIX_NUM; c->lut_mode[1] = LUT_EIGMIX_MISSW;
        break;
    case SNPGPU_INDIV_BETA: c->use_pc = true; c->pc_mode = PM_BETA; break;
    }
    c->n_u32 = c->use_pc ? pair_mode_counters(c->pc_mode) : 0;
    c->n_f64 = c->n_lut;

    int rc = 0;
    const size_t plane = (size_t)c->plane();
    rc |= c->packed.alloc((size_t)c->Bmax * (size_t)c->RB);
    rc |= c->sum.alloc(sizeof(int32_t) * (size_t)c->Bmax);
    rc |= c->num.alloc(sizeof(int32_t) * (size_t)c->Bmax);
    if (kind == SNPGPU_EIGMIX) {
        rc |= c->dvals.alloc(sizeof(double) * 2 * (size_t)c->Bmax);
        rc |= c->samp_dsq.alloc(sizeof(double) * (size_t)c->RB * 4);
        rc |= c->samp_het.alloc(sizeof(uint32_t) * (size_t)c->RB * 4);
        rc |= c->samp_dmiss.alloc(sizeof(double) * (size_t)c->RB * 4);
    }
    rc |= c->scalars.alloc(8 * snpgpu_ctx::SCALAR_SLOTS);
    // 16 entries per SNP pair, whole chunks; x 2: the exact-row tables of blocks without missing calls have 16-byte entries
    for (int i = 0; i < c->n_lut && !rc; i++) rc |= c->lut[i].alloc(sizeof(float2) * 8 * 2 * (size_t)(c->Bmax + 2048));
    if (c->use_pc && !rc) {
        // IBS / KING / beta counters: exact int8 MFMA contractions by default; SNPGPU_PAIR_BACKEND=popcount
        // selects the bit-plane kernel (same counters, kept for comparison and for the GCTA missing mask)
        const char *be = getenv("SNPGPU_PAIR_BACKEND");
        c->pc_i8 = !(be && std::string(be) == "popcount");
        rc |= c->acc_u32.alloc(sizeof(uint32_t) * plane * (size_t)c->n_u32);
        if (c->pc_i8) {
            int tr = 0, tc = 0, wpc0 = 2;
            pair_i8_tile(c->pc_mode, &tr, &tc, &wpc0);
            if (const char *e = getenv("SNPGPU_PAIR_FP4")) c->nomiss_fp4 = atoi(e) != 0;
            {
                const char *e = getenv("SNPGPU_PAIR_FP4_GENERAL");
                int ftr = 0, ftc = 0, fw = 1;
                if (c->nomiss_fp4 && !(e && !atoi(e)) && pair_fp4_tile(c->pc_mode, &ftr, &ftc, &fw)) {
                    c->general_fp4 = true; tr = ftr; tc = ftc; wpc0 = fw;
                }
            }
            rc |= c->w2.alloc(sizeof(uint32_t) * (size_t)(c->Bmax / 16 + 32) * (size_t)c->ncols_pad);  // padding to 128 (fp4 product: 256) SNPs + 4 k-steps of read-ahead
            if (const char *e = getenv("SNPGPU_GCTA_MISS_FP4")) c->miss_fp4 = atoi(e) != 0;
            if (!rc) rc |= build_worklist(c, tr, tc, I8_SUPER, c->i8_work, c->i8_blocks, wpc0);
            // blocks without missing calls: binary 3-product kernel (IBS and KING-robust), 128 x 128 tiles
            if (!rc && (c->pc_mode == PM_IBS || c->pc_mode == PM_KING_ROBUST || c->pc_mode == PM_KING_HOMO) &&
                !getenv("SNPGPU_I8_NO_NOMISS")) {
                // per sample: #het, then #(g == 2), over the blocks the two-product kernel took
                rc |= c->het.alloc(sizeof(uint32_t) * (size_t)(2 * c->ncols_pad));
                if (!rc) rc |= (hipMemset(c->het.p, 0, sizeof(uint32_t) * (size_t)(2 * c->ncols_pad)) != hipSuccess);
                if (!rc) rc |= c->het_blk.alloc(sizeof(uint32_t) * (size_t)(2 * c->ncols_pad));
                if (!rc) rc |= (hipMemset(c->het_blk.p, 0, sizeof(uint32_t) * (size_t)(2 * c->ncols_pad)) != hipSuccess);
                int nr = 0, nc = 0, wpc = 2;
                pair_i8_tile(PM_IBS_NOMISS, &nr, &nc, &wpc);
                if (!rc) rc |= build_worklist(c, nr, nc, I8_SUPER, c->i8_work_nm, c->i8_blocks_nm, wpc);
            }
        } else {
            const size_t pv = (c->pc_mode == PM_GCTA_MISS) ? 4 : 16;  // bytes per (sample, 32-SNP word)
            rc |= c->rowp.alloc(pv * (size_t)c->rows_pad * (size_t)c->KWmax);
            rc |= c->colp.alloc(pv * (size_t)c->ncols_pad * (size_t)c->KWmax);
            if (!rc) rc |= build_tile_grid(c, c->tg_pc, c->tg_pc_tab, PC_TILE_R, PC_TILE_C, PC_SUPER);
        }
        if (c->pc_mode == PM_GCTA_MISS && !rc) rc |= c->miss_diag.alloc(sizeof(uint32_t) * (size_t)c->RB * 4);
        // GCTA denominators: blocks with FEW missing calls count the both-missing pairs from SETS of samples (pair_sparse_miss_kernel,
        // work ~ f^2) instead of the dense int8 product (81 ms per 32 768-SNP block at N = 100 000 whatever f).  Measured at
        // N = 100 000 (A/B on one box, bench.py --missing f): f = 0.2 %: 42 ms, 0.5 %: 75 ms, 1 %: 136 ms, 2 %: 285 ms -- the
        // per-thread nested walk over two 256-bit sets diverges badly and the sets themselves are 160 GB of L2 reads per block; the
        // sparse form is therefore taken up to 0.2 % missing calls in a block (well-called array / sequence data; 0.3 % before the dense
        // product moved to the fp4 instruction: 44 ms per 32 768 SNPs), the dense
        // product beyond.  SNPGPU_GCTA_SPARSE=0: always dense; SNPGPU_GCTA_SPARSE_MAX_RATE overrides the threshold (tests: 0.03)
        if (c->pc_mode == PM_GCTA_MISS && c->pc_i8 && !rc && !(getenv("SNPGPU_GCTA_SPARSE") && !atoi(getenv("SNPGPU_GCTA_SPARSE")))) {
            c->sp_max_rate = 0.002;       // (0.003 while the dense product was the int8 kernel: 81 ms per 32 768 SNPs; fp4: 44)
            if (const char *e = getenv("SNPGPU_GCTA_SPARSE_MAX_RATE")) { const double v = atof(e); if (v >= 0 && v <= 1) c->sp_max_rate = v; }
            rc |= c->mm256.alloc(32 * (size_t)(c->ncols_pad / 256) * (size_t)round_up(c->Bmax, 256));
            if (!rc) rc |= build_worklist(c, 256, 256, 4, c->sp_work, c->sp_blocks, 1);
        }
    }
    if (c->use_mm && !rc) {
        // single-product kernel: blocks padded to 1024 SNPs (one slot per SNP); + read-ahead rows (up to 24 groups)
        const int64_t Bpad = std::max<int64_t>(round_up(c->Bmax, 1024), 2 * UV_CHS), slots_max = Bpad;
        rc |= c->wt.alloc(sizeof(uint32_t) * (size_t)(slots_max / 8 + 96) * (size_t)c->ncols_pad);
        rc |= c->acc_f64.alloc(sizeof(double) * plane * (size_t)c->n_f64);
        if (!rc) rc |= build_tile_grid(c, c->tg_mm, c->tg_mm_tab, MM_TILE_R, MM_TILE_C, MM_SUPER);
        // split-fp16 MFMAs for every SYRK table (GRM / PCA / EIGMIX: |z| <= ~1e3, small values only next to O(1)
        // ones; KING-homo: sqrt(p(1-p)) and p(1-p) are multiplied by 2^8 so that p(1-p) ~ 1e-6 stays in fp16's
        // normal range, the finaliser divides the sums by 2^16); SNPGPU_SYRK=f32 keeps the fp32-MFMA kernel
        const char *sy = getenv("SNPGPU_SYRK");
        c->mm_h3 = !(sy && std::string(sy) == "f32");
        int h3_super = H3_SUPER;
        if (const char *e = getenv("SNPGPU_H3_SUPER")) { const int v = atoi(e); if (v >= 1 && v <= 32) h3_super = v; }   // tuning
        if (c->mm_h3 && !rc) rc |= build_worklist(c, H3_TILE_R, H3_TILE_C, h3_super, c->h3_work, c->h3_blocks);
        // exact-row-side kernel for blocks without missing calls (tables of the form y (g - avg) only)
        c->h3_exact_rows = c->mm_h3 && !(sy && std::string(sy) == "h3") &&
                           (c->lut_mode[0] == LUT_GCTA || c->lut_mode[0] == LUT_BAYES || c->lut_mode[0] == LUT_EIGMIX_NUM);
        // row operand of the two-product kernel per table: -1 none, 0 g - 1 (blocks without missing calls),
        // 1 call indicator (KING-homo weights), 2 missing indicator (EIGMIX both-missing weights)
        for (int i = 0; i < c->n_lut; i++) {
            const int m = c->lut_mode[i];
            c->h3_a_kind[i] = !c->mm_h3 || (sy && std::string(sy) == "h3") ? -1
                              : (i == 0 && c->h3_exact_rows) ? 0
                              : (m == LUT_HOMO_W1 || m == LUT_HOMO_W2) ? 1 : (m == LUT_EIGMIX_MISSW) ? 2 : -1;
        }
        // the exact-row kernel with one wave per SIMD (syrk_x1_kernel) where it applies (GRM / PCA); SNPGPU_SYRK_X1=0: two waves
        // per SIMD (syrk_h3_kernel<2, true>, the round-2 kernel before it; measurement only)
        // (EIGMIX: the list only serves the single-product kernel of its blocks without missing calls)
        if (c->h3_exact_rows && !(getenv("SNPGPU_SYRK_X1") && !atoi(getenv("SNPGPU_SYRK_X1"))) &&
            !getenv("SNPGPU_SYRK_MISS3") && !rc)
        {
            int xs = H3_SUPER / 2;
            if (const char *e = getenv("SNPGPU_X1_SUPER")) { const int v = atoi(e); if (v >= 1 && v <= 32) xs = v; }   // tuning
            rc |= build_worklist(c, X1_TILE, X1_TILE, xs, c->x1_work, c->x1_blocks, 1);
        }
        // |w| = y^2 |g - avg| <= 4N(1 + 1/N) in a block without missing calls (num = N; singleton: p = 1/2N): keep it
        // below 2^15 by moving a power of two to the (exact) row operand.  EIGMIX has y = 1.
        if (c->h3_exact_rows && c->lut_mode[0] != LUT_EIGMIX_NUM)
            while (ldexp(4.04 * (double)c->N, -c->h3_w_shift) > 32768.0) c->h3_w_shift++;
        // the exact-row kernel also for blocks WITH missing calls (GRM / PCA; EIGMIX shares its words with the 8-byte-entry
        // table of the both-missing weights and keeps three products there).  SNPGPU_SYRK_MISS3=1: three products for
        // blocks with missing calls, as in round 1 (A/B measurements).
        c->h3_exact_missing = c->h3_exact_rows && kind != SNPGPU_EIGMIX && !getenv("SNPGPU_SYRK_MISS3");
        // fp32 run lengths (snpgpu_internal.h: H3_PROMOTE_*): SNPGPU_SYRK_FAST=1 = one 32 768-SNP run per flush and one weight
        // target (round 2's kernels: 1.6e-5 instead of < 1e-5 in the off-diagonal figure); SNPGPU_H3_PROMOTE sets both run
        // lengths (measurements); SNPGPU_UV_TARGETS=0: one weight target for every run (measurement)
        const bool fast = getenv("SNPGPU_SYRK_FAST") && atoi(getenv("SNPGPU_SYRK_FAST"));
        c->h3_promote = fast ? H3_PROMOTE_FAST : H3_PROMOTE_EXACT;
        c->uv_promote = fast ? H3_PROMOTE_FAST : H3_PROMOTE_UV;
        if (const char *pr = getenv("SNPGPU_H3_PROMOTE")) {
            const int v = atoi(pr);
            if (v >= 256 && v <= 65536 && (v % 256) == 0) c->h3_promote = c->uv_promote = v;
        }
        // blocks WITHOUT missing calls of a GRM / PCA context: the single-product kernel (syrk_uv_kernel: the SNP weight as
        // a product of two fp16 numbers, integer centres); SNPGPU_SYRK_UV=0: the exact-row kernel for every block
        c->uv_enabled = c->x1_blocks > 0 && c->h3_exact_missing && (c->lut_mode[0] == LUT_GCTA || c->lut_mode[0] == LUT_BAYES) &&
                        !(getenv("SNPGPU_SYRK_UV") && !atoi(getenv("SNPGPU_SYRK_UV")));
        // EIGMIX numerator sum (g_i - 2p)(g_j - 2p): weight 1 = 1 x 1, so the single-product form is EXACT there; its words
        // carry 8 * code for every block (the both-missing weight table and the three-product kernel of the blocks with
        // missing calls have 8-byte entries as well)
        c->uv_eigmix = c->x1_blocks > 0 && kind == SNPGPU_EIGMIX && c->lut_mode[0] == LUT_EIGMIX_NUM &&
                       !(getenv("SNPGPU_SYRK_UV") && !atoi(getenv("SNPGPU_SYRK_UV")));
        if (kind == SNPGPU_EIGMIX && !c->uv_eigmix) { c->x1_work.release(); c->x1_blocks = 0; }
        // a weight target per fp32 run of the single-product kernel (GRM / PCA; EIGMIX's weight 1 is exact)
        c->uv_targets = c->uv_enabled && !fast && !(getenv("SNPGPU_UV_TARGETS") && !atoi(getenv("SNPGPU_UV_TARGETS")));
        // rare variants of blocks WITH missing calls: their carriers' pairs in fp64 beside the exact-row kernel (GRM / PCA
        // weights only; SNPGPU_X1_SPARSE=0: everything in the dense product, as before)
        c->sparse_missing = c->uv_enabled && c->N >= X1_SPARSE_MIN_N && !(getenv("SNPGPU_X1_SPARSE") && !atoi(getenv("SNPGPU_X1_SPARSE")));
        c->x1_sparse_mac = X1_SPARSE_MAC;
        // ... and such blocks as 4096-SNP fp32 runs of the exact-row kernel (SNPGPU_X1_SHORT_RUNS=0: 8192 as every other block)
        c->x1_short_runs = !(getenv("SNPGPU_X1_SHORT_RUNS") && !atoi(getenv("SNPGPU_X1_SHORT_RUNS")));
        if (const char *e = getenv("SNPGPU_X1_SPARSE_MAC")) c->x1_sparse_mac = std::max(1, std::min(atoi(e), X1_SPARSE_MAC));
        c->uv_enabled = c->uv_enabled || c->uv_eigmix;
        // EIGMIX blocks WITH missing calls: the numerator on the exact-row kernel as well (round 3; the three-product kernel it
        // took before drops lo lo': 2.3e-5 of the off-diagonal scale at L = 1e6).  Its 12-byte entries need 12 * code words:
        // a second transposition for such blocks (the both-missing weight table keeps its 8 * code words).
        // SNPGPU_SYRK_MISS3=1: three products as before (measurement)
        c->eigmix_x1 = c->uv_eigmix && !getenv("SNPGPU_SYRK_MISS3");
        if (c->eigmix_x1 && !rc) rc |= c->wt12.alloc(sizeof(uint32_t) * (size_t)(slots_max / 8 + 96) * (size_t)c->ncols_pad);
        if (c->h3_exact_rows && !rc) {
            rc |= c->ccoef.alloc(sizeof(double2) * (size_t)(c->Bmax + 2048));
            rc |= c->tcorr.alloc(sizeof(double) * (size_t)((c->uv_enabled ? 5 : 2) * Bpad / H3_LUTCH + 16) * (size_t)c->ncols_pad);
            rc |= c->colterm.alloc(sizeof(double) * (size_t)c->ncols_pad);
        }
        if (c->uv_enabled && !rc) {
            rc |= c->uvcoef.alloc(sizeof(double4) * (size_t)(slots_max + 512));
            rc |= c->uvsp.alloc(sizeof(double4) * (size_t)(Bpad + 512));
            rc |= c->uvkpart.alloc(sizeof(double) * (size_t)(slots_max / UV_CHUNK + 16));
            rc |= c->uvterm.alloc(sizeof(double) * (size_t)(2 * c->ncols_pad + 2));
            rc |= c->uvlut.alloc(64 * (size_t)(slots_max + 2048));          // 16 entries of 8 bytes per slot pair, whole 1024-slot chunks
            rc |= c->uvslot.alloc(sizeof(int32_t) * (size_t)(2 * slots_max + 64));          // slot -> SNP, SNP -> slot
            // per SNP: {t, avg} (16 bytes), per SNP and target: relative error (float) and u | v << 16
            rc |= c->uvcand.alloc((size_t)(slots_max + 64) * (16 + 8 * UV_QMAX));
        }
    }
    // KING-homo, blocks with missing calls (round 5): masked weight sums = totals - per-sample missing sums + ONE fp16 product of
    // binary operands per weight (homo_uv_tables_kernel, syrk_uv_kernel) instead of two-product SYRKs of an indicator against a
    // hi / lo operand.  Needs the two-scalar form of the blocks without missing calls (the binary counter kernel's contexts);
    // SNPGPU_HOMO_UV=0: the two-product kernels as before
    c->homo_uv = kind == SNPGPU_KING_HOMO && c->mm_h3 && c->het.p != nullptr && !(getenv("SNPGPU_HOMO_UV") && !atoi(getenv("SNPGPU_HOMO_UV")));
    // the single-product kernel on v_mfma_f32_16x16x32_f16 (round 6: the same products and fp32 runs, half the accumulator traffic per flop
    // under the socket power cap).  SNPGPU_SYRK_UV16: 0 = the 32x32x16 form (syrk_uv_kernel); 1 = syrk_uv16_kernel (operands looked up in
    // LDS tables -- what KING-homo's binary tables and EIGMIX always take); 2 = syrk_uv16c_kernel (GRM / PCA contexts: nibble words, one
    // v_cvt_scalef32_pk_f16_fp4 + one v_pk_fma_f16 per operand dword, no tables); 3 = ... and a work item walks the fp32 runs of its tile
    // itself, half of its sub-tile sums carried in the freed LDS between runs.  Default 3 with the pace-maker fetches on (SNPGPU_UVC_PACE=16):
    // -2.4 % per step against the lookup form, panel writes -40 %, word fetches -12 % (profiles/r06_uvc_ab.txt; without the pace-maker the
    // workgroups of an XCD drift apart and fetch 2.4 x the words)
    const int uv16_mode = getenv("SNPGPU_SYRK_UV16") ? std::max(0, std::min(atoi(getenv("SNPGPU_SYRK_UV16")), 3)) : 3;
    c->uv16 = uv16_mode != 0;
    c->uvc = uv16_mode >= 2 && c->uv_enabled && !c->uv_eigmix;
    c->uvc_carry = c->uvc && uv16_mode == 3;
    if (c->uvc && !rc) {
        // the pace-maker (see syrk_uv16c_kernel): 64 KiB per table chunk, fetched by every workgroup alongside its 8 KiB of factors
        c->uvc_pace = getenv("SNPGPU_UVC_PACE") ? (atoi(getenv("SNPGPU_UVC_PACE")) != 0) : 1;      // on / off (the size is fixed: 16 KiB per wave)
        const int64_t Bp = std::max<int64_t>(round_up(c->Bmax, 1024), 2 * UV_CHS);
        rc |= c->uvpace.alloc((size_t)65536 * (size_t)(Bp / UV_CHS + 4));
        if (!rc) rc |= (hipMemset(c->uvpace.p, 0, c->uvpace.bytes) != hipSuccess);
    }
    if (c->homo_uv && !rc) {
        const int64_t Bpad = std::max<int64_t>(round_up(c->Bmax, 1024), 2 * UV_CHS);
        for (int i = 0; i < 2; i++) rc |= c->homo_lut[i].alloc(64 * (size_t)(Bpad + 2048));
        rc |= c->homo_wts.alloc(sizeof(double2) * (size_t)(Bpad + 2048));
        rc |= c->homo_tc.alloc(sizeof(double2) * (size_t)(Bpad / (8 * (H3_LUTCH / 16)) + 16) * (size_t)c->ncols_pad);   // one partial per 256 SNPs
        rc |= c->homo_msum.alloc(sizeof(double) * 2 * (size_t)c->ncols_pad);
        if (!rc) rc |= (hipMemset(c->homo_msum.p, 0, c->homo_msum.bytes) != hipSuccess);
        if (!rc) rc |= build_worklist(c, X1_TILE, X1_TILE, H3_SUPER / 2, c->homo_work, c->homo_blocks, 1, 2);   // both weights in one launch
    }
    if (!rc) {
        hipError_t e = hipSuccess;
        if (c->acc_u32.p) e = hipMemsetAsync(c->acc_u32.p, 0, c->acc_u32.bytes, c->stream);
        if (e == hipSuccess && c->acc_f64.p) e = hipMemsetAsync(c->acc_f64.p, 0, c->acc_f64.bytes, c->stream);
        if (e == hipSuccess && c->miss_diag.p) e = hipMemsetAsync(c->miss_diag.p, 0, c->miss_diag.bytes, c->stream);
        if (e == hipSuccess && c->colterm.p) e = hipMemsetAsync(c->colterm.p, 0, c->colterm.bytes, c->stream);
        if (e == hipSuccess && c->uvterm.p) e = hipMemsetAsync(c->uvterm.p, 0, c->uvterm.bytes, c->stream);
        if (e == hipSuccess && c->samp_het.p) e = hipMemsetAsync(c->samp_het.p, 0, c->samp_het.bytes, c->stream);
        if (e == hipSuccess && c->samp_dmiss.p) e = hipMemsetAsync(c->samp_dmiss.p, 0, c->samp_dmiss.bytes, c->stream);
        if (e == hipSuccess && c->samp_dsq.p) e = hipMemsetAsync(c->samp_dsq.p, 0, c->samp_dsq.bytes, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(c->scalars.p, 0, c->scalars.bytes, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { set_error(std::string("snpgpu_create: memset failed: ") + hipGetErrorString(e)); rc = 1; }
    }
    if (rc) {
        std::string keep = g_err;
        free_ctx(c);
        set_error("snpgpu_create: " + keep);
        return 1;
    }
    *out = c;
    return 0;
}

int snpgpu_destroy(snpgpu_ctx *ctx)
{
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_ctx(ctx);
    return 0;
}

namespace {
struct EvScope {   // records a start/stop event pair around one launch when timing is on
    snpgpu_ctx *c; int which; hipEvent_t a = nullptr, b = nullptr;
    EvScope(snpgpu_ctx *c_, int w) : c(c_), which(w)
    {
        if (!c->timing) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        (void)hipEventRecord(a, c->stream);
    }
    ~EvScope()
    {
        if (!a) return;
        (void)hipEventRecord(b, c->stream);
        c->ev[which].push_back({a, b});
    }
};
}  // namespace

int snpgpu_set_timing(snpgpu_ctx *c, int enable)
{
    if (!c) { set_error("snpgpu_set_timing: NULL context"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(c->device));
    SNPGPU_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int w = 0; w < 2; w++) {
        for (auto &p : c->ev[w]) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
        c->ev[w].clear();
    }
    c->timing = enable != 0;
    return 0;
}

int snpgpu_get_timing(snpgpu_ctx *c, int which, double *ms_sum, int64_t *launches)
{
    if (!c || which < 0 || which > 1) { set_error("snpgpu_get_timing: invalid arguments"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(c->device));
    SNPGPU_HIP_CHECK(hipStreamSynchronize(c->stream));
    double s = 0;
    for (auto &p : c->ev[which]) {
        float ms = 0;
        SNPGPU_HIP_CHECK(hipEventElapsedTime(&ms, p.first, p.second));
        s += ms;
    }
    if (ms_sum) *ms_sum = s;
    if (launches) *launches = (int64_t)c->ev[which].size();
    return 0;
}

// per-SNP statistics handed in by the caller (snpgpu_feed_stats): the block's "holds missing calls" flag from them
__global__ __launch_bounds__(256) void stats_flag_kernel(const int32_t *__restrict__ num, int64_t n_snp, int64_t N,
                                                         unsigned long long *__restrict__ d_missing)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n_snp && (int64_t)num[k] < N) *d_missing = 1ull;
}

static int feed_impl(snpgpu_ctx *c, const void *geno, int64_t n_snp, int format, int mem, const int32_t *ext_sum, const int32_t *ext_num);

int snpgpu_feed(snpgpu_ctx *c, const void *geno, int64_t n_snp, int format, int mem)
{
    return feed_impl(c, geno, n_snp, format, mem, nullptr, nullptr);
}

int snpgpu_feed_stats(snpgpu_ctx *c, const void *geno, int64_t n_snp, int format, int mem, const int32_t *sum, const int32_t *num)
{
    if (!sum || !num) { set_error("snpgpu_feed_stats: NULL statistics"); return 1; }
    return feed_impl(c, geno, n_snp, format, mem, sum, num);
}

int snpgpu_block_stats(snpgpu_ctx *c, const void *geno, int64_t n_snp, int format, int32_t *sum, int32_t *num)
{
    if (!c || !geno || !sum || !num || n_snp < 0 || n_snp > c->Bmax) { set_error("snpgpu_block_stats: invalid arguments"); return 1; }
    if (format != SNPGPU_GENO_U8 && format != SNPGPU_GENO_PACKED2) { set_error("snpgpu_block_stats: invalid format"); return 1; }
    if (n_snp == 0) return 0;
    SNPGPU_HIP_CHECK(hipSetDevice(c->device));
    // (the aligned 2-bit copy lands in the context's own block buffer, which the next feed rewrites anyway; the flag word is
    // cleared by that feed as well)
    return launch_repack_stats(c->stream, geno, format, n_snp, c->N, (uint8_t *)c->packed.p, c->RB, sum, num, c->d_missing());
}

static int feed_impl(snpgpu_ctx *c, const void *geno, int64_t n_snp, int format, int mem, const int32_t *ext_sum, const int32_t *ext_num)
{
    if (!c) { set_error("snpgpu_feed: NULL context"); return 1; }
    if (n_snp == 0) return 0;
    if (!geno || n_snp < 0) { set_error("snpgpu_feed: invalid block"); return 1; }
    if (n_snp > c->Bmax) { set_error("snpgpu_feed: block larger than max_block_snps"); return 1; }
    if (c->frozen) { set_error("snpgpu_feed: the context was finalised in place (snpgpu_finalize_inplace); no blocks may follow"); return 1; }
    if (format != SNPGPU_GENO_U8 && format != SNPGPU_GENO_PACKED2) { set_error("snpgpu_feed: invalid format"); return 1; }
    if (c->kind == SNPGPU_KING_ROBUST && c->n_snp_total + n_snp >= 1073741824LL) {
        // guard of gnrIBD_KING_Robust, src/genKING.cpp:598-602
        set_error("The number of SNPs should be less than 1,073,741,824.");
        return 1;
    }
    SNPGPU_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t in_bytes = (size_t)n_snp * (size_t)(format == SNPGPU_GENO_U8 ? c->N : (c->N + 3) / 4);
    const void *src = geno;
    int turn = -1;
    if (mem == SNPGPU_HOST_PINNED) {
        if (!c->copy_stream) {
            SNPGPU_HIP_CHECK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
            for (int k = 0; k < 2; k++) {
                SNPGPU_HIP_CHECK(hipEventCreateWithFlags(&c->ev_copied[k], hipEventDisableTiming));
                SNPGPU_HIP_CHECK(hipEventCreateWithFlags(&c->ev_consumed[k], hipEventDisableTiming));
            }
        }
        turn = c->raw_turn;
        c->raw_turn ^= 1;
        if (c->raw2[turn].bytes < in_bytes) {
            SNPGPU_HIP_CHECK(hipStreamSynchronize(st));
            SNPGPU_HIP_CHECK(hipStreamSynchronize(c->copy_stream));
            c->raw2[turn].release();
            const size_t want = (size_t)c->Bmax * (size_t)(format == SNPGPU_GENO_U8 ? c->N : (c->N + 3) / 4);
            if (c->raw2[turn].alloc(want)) return 1;
        } else if (c->host_src[turn]) {
            // the device buffer may be overwritten only after the repack that read it
            SNPGPU_HIP_CHECK(hipStreamWaitEvent(c->copy_stream, c->ev_consumed[turn], 0));
        }
        SNPGPU_HIP_CHECK(hipMemcpyAsync(c->raw2[turn].p, geno, in_bytes, hipMemcpyHostToDevice, c->copy_stream));
        SNPGPU_HIP_CHECK(hipEventRecord(c->ev_copied[turn], c->copy_stream));
        SNPGPU_HIP_CHECK(hipStreamWaitEvent(st, c->ev_copied[turn], 0));
        c->host_src[turn] = geno;
        src = c->raw2[turn].p;
    } else if (mem == SNPGPU_HOST) {
        if (c->raw.bytes < in_bytes) {  // host feeds are staged through a device copy of the raw block
            SNPGPU_HIP_CHECK(hipStreamSynchronize(st));
            c->raw.release();
            const size_t want = (size_t)c->Bmax * (size_t)(format == SNPGPU_GENO_U8 ? c->N : (c->N + 3) / 4);
            if (c->raw.alloc(want)) return 1;
        }
        SNPGPU_HIP_CHECK(hipMemcpyAsync(c->raw.p, geno, in_bytes, hipMemcpyHostToDevice, st));
        src = c->raw.p;
    }
    uint8_t *packed = (uint8_t *)c->packed.p;
    SNPGPU_HIP_CHECK(hipMemsetAsync(c->d_missing(), 0, sizeof(unsigned long long), st));
    SNPGPU_HIP_CHECK(hipMemsetAsync(c->d_short_runs(), 0, sizeof(unsigned long long), st));
    // IBS / KING-robust counters fed with 2-bit rows: one pre-pass kernel straight from the caller's block (no statistics
    // are needed by these kinds beyond the missing-call flag); SNPGPU_PREP_TWO_PASS=1 keeps the two-kernel form
    const bool direct = c->use_pc && c->pc_i8 && !c->use_mm && (c->pc_mode == PM_IBS || c->pc_mode == PM_KING_ROBUST) &&
                        format == SNPGPU_GENO_PACKED2 && (((c->N + 3) / 4) % 4) == 0 && !ext_sum &&
                        (reinterpret_cast<uintptr_t>(src) & 3u) == 0 && !getenv("SNPGPU_PREP_TWO_PASS");
    if (direct) {
        const int64_t n_pad = round_up(n_snp, 256);        // whole loop rounds of the pair kernels (4 k-steps of 64 SNPs for the fp4 form)
        if (launch_transpose2_direct(st, (const uint8_t *)src, c->N, n_snp, c->col0, c->ncols_pad, (int)(n_pad / 16),
                                     (uint32_t *)c->w2.p, (uint32_t *)c->het.p, (uint32_t *)c->het_blk.p, c->d_missing()))
            return 1;
        if (turn >= 0) SNPGPU_HIP_CHECK(hipEventRecord(c->ev_consumed[turn], st));
        if (c->het.p) c->het_pending = true;
        {
            EvScope ev(c, 0);
            if (launch_pair_i8(st, c->pc_mode, (const int4 *)c->i8_work.p, c->i8_blocks, (const uint32_t *)c->w2.p,
                               c->ncols_pad, (int)(n_pad / 32), (int)n_snp, (uint32_t *)c->acc_u32.p, c->plane(),
                               c->het.p ? c->d_missing() : nullptr, (const int4 *)c->i8_work_nm.p, c->i8_blocks_nm, c->nomiss_fp4, c->general_fp4))
                return 1;
        }
        c->n_snp_total += n_snp;
        if (mem == SNPGPU_HOST) SNPGPU_HIP_CHECK(hipStreamSynchronize(st));
        return 0;
    }
    if (ext_sum) {
        // the caller computed this block's per-SNP statistics elsewhere (its share of the SNPs on every rank + an all-gather,
        // multigpu.py shared_stats): re-layout only, statistics and the missing-call flag from the arrays (device memory)
        if (launch_repack(st, src, format, n_snp, c->N, packed, c->RB)) return 1;
        SNPGPU_HIP_CHECK(hipMemcpyAsync(c->sum.p, ext_sum, sizeof(int32_t) * (size_t)n_snp, hipMemcpyDeviceToDevice, st));
        SNPGPU_HIP_CHECK(hipMemcpyAsync(c->num.p, ext_num, sizeof(int32_t) * (size_t)n_snp, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(stats_flag_kernel, dim3((unsigned)((n_snp + 255) / 256)), dim3(256), 0, st, (const int32_t *)c->num.p, n_snp, c->N,
                           c->d_missing());
    } else if (launch_repack_stats(st, src, format, n_snp, c->N, packed, c->RB, (int32_t *)c->sum.p, (int32_t *)c->num.p,
                                   c->d_missing()))
        return 1;
    if (turn >= 0) SNPGPU_HIP_CHECK(hipEventRecord(c->ev_consumed[turn], st));

    const int KW = (int)(2 * ((n_snp + 63) / 64));
    if (c->use_pc) {
        if (c->pc_mode == PM_GCTA_MISS && c->pc_i8) {
            const int64_t n_pad = round_up(n_snp, 256);    // whole loop rounds of the pair kernel (4 k-steps of 64 / up to 4 of 32 SNPs)
            if (launch_transpose2_missmask(st, packed, c->RB, n_snp, c->N, (const int32_t *)c->sum.p, (const int32_t *)c->num.p,
                                           c->col0, c->ncols_pad, (int)(n_pad / 16), (uint32_t *)c->w2.p,
                                           (uint32_t *)c->miss_diag.p, c->d_missing()))
                return 1;
            const bool sparse = c->sp_blocks > 0;
            if (sparse) {      // the block's route (device side): sparse sets up to sp_max_rate missing calls, the dense product beyond
                const unsigned long long max_cells = (unsigned long long)(c->sp_max_rate * (double)c->N * (double)n_snp);
                if (launch_missmask256(st, packed, c->RB, n_snp, c->N, (const int32_t *)c->sum.p, (const int32_t *)c->num.p, c->col0,
                                       (int)(c->ncols_pad / 256), round_up(c->Bmax, 256), (uint4 *)c->mm256.p, c->d_missing(), max_cells,
                                       c->d_miss_route()))
                    return 1;
            }
            {
                EvScope ev(c, 0);
                if (sparse && launch_pair_sparse_miss(st, (const uint4 *)c->mm256.p, round_up(c->Bmax, 256), (int)n_snp, (uint32_t *)c->acc_u32.p,
                                                      c->ncols_pad, (const int4 *)c->sp_work.p, c->sp_blocks, c->d_miss_route()))
                    return 1;
                const unsigned long long *run = sparse ? c->d_miss_route() + 1 : c->d_missing();
                if (c->miss_fp4 ? launch_pair_fp4_miss(st, (const int4 *)c->i8_work.p, c->i8_blocks, (const uint32_t *)c->w2.p, c->ncols_pad,
                                                       (int)(n_pad / 64), (uint32_t *)c->acc_u32.p, run)
                                : launch_pair_i8(st, c->pc_mode, (const int4 *)c->i8_work.p, c->i8_blocks, (const uint32_t *)c->w2.p,
                                                 c->ncols_pad, (int)(n_pad / 32), (int)n_snp, (uint32_t *)c->acc_u32.p, c->plane(), run))
                    return 1;
            }
        } else if (c->pc_mode == PM_GCTA_MISS) {
            if (launch_bitplanes_miss(st, packed, c->RB, n_snp, c->N, (const int32_t *)c->sum.p,
                                      (const int32_t *)c->num.p, c->col0, c->ncols_pad, c->rows_pad, KW,
                                      (uint2 *)c->rowp.p, (uint2 *)c->colp.p, c->d_missing()))
                return 1;
            if (launch_miss_diag(st, (const uint2 *)c->colp.p, KW / 2, c->ncols_pad, c->col0,
                                 (uint32_t *)c->miss_diag.p, c->d_missing()))
                return 1;
            {
                EvScope ev(c, 0);
                if (launch_pair_popcount(st, c->pc_mode, c->tg_pc, c->rowp.p, c->colp.p, KW, c->ncols_pad,
                                         (uint32_t *)c->acc_u32.p, c->plane(), c->d_missing()))
                    return 1;
            }
        } else if (c->pc_i8) {
            const int64_t n_pad = round_up(n_snp, 256);
            // (+ per-sample het counts of a block without missing calls, for the binary pair kernel)
            if (launch_transpose2(st, packed, c->RB, n_snp, c->col0, c->ncols_pad, (int)(n_pad / 16), (uint32_t *)c->w2.p,
                                  (uint32_t *)c->het.p, c->d_missing()))
                return 1;
            if (c->het.p) c->het_pending = true;
            {
                EvScope ev(c, 0);
                if (launch_pair_i8(st, c->pc_mode, (const int4 *)c->i8_work.p, c->i8_blocks, (const uint32_t *)c->w2.p,
                                   c->ncols_pad, (int)(n_pad / 32), (int)n_snp, (uint32_t *)c->acc_u32.p, c->plane(),
                                   c->het.p ? c->d_missing() : nullptr, (const int4 *)c->i8_work_nm.p, c->i8_blocks_nm, c->nomiss_fp4, c->general_fp4))
                    return 1;
            }
        } else {
            if (launch_bitplanes4(st, packed, c->RB, n_snp, c->N, c->col0, c->ncols_pad, c->rows_pad, KW,
                                  (uint4 *)c->rowp.p, (uint4 *)c->colp.p))
                return 1;
            {
                EvScope ev(c, 0);
                if (launch_pair_popcount(st, c->pc_mode, c->tg_pc, c->rowp.p, c->colp.p, KW, c->ncols_pad,
                                         (uint32_t *)c->acc_u32.p, c->plane(), nullptr))
                    return 1;
            }
        }
    }
    if (c->use_mm) {
        // syrk_x1_kernel walks rounds of eight 16-SNP groups, syrk_uv_kernel of sixteen
        // the single-product kernel runs a block as fp32 runs of `cpr` table chunks (one launch and one fp64 flush each); with
        // more than one run every run carries its own weight target and the block's SNPs are dealt to the runs (uv_assign_kernel)
        // (at least two runs = two targets: one target leaves the weights at 1.05e-6 rms, 1.1e-5 at worst over the 1.6e8 entries
        // of an 18 000-sample panel; a block of a single table chunk is spread over two half-empty ones -- twice the MFMA work
        // of a block that is small anyway)
        int uv_runs = 1, uv_cpr = 1, uv_chunks = 0;
        const bool uv_blk = c->uv_enabled;
        if (c->uv_enabled) {
            const int64_t run0 = std::max<int64_t>(UV_CHS, (int64_t)c->uv_promote / UV_CHS * UV_CHS);
            const bool targets = c->uv_targets && !c->uv_eigmix;
            const int n_chunk = std::max((int)(round_up(n_snp, UV_CHS) / UV_CHS), targets ? 2 : 1);
            uv_chunks = n_chunk;
            if (n_snp > run0 || targets) {
                const int runs0 = std::max(targets ? 2 : 1, (int)((n_chunk * (int64_t)UV_CHS + run0 - 1) / run0));
                uv_cpr = (n_chunk + runs0 - 1) / runs0;                  // balanced: 32 chunks at <= 11 per run = 11 + 11 + 10
                uv_runs = (n_chunk + uv_cpr - 1) / uv_cpr;
            }
        }
        const int uv_q = (uv_blk && c->uv_targets && !c->uv_eigmix && uv_runs > 1) ? std::min(uv_runs, UV_QMAX) : 1;
        const int64_t n_pad = (uv_blk && uv_runs > 1) ? (int64_t)uv_chunks * UV_CHS
                                                      : round_up(n_snp, (uv_blk || c->homo_uv) ? 256 : c->x1_blocks ? 128 : 64);
        const int n_q = (int)(n_pad / 16);    // groups of 16 SNPs (= 2 pair-coded dwords per sample)
        const int64_t n_slots = n_pad;        // the single-product kernel's K dimension: one slot per SNP
        int32_t *slot_src = (int32_t *)c->uvslot.p, *slot_of = slot_src ? slot_src + (c->uvslot.bytes / 8) : nullptr;
        // tables, row / column coefficients and the slot -> SNP map of a block without missing calls (table 0 of GRM / PCA /
        // EIGMIX contexts) come first: the transposition below follows the map
        if (uv_blk && c->h3_a_kind[0] == 0) {
            char *cb = (char *)c->uvcand.p;
            const size_t nmax = c->uvcand.bytes / (16 + 8 * UV_QMAX);
            if (launch_build_uv(st, (const int32_t *)c->sum.p, (const int32_t *)c->num.p, n_snp, n_pad, c->lut_mode[0],
                                (uint2 *)c->uvlut.p, (double4 *)c->uvcoef.p, (double *)c->uvkpart.p, (double4 *)c->uvsp.p,
                                (float *)(cb + 16 * nmax), (uint32_t *)(cb + (16 + 4 * UV_QMAX) * nmax), (double2 *)cb,
                                slot_of, slot_src, uv_q, uv_cpr, c->d_missing(), c->uvc ? 2 : c->uv16 ? 1 : 0))
                return 1;
        }
        if (launch_transpose8(st, packed, c->RB, n_snp, c->col0, c->ncols_pad, (int)(n_slots / 8), (uint32_t *)c->wt.p,
                              (c->h3_exact_rows && !c->uv_eigmix) ? c->d_missing() : nullptr,
                              c->uv_eigmix ? 0 : uv_blk ? 3 : c->x1_blocks ? 2 : (c->h3_exact_missing ? 1 : 0),
                              uv_q > 1 ? (const int32_t *)slot_src : nullptr, c->uvc ? 1 : 0))
            return 1;
        if (c->eigmix_x1 && launch_transpose8(st, packed, c->RB, n_snp, c->col0, c->ncols_pad, (int)(n_pad / 8), (uint32_t *)c->wt12.p,
                                              c->d_missing(), 4))
            return 1;
        // KING-homo: in a block without missing calls the masked weight sums are the same for every pair -- the table
        // pass adds them to two scalars, the SYRK of both tables exits (and the two-product counter kernel takes the block)
        const bool homo_nm = (c->kind == SNPGPU_KING_HOMO && c->het.p != nullptr);
        for (int i = 0; i < c->n_lut; i++) {
            unsigned long long *nl = (i == 0 && c->kind == SNPGPU_GRM_GCTA) ? c->d_nlocus() : nullptr;
            const bool eig0 = (c->kind == SNPGPU_EIGMIX && i == 0);
            if (launch_build_lut(st, (const int32_t *)c->sum.p, (const int32_t *)c->num.p, n_snp, n_pad,
                                 c->lut_mode[i], c->mm_h3 ? 1 : 0, (float2 *)c->lut[i].p, nl, eig0 ? c->d_sumden() : nullptr,
                                 eig0 ? (double *)c->dvals.p : nullptr, c->d_missing(),
                                 (i == 0 && c->h3_exact_rows) ? (double2 *)c->ccoef.p : nullptr, c->h3_a_kind[i] > 0,
                                 c->h3_w_shift, c->h3_exact_missing || (i == 0 && c->eigmix_x1), (i == 0 && c->x1_blocks) ? 1 : 0,
                                 (homo_nm ? c->d_homo_w() + i : nullptr),
                                 (i == 0 && c->sparse_missing) ? (double4 *)c->uvsp.p : nullptr, c->x1_sparse_mac,
                                 (i == 0 && c->sparse_missing && c->x1_short_runs) ? c->d_short_runs() : nullptr))
                return 1;
            const bool exact_rows = (c->h3_a_kind[i] == 0);
            const bool uv = exact_rows && uv_blk;
            // (EIGMIX with the single-product kernel: its exact-row kernel never runs, no column term)
            if (exact_rows && !c->uv_eigmix && launch_colcorr(st, (const uint32_t *)c->wt.p, c->ncols_pad, (int)(n_pad / 8),
                                             (const double2 *)c->ccoef.p, (double *)c->tcorr.p, (double *)c->colterm.p,
                                             c->d_missing(), uv ? 2 : c->h3_exact_missing, c->x1_blocks ? 1 : 0))
                return 1;
            if (uv) {     // a block without missing calls: rare variants in fp64, row / column terms of every slot
                if (launch_uv_sparse(st, packed, c->RB, n_snp, c->N, c->row0, c->row1, c->col0, (const double4 *)c->uvsp.p,
                                     (double *)c->acc_f64.p + (size_t)i * (size_t)c->plane(), c->ncols_pad, c->acc_tiles_c,
                                     c->ncols_pad, (double *)c->uvterm.p, c->d_missing()) ||
                    launch_uvcorr(st, (const uint32_t *)c->wt.p, c->ncols_pad, (int)(n_slots / 8), (const double4 *)c->uvcoef.p,
                                  (const double *)c->uvkpart.p, (int)(n_slots / UV_CHUNK), (double2 *)c->tcorr.p,
                                  (double *)c->uvterm.p, c->d_missing(), c->uvc ? 1 : 0))
                    return 1;
            }
            // a block WITH missing calls: what the carriers of its rare variants lack in the exact-row product
            if (exact_rows && c->sparse_missing &&
                launch_uv_sparse(st, packed, c->RB, n_snp, c->N, c->row0, c->row1, c->col0, (const double4 *)c->uvsp.p,
                                 (double *)c->acc_f64.p + (size_t)i * (size_t)c->plane(), c->ncols_pad, c->acc_tiles_c,
                                 c->ncols_pad, (double *)c->uvterm.p, c->d_missing(), 1))
                return 1;
            // EIGMIX numerator of a block with missing calls: the exact-row kernel's column term from the 12 * code words
            if (exact_rows && c->eigmix_x1 && launch_colcorr(st, (const uint32_t *)c->wt12.p, c->ncols_pad, (int)(n_pad / 8),
                                                             (const double2 *)c->ccoef.p, (double *)c->tcorr.p, (double *)c->colterm.p,
                                                             c->d_missing(), 2, 1))
                return 1;
            if (exact_rows) c->colterm_pending = true;
            if (eig0 && launch_eigmix_samples(st, (const uint32_t *)c->wt.p, (int)(n_pad / 8), c->ncols_pad, c->col0,
                                              (const double *)c->dvals.p, (uint32_t *)c->samp_het.p,
                                              (double *)c->samp_dmiss.p, (double *)c->samp_dsq.p,
                                              (c->h3_exact_rows && !c->uv_eigmix) ? c->d_missing() : nullptr))
                return 1;
            // the weighted both-missing sums are only needed for blocks that contain missing calls
            const unsigned long long *skip = (c->lut_mode[i] == LUT_EIGMIX_MISSW || uv || homo_nm) ? c->d_missing() : nullptr;
            if (c->homo_uv) {
                // KING-homo block with missing calls: tables, effective weights, totals and per-sample missing sums of BOTH weights
                // once (i == 0), then one single-product launch per weight into its plane
                if (i == 0 && launch_homo_uv(st, (const int32_t *)c->sum.p, (const int32_t *)c->num.p, n_snp, n_pad, (uint2 *)c->homo_lut[0].p,
                                             (uint2 *)c->homo_lut[1].p, (double2 *)c->homo_wts.p, c->d_homo_w(), (const uint32_t *)c->wt.p,
                                             c->ncols_pad, (double2 *)c->homo_tc.p, (double *)c->homo_msum.p, c->d_missing(), c->uv16 ? 1 : 0))
                    return 1;
                // (both weights in ONE launch: work items (tile, weight), the copy index picks table and plane)
                EvScope ev(c, 1);
                if (i == 0 && launch_syrk_uv(st, (const int4 *)c->homo_work.p, c->homo_blocks, (const uint32_t *)c->wt.p, c->ncols_pad,
                                             (const uint2 *)c->homo_lut[0].p, n_q, (double *)c->acc_f64.p, c->ncols_pad, c->acc_tiles_c,
                                             c->d_missing(), c->N - c->row0, 0, 1, 1,
                                             (int64_t)((const char *)c->homo_lut[1].p - (const char *)c->homo_lut[0].p), (int64_t)c->plane(),
                                             c->uv16 ? 1 : 0))
                    return 1;
                continue;
            }
            {
                EvScope ev(c, 1);
                double *accp = (double *)c->acc_f64.p + (size_t)i * (size_t)c->plane();
                if (c->mm_h3) {
                    const bool x1e = exact_rows && c->eigmix_x1;       // EIGMIX numerator: exact-row kernel on its own words
                    const bool x1m = exact_rows && (c->h3_exact_missing || x1e);
                    if (launch_syrk_h3(st, (const int4 *)c->h3_work.p, c->h3_blocks, (const uint32_t *)(x1e ? c->wt12.p : c->wt.p),
                                       c->ncols_pad, (const uint2 *)c->lut[i].p, n_q, accp, c->ncols_pad, c->acc_tiles_c, skip,
                                       c->h3_a_kind[i], x1m ? nullptr : c->d_missing(),
                                       c->N - c->row0, c->h3_promote,
                                       (x1m && c->x1_blocks) ? (const int4 *)c->x1_work.p : nullptr,
                                       c->x1_blocks, (i == 0 && c->sparse_missing && c->x1_short_runs) ? c->d_short_runs() : nullptr))
                        return 1;
                    if (uv && launch_syrk_uv(st, (const int4 *)c->x1_work.p, c->x1_blocks, (const uint32_t *)c->wt.p, c->ncols_pad,
                                             (const uint2 *)c->uvlut.p, (int)(n_slots / 16), accp, c->ncols_pad, c->acc_tiles_c, c->d_missing(),
                                             c->N - c->row0, uv_runs > 1 ? uv_cpr : 0, uv_q, 0, 0, 0, c->uvc_carry ? 3 : c->uvc ? 2 : c->uv16 ? 1 : 0,
                                             c->uvpace.p, c->uvc ? c->uvc_pace : 0))
                        return 1;
                } else if (launch_syrk(st, c->tg_mm, (const uint32_t *)c->wt.p, c->ncols_pad,
                                       (const float2 *)c->lut[i].p, n_q, accp, c->ncols_pad, c->acc_tiles_c, skip))
                    return 1;
            }
        }
    }
    c->n_snp_total += n_snp;
    c->acc_f32_valid = false;       // (the eigen solver's fp32 copy of the sums is stale now)
    if (mem == SNPGPU_HOST) SNPGPU_HIP_CHECK(hipStreamSynchronize(st));  // caller may reuse its buffer
    return 0;
}

int snpgpu_host_alloc(size_t bytes, void **out)
{
    if (!out) { set_error("snpgpu_host_alloc: out is NULL"); return 1; }
    SNPGPU_HIP_CHECK(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
    return 0;
}

int snpgpu_host_free(void *p)
{
    if (p) SNPGPU_HIP_CHECK(hipHostFree(p));
    return 0;
}

int snpgpu_host_wait(snpgpu_ctx *c, const void *host_buf)
{
    if (!c) { set_error("snpgpu_host_wait: NULL context"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(c->device));
    for (int k = 0; k < 2; k++)
        if (c->host_src[k] == host_buf && c->ev_copied[k]) SNPGPU_HIP_CHECK(hipEventSynchronize(c->ev_copied[k]));
    return 0;
}

int snpgpu_sync(snpgpu_ctx *c)
{
    if (!c) return 0;
    SNPGPU_HIP_CHECK(hipSetDevice(c->device));
    if (c->copy_stream) SNPGPU_HIP_CHECK(hipStreamSynchronize(c->copy_stream));
    SNPGPU_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int snpgpu_counts(snpgpu_ctx *c, int64_t *n_snp_total, int64_t *n_locus)
{
    if (!c) { set_error("snpgpu_counts: NULL context"); return 1; }
    if (n_snp_total) *n_snp_total = c->n_snp_total;
    if (n_locus) {
        SNPGPU_HIP_CHECK(hipSetDevice(c->device));
        unsigned long long v = 0;
        SNPGPU_HIP_CHECK(hipMemcpyAsync(&v, c->d_nlocus(), sizeof(v), hipMemcpyDeviceToHost, c->stream));
        SNPGPU_HIP_CHECK(hipStreamSynchronize(c->stream));
        *n_locus = (int64_t)v;
    }
    return 0;
}

int64_t snpgpu_slab_size(const snpgpu_ctx *c)
{
    if (!c) return 0;
    // rows row0..row1-1 of the packed triangle: sum over i of (N - i)
    const int64_t r = c->row1 - c->row0;
    return r * c->N - (c->row0 + c->row1 - 1) * r / 2;
}

}  // extern "C"

// column term of the exact-row SYRK (table 0 only): applied to the panel before anything reads the sums
int snpgpu::ctx_settle(snpgpu_ctx *c)
{
    if (!c->colterm_pending) return 0;
    SNPGPU_HIP_CHECK(hipSetDevice(c->device));
    const int64_t rows_real = std::min<int64_t>(c->row1 - c->row0, c->N - c->row0);
    if (launch_colterm_settle(c->stream, (double *)c->acc_f64.p, c->ncols_pad, c->acc_tiles_c, rows_real, c->ncols_pad, c->N - c->col0,
                              (double *)c->colterm.p, (double *)c->uvterm.p))
        return 1;
    c->colterm_pending = false;
    return 0;
}

// ---------------------------------------------------------------------------
// output staging: finalisers write device buffers; host destinations go through a temporary
namespace {

struct OutBuf {
    snpgpu_ctx *c;
    void *user;
    void *dev = nullptr;
    size_t bytes;
    int mem;
    bool temp = false;
    OutBuf(snpgpu_ctx *c_, void *user_, size_t bytes_, int mem_) : c(c_), user(user_), bytes(bytes_), mem(mem_) {}
    int prepare()
    {
        if (mem == SNPGPU_DEVICE) { dev = user; return 0; }
        SNPGPU_HIP_CHECK(hipMalloc(&dev, bytes ? bytes : 16));
        temp = true;
        return 0;
    }
    int commit()
    {
        if (temp) SNPGPU_HIP_CHECK(hipMemcpyAsync(user, dev, bytes, hipMemcpyDeviceToHost, c->stream));
        return 0;
    }
    ~OutBuf()
    {
        if (temp && dev) {
            (void)hipStreamSynchronize(c->stream);
            (void)hipFree(dev);
        }
    }
};

int settle_colterm(snpgpu_ctx *c) { return snpgpu::ctx_settle(c); }

int check_out(snpgpu_ctx *c, int kind_a, int kind_b, int packed, const char *fn, bool settle = true)
{
    if (!c) { set_error(std::string(fn) + ": NULL context"); return 1; }
    if (c->kind != kind_a && c->kind != kind_b) { set_error(std::string(fn) + ": wrong context kind"); return 1; }
    if (!packed && !c->full) { set_error(std::string(fn) + ": full-matrix output needs a full (non-panel) context"); return 1; }
    if (settle && settle_colterm(c)) return 1;
    if (c->het_pending) {       // rank-one terms of the blocks the binary pair kernel took
        SNPGPU_HIP_CHECK(hipSetDevice(c->device));
        const bool homo = (c->pc_mode == PM_KING_HOMO);     // planes {ibs1, 2 ibs0} instead of {n, ibs1, 2 ibs0, ...}
        if (launch_het_settle(c->stream, (uint32_t *)c->acc_u32.p, c->plane(), c->rows_pad, c->ncols_pad, (uint32_t *)c->het.p,
                              c->kind == SNPGPU_KING_ROBUST, homo ? 0 : 1, homo ? 1 : 2))
            return 1;
        c->het_pending = false;
    }
    if (hipSetDevice(c->device) != hipSuccess) { set_error(std::string(fn) + ": hipSetDevice failed"); return 1; }
    return 0;
}

size_t out_elems(snpgpu_ctx *c, int packed) { return packed ? (size_t)snpgpu_slab_size(c) : (size_t)c->N * (size_t)c->N; }

int finish(snpgpu_ctx *c)
{
    SNPGPU_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

}  // namespace

extern "C" {

int snpgpu_ibs_num(snpgpu_ctx *c, int32_t *ibs0, int32_t *ibs1, int32_t *ibs2, int packed, int mem)
{
    if (check_out(c, SNPGPU_IBS, SNPGPU_IBS, packed, "snpgpu_ibs_num")) return 1;
    const size_t n = out_elems(c, packed) * sizeof(int32_t);
    OutBuf b0(c, ibs0, n, mem), b1(c, ibs1, n, mem), b2(c, ibs2, n, mem);
    if (b0.prepare() || b1.prepare() || b2.prepare()) return 1;
    if (launch_fin_ibs_num(c->stream, c->geom(), (const uint32_t *)c->acc_u32.p, (int32_t *)b0.dev, (int32_t *)b1.dev,
                           (int32_t *)b2.dev, packed))
        return 1;
    if (b0.commit() || b1.commit() || b2.commit()) return 1;
    return finish(c);
}

int snpgpu_ibs_ave(snpgpu_ctx *c, double *out, int packed, int mem)
{
    if (check_out(c, SNPGPU_IBS, SNPGPU_IBS, packed, "snpgpu_ibs_ave")) return 1;
    OutBuf b(c, out, out_elems(c, packed) * sizeof(double), mem);
    if (b.prepare()) return 1;
    if (launch_fin_ibs_ave(c->stream, c->geom(), (const uint32_t *)c->acc_u32.p, (double *)b.dev, packed)) return 1;
    if (b.commit()) return 1;
    return finish(c);
}

int snpgpu_king_robust_counts(snpgpu_ctx *c, uint32_t *out5, int mem)
{
    if (check_out(c, SNPGPU_KING_ROBUST, SNPGPU_KING_ROBUST, 1, "snpgpu_king_robust_counts")) return 1;
    OutBuf b(c, out5, out_elems(c, 1) * 5 * sizeof(uint32_t), mem);
    if (b.prepare()) return 1;
    if (launch_fin_king_counts(c->stream, c->geom(), (const uint32_t *)c->acc_u32.p, (uint32_t *)b.dev)) return 1;
    if (b.commit()) return 1;
    return finish(c);
}

int snpgpu_king_robust(snpgpu_ctx *c, const int32_t *family, double *ibs0, double *kinship, int packed, int mem)
{
    if (check_out(c, SNPGPU_KING_ROBUST, SNPGPU_KING_ROBUST, packed, "snpgpu_king_robust")) return 1;
    const int32_t *dfam = nullptr;
    if (family) {
        if (!c->family.p && c->family.alloc(sizeof(int32_t) * (size_t)c->N)) return 1;
        SNPGPU_HIP_CHECK(hipMemcpyAsync(c->family.p, family, sizeof(int32_t) * (size_t)c->N, hipMemcpyHostToDevice, c->stream));
        dfam = (const int32_t *)c->family.p;
    }
    const size_t n = out_elems(c, packed) * sizeof(double);
    OutBuf b0(c, ibs0, n, mem), b1(c, kinship, n, mem);
    if (b0.prepare() || b1.prepare()) return 1;
    if (launch_fin_king_robust(c->stream, c->geom(), (const uint32_t *)c->acc_u32.p, dfam, (double *)b0.dev,
                               (double *)b1.dev, packed))
        return 1;
    if (b0.commit() || b1.commit()) return 1;
    return finish(c);
}

int snpgpu_king_homo(snpgpu_ctx *c, double *k0, double *k1, int packed, int mem)
{
    if (check_out(c, SNPGPU_KING_HOMO, SNPGPU_KING_HOMO, packed, "snpgpu_king_homo")) return 1;
    const size_t n = out_elems(c, packed) * sizeof(double);
    OutBuf b0(c, k0, n, mem), b1(c, k1, n, mem);
    if (b0.prepare() || b1.prepare()) return 1;
    // split-fp16 tables are pre-scaled by 2^H3_HOMO_SHIFT (both operands): the sums carry 2^(2 shift)
    const double fscale = c->mm_h3 ? std::ldexp(1.0, -2 * H3_HOMO_SHIFT) : 1.0;
    if (launch_fin_king_homo(c->stream, c->geom(), (const uint32_t *)c->acc_u32.p, (const double *)c->acc_f64.p, fscale,
                             (double *)b0.dev, (double *)b1.dev, packed, c->het.p ? c->d_homo_w() : nullptr,
                             c->homo_uv ? (const double *)c->homo_msum.p : nullptr))
        return 1;
    if (b0.commit() || b1.commit()) return 1;
    return finish(c);
}

int snpgpu_grm_gcta(snpgpu_ctx *c, double *out, int packed, int mem)
{
    // the pending column / row terms of the fp16 SYRK are applied by the finaliser itself (one pass over the panel less)
    if (check_out(c, SNPGPU_GRM_GCTA, SNPGPU_GRM_GCTA, packed, "snpgpu_grm_gcta", false)) return 1;
    OutBuf b(c, out, out_elems(c, packed) * sizeof(double), mem);
    if (b.prepare()) return 1;
    if (c->frozen) {            // the panel already holds the final values (snpgpu_finalize_inplace)
        if (launch_fin_cov(c->stream, c->geom(), (const double *)c->acc_f64.p, 1.0, (double *)b.dev, packed)) return 1;
    } else if (launch_fin_gcta(c->stream, c->geom(), (const double *)c->acc_f64.p, (const uint32_t *)c->acc_u32.p,
                        (const uint32_t *)c->miss_diag.p, c->d_nlocus(), (double *)b.dev, packed,
                        c->colterm_pending ? (const double *)c->colterm.p : nullptr,
                        c->colterm_pending ? (const double *)c->uvterm.p : nullptr))
        return 1;
    if (b.commit()) return 1;
    return finish(c);
}

int snpgpu_pca_cov(snpgpu_ctx *c, double *out, int packed, int normalize, double trace_in, double *trace_xtx, int mem)
{
    if (check_out(c, SNPGPU_PCA_COV, SNPGPU_PCA_COV, packed, "snpgpu_pca_cov")) return 1;
    double tr = 0;
    if (launch_trace(c->stream, c->geom(), (const double *)c->acc_f64.p, c->d_trace())) return 1;
    SNPGPU_HIP_CHECK(hipMemcpyAsync(&tr, c->d_trace(), sizeof(double), hipMemcpyDeviceToHost, c->stream));
    SNPGPU_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (trace_xtx) *trace_xtx = tr;
    double scale = 1.0;
    if (normalize) {
        if (trace_in > 0) tr = trace_in;
        else if (!c->full) { set_error("snpgpu_pca_cov: normalisation of a panel needs trace_in"); return 1; }
        scale = (double)(c->N - 1) / tr;  // genPCA.cpp:1386-1390
    }
    if (!out) return 0;
    OutBuf b(c, out, out_elems(c, packed) * sizeof(double), mem);
    if (b.prepare()) return 1;
    if (launch_fin_cov(c->stream, c->geom(), (const double *)c->acc_f64.p, scale, (double *)b.dev, packed)) return 1;
    if (b.commit()) return 1;
    return finish(c);
}


int snpgpu_pca_panel_trace(snpgpu_ctx *c, double *trace)
{
    if (!c || c->kind != SNPGPU_PCA_COV) { set_error("snpgpu_pca_panel_trace: needs a PCA_COV context"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(c->device));
    if (settle_colterm(c)) return 1;
    if (launch_trace(c->stream, c->geom(), (const double *)c->acc_f64.p, c->d_trace())) return 1;
    double tr = 0;
    SNPGPU_HIP_CHECK(hipMemcpyAsync(&tr, c->d_trace(), sizeof(double), hipMemcpyDeviceToHost, c->stream));
    SNPGPU_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (trace) *trace = tr;
    return 0;
}

int snpgpu_pca_panel_matmul(snpgpu_ctx *c, double scale, const double *Q, int m, double *Y)
{
    if (snpgpu::ctx_panel_matmul_enqueue(c, scale, Q, m, Y)) return 1;
    SNPGPU_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int snpgpu_pca_panel_matmul_f32(snpgpu_ctx *c, double scale, const double *Q, int m, double *Y)
{
    if (snpgpu::ctx_panel_matmul_enqueue(c, scale, Q, m, Y, true)) return 1;
    SNPGPU_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int snpgpu_finalize_inplace(snpgpu_ctx *c, int diagadj, double scale)
{
    if (!c) { set_error("snpgpu_finalize_inplace: NULL context"); return 1; }
    if (c->kind != SNPGPU_PCA_COV && c->kind != SNPGPU_GRM_GCTA && c->kind != SNPGPU_EIGMIX) {
        set_error("snpgpu_finalize_inplace: needs a PCA_COV, GRM_GCTA or EIGMIX context");
        return 1;
    }
    if (c->frozen) return 0;
    SNPGPU_HIP_CHECK(hipSetDevice(c->device));
    double *P = (double *)c->acc_f64.p;
    if (c->kind == SNPGPU_PCA_COV) return snpgpu::ctx_settle(c);    // raw sums; the (n-1)/trace factor travels with the products
    if (c->kind == SNPGPU_GRM_GCTA) {
        if (launch_fin_gcta(c->stream, c->geom(), P, (const uint32_t *)c->acc_u32.p, (const uint32_t *)c->miss_diag.p,
                            c->d_nlocus(), P, 2, c->colterm_pending ? (const double *)c->colterm.p : nullptr,
                            c->colterm_pending ? (const double *)c->uvterm.p : nullptr))
            return 1;
        c->colterm_pending = false;
        c->frozen_scale = 1.0;
    } else {
        if (check_out(c, SNPGPU_EIGMIX, SNPGPU_EIGMIX, 1, "snpgpu_finalize_inplace")) return 1;
        if (launch_fin_eigmix(c->stream, c->geom(), P, P + c->plane(), (const uint32_t *)c->samp_het.p,
                              (const double *)c->samp_dmiss.p, (const double *)c->samp_dsq.p, c->d_sumden(), diagadj,
                              scale, P, 2))
            return 1;
        c->frozen_diagadj = diagadj;
        c->frozen_scale = scale;
    }
    c->acc_f32_valid = false;
    c->frozen = true;
    SNPGPU_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

__global__ __launch_bounds__(256) void panel_entries_kernel(const double *__restrict__ P, int64_t ld, int64_t tiles_c, int64_t col0,
                                                            const int64_t *__restrict__ rows, const int64_t *__restrict__ cols, int64_t n,
                                                            double *__restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) out[k] = P[snpgpu::acc_off(ld, tiles_c, rows[k] - col0, cols[k] - col0)];
}

int snpgpu_panel_entries(snpgpu_ctx *c, const int64_t *rows, const int64_t *cols, int64_t n_entries, double *out)
{
    if (!c || !(c->kind == SNPGPU_PCA_COV || ((c->kind == SNPGPU_GRM_GCTA || c->kind == SNPGPU_EIGMIX) && c->frozen))) {
        set_error("snpgpu_panel_entries: needs a PCA_COV context, or a GRM_GCTA / EIGMIX context after snpgpu_finalize_inplace");
        return 1;
    }
    if (n_entries <= 0) return 0;
    if (!rows || !cols || !out) { set_error("snpgpu_panel_entries: invalid arguments"); return 1; }
    for (int64_t k = 0; k < n_entries; k++)
        if (rows[k] < c->row0 || rows[k] >= c->row1 || cols[k] < rows[k] || cols[k] >= c->N) {
            set_error("snpgpu_panel_entries: entry " + std::to_string(k) + " lies outside the panel's upper trapezoid");
            return 1;
        }
    SNPGPU_HIP_CHECK(hipSetDevice(c->device));
    if (settle_colterm(c)) return 1;
    DevBuf idx, res;
    if (idx.alloc(sizeof(int64_t) * 2 * (size_t)n_entries) || res.alloc(sizeof(double) * (size_t)n_entries)) { idx.release(); res.release(); return 1; }
    int rc = 0;
    do {
        if (hipMemcpyAsync(idx.p, rows, sizeof(int64_t) * (size_t)n_entries, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync((int64_t *)idx.p + n_entries, cols, sizeof(int64_t) * (size_t)n_entries, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = 1; break; }
        hipLaunchKernelGGL(panel_entries_kernel, dim3((unsigned)((n_entries + 255) / 256)), dim3(256), 0, c->stream, (const double *)c->acc_f64.p,
                           c->ncols_pad, c->acc_tiles_c, c->col0, (const int64_t *)idx.p, (const int64_t *)idx.p + n_entries, n_entries, (double *)res.p);
        if (hipMemcpyAsync(out, res.p, sizeof(double) * (size_t)n_entries, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) { rc = 1; break; }
    } while (0);
    idx.release(); res.release();
    if (rc) { set_error("snpgpu_panel_entries: copy or launch failed"); return 1; }
    return 0;
}

}  // extern "C"

// Y += scale * (this panel's part of the symmetric matrix) Q, enqueued on the context's stream
int snpgpu::ctx_panel_matmul_enqueue(snpgpu_ctx *c, double scale, const double *Q, int m, double *Y, bool fp32_products)
{
    if (!c || !(c->kind == SNPGPU_PCA_COV || ((c->kind == SNPGPU_GRM_GCTA || c->kind == SNPGPU_EIGMIX) && c->frozen))) {
        set_error("snpgpu_pca_panel_matmul: needs a PCA_COV context, or a GRM_GCTA / EIGMIX context after snpgpu_finalize_inplace");
        return 1;
    }
    if (!Q || !Y || m <= 0) { set_error("snpgpu_pca_panel_matmul: invalid arguments"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(c->device));
    if (settle_colterm(c)) return 1;
    double *P = (double *)c->acc_f64.p;      // row-major [rows_pad][ld]  ==  column-major M (ld x rows), M[j,i] = P[i,j]
    const int64_t n = c->N, r0 = c->row0, r1 = c->row1, ld = c->ncols_pad;
    if (!getenv("SNPGPU_EIG_BLAS")) {
        // one pass over the panel, every tile used for both triangles (kernels_eig.hip); below the diagonal
        // it reads only the 64 x 64 tiles on it
        if (c->diag_mirrored == 0) {
            if (launch_mirror_diag_tiles(c->stream, c->geom(), P, 64)) return 1;
            c->diag_mirrored = 1;
        }
        if (!c->eig_qt.p && c->eig_qt.alloc(sizeof(double) * 48 * (size_t)(n + 16))) return 1;
        // fp32 products stream an fp32 COPY of the (settled, mirrored) plane where the device has room for it next to 8 GiB of
        // head room: half the bytes per product, no conversions (SNPGPU_EIG_F32_PANEL=0: convert the fp64 plane on the fly)
        const float *P32 = nullptr;
        if (fp32_products && !(getenv("SNPGPU_EIG_F32_PANEL") && !atoi(getenv("SNPGPU_EIG_F32_PANEL")))) {
            const size_t elems = (size_t)c->plane();
            if (!c->acc_f32.p) {
                size_t fr = 0, tot = 0;
                if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr > elems * sizeof(float) + ((size_t)8 << 30)) {
                    if (c->acc_f32.alloc(elems * sizeof(float))) return 1;
                    c->acc_f32_valid = false;
                }
            }
            if (c->acc_f32.p && !c->acc_f32_valid) {
                if (launch_panel_to_f32(c->stream, P, (float *)c->acc_f32.p, elems)) return 1;
                c->acc_f32_valid = true;
            }
            if (c->acc_f32.p) P32 = (const float *)c->acc_f32.p;
        }
        return launch_sym_panel_matmul(c->stream, P, ld, c->acc_tiles_c, r1 - r0, n - r0, r0, n, scale, Q, m, Y, (double *)c->eig_qt.p,
                                       fp32_products, P32);
    }
    if (c->acc_tiles_c) { set_error("snpgpu_pca_panel_matmul: SNPGPU_EIG_BLAS must be set when the context is created (row-major panel)"); return 1; }
    if (!c->blas) {
        rocblas_handle hb = nullptr;
        if (rocblas_create_handle(&hb) != rocblas_status_success) { set_error("rocblas_create_handle failed"); return 1; }
        rocblas_set_stream(hb, c->stream);
        rocblas_set_pointer_mode(hb, rocblas_pointer_mode_host);
        c->blas = hb;
    }
    if (c->diag_mirrored != 2) {             // the dgemm form needs the whole diagonal square
        if (launch_mirror_diag(c->stream, c->geom(), P)) return 1;
        c->diag_mirrored = 2;
    }
    rocblas_handle h = (rocblas_handle)c->blas;
    const int64_t nI = r1 - r0, nJ = n - r0, nR = n - r1;
    const double one = 1.0;
    // Y[I] += scale * P[I, r0:N] * Q[r0:N]        (P = M^T)
    rocblas_status st = rocblas_dgemm(h, rocblas_operation_transpose, rocblas_operation_none, (rocblas_int)nI, m,
                                      (rocblas_int)nJ, &scale, P, (rocblas_int)ld, Q + r0, (rocblas_int)n, &one,
                                      Y + r0, (rocblas_int)n);
    if (st != rocblas_status_success) { set_error("rocblas_dgemm (panel rows) failed"); return 1; }
    if (nR > 0) {
        // Y[r1:N] += scale * P[I, r1:N]^T * Q[I]   (= M[r1-r0 : , :] * Q[I])
        st = rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, (rocblas_int)nR, m, (rocblas_int)nI, &scale,
                           P + nI, (rocblas_int)ld, Q + r0, (rocblas_int)n, &one, Y + r1, (rocblas_int)n);
        if (st != rocblas_status_success) { set_error("rocblas_dgemm (panel columns) failed"); return 1; }
    }
    return 0;
}

extern "C" {

int snpgpu_ibd_mom(snpgpu_ctx *c, const double *e, int kinship_constraint, double *k0, double *k1, int packed, int mem)
{
    if (check_out(c, SNPGPU_IBS, SNPGPU_IBS, packed, "snpgpu_ibd_mom")) return 1;
    if (!e) { set_error("snpgpu_ibd_mom: e is NULL"); return 1; }
    const size_t n = out_elems(c, packed) * sizeof(double);
    OutBuf b0(c, k0, n, mem), b1(c, k1, n, mem);
    if (b0.prepare() || b1.prepare()) return 1;
    if (launch_fin_mom(c->stream, c->geom(), (const uint32_t *)c->acc_u32.p, e, kinship_constraint, (double *)b0.dev,
                       (double *)b1.dev, packed))
        return 1;
    if (b0.commit() || b1.commit()) return 1;
    return finish(c);
}

int snpgpu_eigmix(snpgpu_ctx *c, int diagadj, double scale, double *out, int packed, int mem)
{
    if (check_out(c, SNPGPU_EIGMIX, SNPGPU_EIGMIX, packed, "snpgpu_eigmix")) return 1;
    OutBuf b(c, out, out_elems(c, packed) * sizeof(double), mem);
    if (b.prepare()) return 1;
    const double *num = (const double *)c->acc_f64.p;
    if (c->frozen) {
        if ((diagadj != 0) != (c->frozen_diagadj != 0)) { set_error("snpgpu_eigmix: the context was finalised in place with another 'diagadj'"); return 1; }
        if (launch_fin_cov(c->stream, c->geom(), num, scale / c->frozen_scale, (double *)b.dev, packed)) return 1;
    } else if (launch_fin_eigmix(c->stream, c->geom(), num, num + c->plane(), (const uint32_t *)c->samp_het.p,
                          (const double *)c->samp_dmiss.p, (const double *)c->samp_dsq.p, c->d_sumden(), diagadj, scale,
                          (double *)b.dev, packed))
        return 1;
    if (b.commit()) return 1;
    return finish(c);
}

int snpgpu_indiv_beta(snpgpu_ctx *c, int mode, double *out, double *avg_val, int packed, int mem)
{
    if (check_out(c, SNPGPU_INDIV_BETA, SNPGPU_INDIV_BETA, packed, "snpgpu_indiv_beta")) return 1;
    if (!c->full) { set_error("snpgpu_indiv_beta: needs a full (non-panel) context"); return 1; }
    if (mode < 0 || mode > 2) { set_error("snpgpu_indiv_beta: invalid mode"); return 1; }
    const int nb = 1024;
    DevBuf part;
    if (part.alloc(sizeof(double) * 2 * nb)) return 1;
    std::vector<double> h(2 * nb);
    int rc = launch_beta_reduce(c->stream, c->geom(), (const uint32_t *)c->acc_u32.p, mode != 0, (double *)part.p,
                                (double *)part.p + nb, nb);
    if (!rc && hipMemcpyAsync(h.data(), part.p, sizeof(double) * 2 * nb, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = 1;
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = 1;
    part.release();
    if (rc) { set_error("snpgpu_indiv_beta: reduction failed"); return 1; }
    double mn = h[0], sum = 0;
    for (int i = 0; i < nb; i++) { if (h[i] < mn) mn = h[i]; sum += h[nb + i]; }
    const double avg = sum / (double)(c->N * (c->N - 1) / 2);
    if (avg_val) *avg_val = avg;
    if (!out) return 0;
    OutBuf b(c, out, out_elems(c, packed) * sizeof(double), mem);
    if (b.prepare()) return 1;
    if (launch_fin_beta(c->stream, c->geom(), (const uint32_t *)c->acc_u32.p, mode, avg, mn, (double *)b.dev, packed)) return 1;
    if (b.commit()) return 1;
    return finish(c);
}

}  // extern "C"
