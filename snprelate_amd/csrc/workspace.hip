// C ABI of libsnpgpu, level (2): host-side mirrors of the reference's registered `.Call`
// routines for this path (src/SNPRelate.cpp:1154-1205) over an in-memory genotype matrix,
// plus the top-k eigen solver behind gnrPCA.
//
// The reference keeps an implicit process-global working space (GWAS::MCWorkingGeno,
// src/dGenGWAS.cpp:2000) that gnrSetGenoSpace / gnrSelSNP_Base set and the compute calls
// consume; g_ws below plays that role.  In an R deployment the GDS-backed reader is kept and
// feeds level (1) directly; this layer exists for hosts without gdsfmt.
#include <hipsolver/hipsolver.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "snpgpu_internal.h"

using namespace snpgpu;

namespace {

struct WorkSpace {
    bool set = false;
    int device = 0;
    int64_t n_snp_all = 0, n_samp = 0, rb = 0;  // rb = ceil(n_samp/4) bytes per SNP row
    std::vector<uint8_t> packed;                // [n_snp_all][rb] 2-bit rows
    std::vector<int64_t> sel;                   // indices of the currently selected SNPs
};
WorkSpace g_ws;
double g_grm_avg_value = 0;   // grm_avg_value, src/genPCA.cpp:1605

constexpr int64_t WS_BLOCK = 8192;  // SNPs per feed; plays the role of the reference's cache-sized block

int need_ws(const char *fn)
{
    if (!g_ws.set) { set_error(std::string(fn) + ": no genotype working space (call snpgpu_ws_set_geno first)"); return 1; }
    return 0;
}

// gather the selected SNP rows [i0,i1) into a contiguous 2-bit block
void gather_block(int64_t i0, int64_t i1, std::vector<uint8_t> &buf)
{
    buf.resize((size_t)(i1 - i0) * (size_t)g_ws.rb);
    for (int64_t i = i0; i < i1; i++)
        memcpy(&buf[(size_t)(i - i0) * g_ws.rb], &g_ws.packed[(size_t)g_ws.sel[i] * g_ws.rb], (size_t)g_ws.rb);
}

// run the block loop of CXxx::Run over the selected SNPs
int run_stream(int kind, int bayesian, snpgpu_ctx **out)
{
    snpgpu_opts o{};
    o.device = g_ws.device;
    o.bayesian = bayesian;
    o.max_block_snps = WS_BLOCK;
    snpgpu_ctx *c = nullptr;
    if (snpgpu_create(kind, g_ws.n_samp, &o, &c)) return 1;
    std::vector<uint8_t> buf;
    const int64_t L = (int64_t)g_ws.sel.size();
    for (int64_t i0 = 0; i0 < L; i0 += WS_BLOCK) {
        const int64_t i1 = std::min(L, i0 + WS_BLOCK);
        gather_block(i0, i1, buf);
        if (snpgpu_feed(c, buf.data(), i1 - i0, SNPGPU_GENO_PACKED2, SNPGPU_HOST)) {
            snpgpu_destroy(c);
            return 1;
        }
    }
    *out = c;
    return 0;
}

struct CtxGuard {
    snpgpu_ctx *c = nullptr;
    ~CtxGuard() { if (c) snpgpu_destroy(c); }
};

// per-SNP sum / num over the selected SNPs, on the device (launch_snp_stats)
int ws_stats(std::vector<int32_t> &sum, std::vector<int32_t> &num, std::vector<int32_t> *het = nullptr)
{
    const int64_t L = (int64_t)g_ws.sel.size();
    sum.assign((size_t)L, 0);
    num.assign((size_t)L, 0);
    if (het) het->assign((size_t)L, 0);
    if (L == 0) return 0;
    SNPGPU_HIP_CHECK(hipSetDevice(g_ws.device));
    const int64_t N = g_ws.n_samp, RB = (N + 255) / 256 * 64;
    DevBuf raw, packed, dsum, dnum, dhet, dmiss;
    int rc = raw.alloc((size_t)WS_BLOCK * g_ws.rb) | packed.alloc((size_t)WS_BLOCK * RB) |
             dsum.alloc(sizeof(int32_t) * WS_BLOCK) | dnum.alloc(sizeof(int32_t) * WS_BLOCK) |
             dhet.alloc(sizeof(int32_t) * WS_BLOCK) | dmiss.alloc(8);
    std::vector<uint8_t> buf;
    for (int64_t i0 = 0; i0 < L && !rc; i0 += WS_BLOCK) {
        const int64_t i1 = std::min(L, i0 + WS_BLOCK), nb = i1 - i0;
        gather_block(i0, i1, buf);
        hipError_t e = hipMemcpy(raw.p, buf.data(), buf.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(dmiss.p, 0, 8);
        if (e != hipSuccess) { set_error(std::string("ws_stats: ") + hipGetErrorString(e)); rc = 1; break; }
        rc |= launch_repack(nullptr, raw.p, SNPGPU_GENO_PACKED2, nb, N, (uint8_t *)packed.p, RB);
        rc |= launch_snp_stats(nullptr, (const uint8_t *)packed.p, RB, nb, N, (int32_t *)dsum.p, (int32_t *)dnum.p,
                               (unsigned long long *)dmiss.p, (int32_t *)dhet.p);
        if (rc) break;
        e = hipMemcpy(&sum[i0], dsum.p, sizeof(int32_t) * nb, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(&num[i0], dnum.p, sizeof(int32_t) * nb, hipMemcpyDeviceToHost);
        if (e == hipSuccess && het) e = hipMemcpy(&(*het)[i0], dhet.p, sizeof(int32_t) * nb, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_error(std::string("ws_stats: ") + hipGetErrorString(e)); rc = 1; }
    }
    raw.release(); packed.release(); dsum.release(); dnum.release(); dhet.release(); dmiss.release();
    return rc;
}

__global__ void negate_kernel(double *a, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = -a[i];
}

// randomised PCA plumbing (all fp64)
__global__ void rp_scale_kernel(double *a, size_t n, double f)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] *= f;
}
// block result [nb][A] -> scaled in place and scattered into rows row0 .. row0+A-1 of Ht [hsize][L]
__global__ void rp_scatter_kernel(double *blk, int64_t nb, int A, double f, double *Ht, int64_t L, int64_t snp0, int row0)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nb * A) return;
    const int64_t s = idx / A;
    const int j = (int)(idx - s * A);
    const double v = blk[idx] * f;
    blk[idx] = v;
    Ht[(int64_t)(row0 + j) * L + snp0 + s] = v;
}
// rows snp0 .. snp0+nb-1 of Q (column-major L x hs) -> contiguous [nb][hs]
__global__ void rp_gather_kernel(const double *Q, int64_t L, int64_t snp0, int64_t nb, int hs, double *out)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nb * hs) return;
    const int64_t s = idx / hs;
    const int r = (int)(idx - s * hs);
    out[idx] = Q[(int64_t)r * L + snp0 + s];
}
// column-major m x n -> its transpose (column-major n x m)
__global__ void rp_transpose_kernel(const double *a, int64_t m, int64_t n, double *out)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= m * n) return;
    const int64_t i = idx % m, j = idx / m;
    out[i * n + j] = a[idx];
}

}  // namespace

extern "C" {

int snpgpu_ws_set_geno(const void *geno, int64_t n_snp, int64_t n_samp, int format, int device)
{
    if (!geno || n_snp <= 0 || n_samp <= 0) { set_error("snpgpu_ws_set_geno: invalid arguments"); return 1; }
    if (format != SNPGPU_GENO_U8 && format != SNPGPU_GENO_PACKED2) { set_error("snpgpu_ws_set_geno: invalid format"); return 1; }
    int ndev = 0;
    SNPGPU_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) { set_error("snpgpu_ws_set_geno: invalid device ordinal"); return 1; }
    g_ws.set = false;
    g_ws.device = device;
    g_ws.n_snp_all = n_snp; g_ws.n_samp = n_samp; g_ws.rb = (n_samp + 3) / 4;
    g_ws.packed.assign((size_t)n_snp * g_ws.rb, 0);
    const uint8_t *src = (const uint8_t *)geno;
    if (format == SNPGPU_GENO_PACKED2) {
        memcpy(g_ws.packed.data(), src, g_ws.packed.size());
    } else {
        for (int64_t l = 0; l < n_snp; l++) {
            uint8_t *row = &g_ws.packed[(size_t)l * g_ws.rb];
            const uint8_t *g = src + (size_t)l * n_samp;
            for (int64_t i = 0; i < n_samp; i++) {
                unsigned v = g[i] > 3 ? 3u : g[i];
                row[i >> 2] |= (uint8_t)(v << (2 * (i & 3)));
            }
        }
    }
    const int tail = (int)(g_ws.rb * 4 - n_samp);  // padding codes are stored as missing
    if (tail)
        for (int64_t l = 0; l < n_snp; l++) g_ws.packed[(size_t)(l + 1) * g_ws.rb - 1] |= (uint8_t)(0xFF << (2 * (4 - tail)));
    g_ws.sel.resize((size_t)n_snp);
    for (int64_t l = 0; l < n_snp; l++) g_ws.sel[(size_t)l] = l;
    g_ws.set = true;
    return 0;
}

int snpgpu_ws_clear(void)
{
    g_ws = WorkSpace();
    return 0;
}

int snpgpu_ws_get_geno_dim(int64_t *n_snp, int64_t *n_samp)
{
    if (need_ws("snpgpu_ws_get_geno_dim")) return 1;
    if (n_snp) *n_snp = (int64_t)g_ws.sel.size();
    if (n_samp) *n_samp = g_ws.n_samp;
    return 0;
}

// Get_AF_MR_perSNP, src/dGenGWAS.cpp:472-552 (RDim_Sample_X_SNP branch)
int snpgpu_ws_snp_rate_freq(double *af, double *maf, double *missrate)
{
    if (need_ws("snpgpu_ws_snp_rate_freq")) return 1;
    std::vector<int32_t> sum, num;
    if (ws_stats(sum, num)) return 1;
    const double nan = std::numeric_limits<double>::quiet_NaN();
    for (size_t l = 0; l < sum.size(); l++) {
        const double F = (num[l] > 0) ? ((double)sum[l] / (2 * num[l])) : nan;
        if (af) af[l] = F;
        if (maf) maf[l] = std::isnan(F) ? nan : std::min(F, 1 - F);
        if (missrate) missrate[l] = 1 - ((double)num[l]) / (double)g_ws.n_samp;
    }
    return 0;
}

// CdBaseWorkSpace::Select_SNP_Base, src/dGenGWAS.cpp:361-397
int snpgpu_ws_sel_snp_base(int remove_mono, double maf, double missrate, int32_t *n_excluded, uint8_t *sel_out)
{
    if (need_ws("snpgpu_ws_sel_snp_base")) return 1;
    std::vector<int32_t> sum, num;
    if (ws_stats(sum, num)) return 1;
    std::vector<int64_t> keep;
    int32_t excluded = 0;
    for (size_t l = 0; l < sum.size(); l++) {
        bool flag;
        if (num[l] > 0) {
            const double F = (double)sum[l] / (2 * num[l]);
            const double MAF = std::min(F, 1 - F);
            const double MR = 1 - ((double)num[l]) / (double)g_ws.n_samp;
            flag = true;
            if (remove_mono && MAF <= 0) flag = false;
            if (flag && MAF < maf) flag = false;
            if (flag && MR > missrate) flag = false;
        } else
            flag = false;
        if (sel_out) sel_out[l] = flag ? 1 : 0;
        if (flag) keep.push_back(g_ws.sel[l]); else excluded++;
    }
    g_ws.sel.swap(keep);
    if (n_excluded) *n_excluded = excluded;
    return 0;
}

// gnrSelSNP_Base_Ex(afreq, remove_mono, maf, missrate), src/SNPRelate.cpp:215-239 -> Select_SNP_Base_Ex,
// src/dGenGWAS.cpp:399-469: the monomorphic / MAF tests use the CALLER's allele frequencies (a non-finite one drops
// the SNP), only the missing rate comes from the genotypes.
int snpgpu_ws_sel_snp_base_ex(const double *afreq, int remove_mono, double maf, double missrate, int32_t *n_excluded,
                              uint8_t *sel_out)
{
    if (need_ws("snpgpu_ws_sel_snp_base_ex")) return 1;
    if (!afreq) { set_error("snpgpu_ws_sel_snp_base_ex: afreq is NULL"); return 1; }
    std::vector<int32_t> sum, num;
    if (ws_stats(sum, num)) return 1;
    std::vector<int64_t> keep;
    int32_t excluded = 0;
    for (size_t l = 0; l < num.size(); l++) {
        bool flag = true;
        if (std::isfinite(afreq[l])) {
            const double MF = std::min(afreq[l], 1 - afreq[l]);
            const double MR = 1.0 - (double)num[l] / (double)g_ws.n_samp;
            if (remove_mono && MF <= 0) flag = false;
            if (flag && MF < maf) flag = false;
            if (flag && MR > missrate) flag = false;
        } else
            flag = false;
        if (sel_out) sel_out[l] = flag ? 1 : 0;
        if (flag) keep.push_back(g_ws.sel[l]); else excluded++;
    }
    g_ws.sel.swap(keep);
    if (n_excluded) *n_excluded = excluded;
    return 0;
}

int snpgpu_gnrIBSNum(int, int, int32_t *ibs0, int32_t *ibs1, int32_t *ibs2)
{
    if (need_ws("snpgpu_gnrIBSNum")) return 1;
    CtxGuard g;
    if (run_stream(SNPGPU_IBS, 0, &g.c)) return 1;
    return snpgpu_ibs_num(g.c, ibs0, ibs1, ibs2, 0, SNPGPU_HOST);
}

int snpgpu_gnrIBSAve(int, int use_matrix, int, double *out)
{
    if (need_ws("snpgpu_gnrIBSAve")) return 1;
    CtxGuard g;
    if (run_stream(SNPGPU_IBS, 0, &g.c)) return 1;
    return snpgpu_ibs_ave(g.c, out, use_matrix ? 1 : 0, SNPGPU_HOST);
}

int snpgpu_gnrIBD_KING_Robust(const int32_t *family, int, int use_matrix, int, double *ibs0, double *kinship)
{
    if (need_ws("snpgpu_gnrIBD_KING_Robust")) return 1;
    if ((int64_t)g_ws.sel.size() >= 1073741824LL) {  // src/genKING.cpp:598-602
        set_error("The number of SNPs should be less than 1,073,741,824.");
        return 1;
    }
    CtxGuard g;
    if (run_stream(SNPGPU_KING_ROBUST, 0, &g.c)) return 1;
    return snpgpu_king_robust(g.c, family, ibs0, kinship, use_matrix ? 1 : 0, SNPGPU_HOST);
}

int snpgpu_gnrIBD_KING_Homo(int, int use_matrix, int, double *k0, double *k1)
{
    if (need_ws("snpgpu_gnrIBD_KING_Homo")) return 1;
    CtxGuard g;
    if (run_stream(SNPGPU_KING_HOMO, 0, &g.c)) return 1;
    return snpgpu_king_homo(g.c, k0, k1, use_matrix ? 1 : 0, SNPGPU_HOST);
}

// gnrGRM method switch, src/genPCA.cpp:1633-1710
int snpgpu_gnrGRM(int, const char *method, int use_matrix, int, double *out)
{
    if (need_ws("snpgpu_gnrGRM")) return 1;
    if (!method) { set_error("Invalid 'method'!"); return 1; }
    const int64_t n = g_ws.n_samp;
    CtxGuard g;
    if (strcmp(method, "Eigenstrat") == 0) {
        if (run_stream(SNPGPU_PCA_COV, 0, &g.c)) return 1;
        return snpgpu_pca_cov(g.c, out, use_matrix ? 1 : 0, 1, 0.0, nullptr, SNPGPU_HOST);
    } else if (strcmp(method, "GCTA") == 0) {
        if (run_stream(SNPGPU_GRM_GCTA, 0, &g.c)) return 1;
        return snpgpu_grm_gcta(g.c, out, use_matrix ? 1 : 0, SNPGPU_HOST);
    } else if (strcmp(method, "Corr") == 0) {
        // always a full matrix, src/genPCA.cpp:1658-1685
        if (run_stream(SNPGPU_GRM_GCTA, 0, &g.c)) return 1;
        if (snpgpu_grm_gcta(g.c, out, 0, SNPGPU_HOST)) return 1;
        std::vector<double> diag((size_t)n);
        for (int64_t i = 0; i < n; i++) diag[(size_t)i] = sqrt(out[i + n * i]);
        for (int64_t i = 0; i < n; i++) {
            out[i + n * i] = 1;
            for (int64_t j = i + 1; j < n; j++)
                out[i + n * j] = out[j + n * i] = out[j + n * i] / (diag[(size_t)i] * diag[(size_t)j]);
        }
        return 0;
    }
    else if (strcmp(method, "EIGMIX") == 0) {      // CalcEigMixGRM: no diagonal adjustment, times 2
        if (run_stream(SNPGPU_EIGMIX, 0, &g.c)) return 1;
        return snpgpu_eigmix(g.c, 0, 2.0, out, use_matrix ? 1 : 0, SNPGPU_HOST);
    } else if (strcmp(method, "IndivBeta") == 0) {  // CalcIndivBetaGRM
        if (run_stream(SNPGPU_INDIV_BETA, 0, &g.c)) return 1;
        double avg = 0;
        if (snpgpu_indiv_beta(g.c, 2, out, &avg, use_matrix ? 1 : 0, SNPGPU_HOST)) return 1;
        g_grm_avg_value = avg;
        return 0;
    }
    set_error("Invalid 'method'!");  // src/genPCA.cpp:1710
    return 1;
}

// top-k eigenpairs of a dense symmetric matrix A (device, n x n, overwritten): the reference negates
// the packed matrix and asks LAPACK dspevx for eigenvalues IL=1..IU=k (src/genPCA.cpp:1308-1341,
// :1419; src/genEIGMIX.cpp:700-702); here A is negated on the device and hipSOLVER's syevdx is asked
// for the same index range.  eigval: k values (descending), eigvec: n x k column-major.
static int dense_topk(int device, hipStream_t stream, double *A, int64_t n, int k, double *eigval, double *eigvec, int mem)
{
    if (k <= 0 || k > n) { set_error("Invalid 'eigen.cnt'."); return 1; }
    if (n > 46340) { set_error("dense eigen solver limited to n <= 46340 (larger n take the block-Krylov solver, csrc/eigen.hip)"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(device));
    DevBuf W, work, info;
    int rc = W.alloc(sizeof(double) * (size_t)n) | info.alloc(sizeof(int));
    hipsolverHandle_t h = nullptr;
    do {
        if (rc) break;
        const size_t nn = (size_t)n * (size_t)n;
        hipLaunchKernelGGL(negate_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, stream, A, nn);
        if (hipStreamSynchronize(stream) != hipSuccess) { set_error("dense_topk: negate failed"); rc = 1; break; }
        if (hipsolverCreate(&h) != HIPSOLVER_STATUS_SUCCESS) { set_error("hipsolverCreate failed"); rc = 1; break; }
        hipsolverSetStream(h, stream);
        int lwork = 0, nev = 0;
        if (hipsolverDnDsyevdx_bufferSize(h, HIPSOLVER_EIG_MODE_VECTOR, HIPSOLVER_EIG_RANGE_I, HIPBLAS_FILL_MODE_LOWER,
                                          (int)n, A, (int)n, 0.0, 0.0, 1, k, &nev, (double *)W.p,
                                          &lwork) != HIPSOLVER_STATUS_SUCCESS) {
            // (seen at n = 24 000: the workspace size does not fit the interface) -- the callers fall back to the block-Krylov solver
            set_error("hipsolverDnDsyevdx_bufferSize failed"); rc = 2; break;
        }
        if (work.alloc(sizeof(double) * (size_t)std::max(lwork, 1))) { rc = 1; break; }
        hipsolverStatus_t s = hipsolverDnDsyevdx(h, HIPSOLVER_EIG_MODE_VECTOR, HIPSOLVER_EIG_RANGE_I,
                                                 HIPBLAS_FILL_MODE_LOWER, (int)n, A, (int)n, 0.0, 0.0, 1, k,
                                                 &nev, (double *)W.p, (double *)work.p, lwork, (int *)info.p);
        int hinfo = 0;
        hipError_t e = hipMemcpyAsync(&hinfo, info.p, sizeof(int), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (s != HIPSOLVER_STATUS_SUCCESS || e != hipSuccess || hinfo != 0) {
            // message of src/genPCA.cpp:1333
            set_error("LAPACK::DSPEVX error (" + std::to_string(hinfo) +
                      "), infinite or missing values in the genetic covariance matrix!");
            rc = 1; break;
        }
        std::vector<double> w((size_t)k);
        if (hipMemcpy(w.data(), W.p, sizeof(double) * (size_t)k, hipMemcpyDeviceToHost) != hipSuccess) { set_error("eigen copy failed"); rc = 1; break; }
        for (int i = 0; i < k; i++) w[(size_t)i] = -w[(size_t)i];
        const hipMemcpyKind kind = (mem == SNPGPU_DEVICE) ? hipMemcpyHostToDevice : hipMemcpyHostToHost;
        if (eigval && hipMemcpy(eigval, w.data(), sizeof(double) * (size_t)k, kind) != hipSuccess) { set_error("eigen copy failed"); rc = 1; break; }
        const hipMemcpyKind kv = (mem == SNPGPU_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        if (eigvec && hipMemcpy(eigvec, A, sizeof(double) * (size_t)n * (size_t)k, kv) != hipSuccess) { set_error("eigen copy failed"); rc = 1; break; }
    } while (0);
    if (h) hipsolverDestroy(h);
    W.release(); work.release(); info.release();
    return rc;
}

// samples up to which the dense solver is used (the reference's own route: exact to LAPACK's tolerance, O(n^3)); beyond it
// the block-Krylov solver works on the resident panel, without an n x n copy.  hipSOLVER's syevdx takes 4 s at n = 4096, 21 s at
// 8192, 124 s at 16 384 and 210 s at 20 000 on an MI355X (tools/scratch measurement, round 3) against seconds for the Krylov
// solver: the dense route is kept for small n only.
static int64_t dense_eigen_max()
{
    if (const char *e = getenv("SNPGPU_EIG_DENSE_MAX")) { const long long v = atoll(e); if (v >= 0 && v <= 46340) return v; }
    return 2048;
}

// A request for a large share of the spectrum (eigen.cnt <= 0 = all, eigen.method = "DSPEV", or k of the order of n) is not a
// top-k problem: the Krylov solver's block would be k + 8 vectors wide with room for one or two blocks per cycle (n = 3000,
// k = 1000: one new block per restart), and with k = n it degenerates to a full QR + syevd inside six n x n work arrays.  Up to
// 16 384 samples (124 s of syevdx) such requests take the dense route whatever SNPGPU_EIG_DENSE_MAX says.
static bool dense_route(int64_t n, int k)
{
    if (n <= dense_eigen_max()) return true;
    return n <= 16384 && (int64_t)k * 8 > n;
}

// the (n - 1) / trace factor of the iterative route, checked as LAPACK would report such a matrix (src/genPCA.cpp:1333)
static int krylov_scale(snpgpu_ctx *c, double *scale)
{
    double tr = 0;
    if (snpgpu_pca_panel_trace(c, &tr)) return 1;
    if (!(tr > 0) || !std::isfinite(tr)) {
        snpgpu::set_error("LAPACK::DSPEVX error (-1), infinite or missing values in the genetic covariance matrix!");
        return 1;
    }
    *scale = (double)(c->N - 1) / tr;
    return 0;
}

int snpgpu_pca_eigen(snpgpu_ctx *c, int k, double *eigval, double *eigvec, int mem)
{
    if (!c || c->kind != SNPGPU_PCA_COV || !c->full) { set_error("snpgpu_pca_eigen: needs a full PCA_COV context"); return 1; }
    const int64_t n = c->N;
    if (k <= 0 || k > n) { set_error("Invalid 'eigen.cnt'."); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(c->device));
    if (!dense_route(n, k)) {
        double scale = 0;
        if (krylov_scale(c, &scale)) return 1;
        snpgpu_ctx *panels[1] = {c};
        return snpgpu_panels_topk_eigen(panels, 1, scale, k, nullptr, eigval, eigvec, mem, nullptr);
    }
    DevBuf A;
    if (A.alloc(sizeof(double) * (size_t)n * (size_t)n)) return 1;
    int rc = snpgpu_pca_cov(c, (double *)A.p, 0, 1, 0.0, nullptr, SNPGPU_DEVICE);
    if (!rc) rc = dense_topk(c->device, c->stream, (double *)A.p, n, k, eigval, eigvec, mem);
    A.release();
    if (rc == 2) {              // the dense solver declined this size: block Krylov on the resident panel
        double scale = 0;
        set_error("");          // (the declined workspace query is not this call's outcome)
        if (krylov_scale(c, &scale)) return 1;
        snpgpu_ctx *panels[1] = {c};
        return snpgpu_panels_topk_eigen(panels, 1, scale, k, nullptr, eigval, eigvec, mem, nullptr);
    }
    return rc;
}


// ---- gnrGRMMerge, src/genPCA.cpp:1721-1853 ---------------------------------------------------------------------
// The reference streams row i of every input GDS file, combines and appends; here the merged matrix lives on the
// device and the inputs travel in row slabs.
namespace {

constexpr size_t MERGE_SLAB_BYTES = (size_t)256 << 20;

__device__ inline void atomic_min_f64(double *addr, double v)
{
    unsigned long long *a = (unsigned long long *)addr, old = *a;
    while (__longlong_as_double((long long)old) > v) {
        const unsigned long long prev = atomicCAS(a, old, (unsigned long long)__double_as_longlong(v));
        if (prev == old) break;
        old = prev;
    }
}

__device__ inline double block_sum(double v, double *sh)
{
    for (int o = 32; o; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0) for (unsigned w = 0; w < blockDim.x / 64; w++) t += sh[w];
    __syncthreads();
    return t;   // valid on thread 0
}

__device__ inline double block_min(double v, double *sh)
{
    for (int o = 32; o; o >>= 1) v = fmin(v, __shfl_down(v, o));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = v;
    if (threadIdx.x == 0) for (unsigned w = 0; w < blockDim.x / 64; w++) t = fmin(t, sh[w]);
    __syncthreads();
    return t;
}

// out[row0 .. row0+rows) += w * in                                                    (vec_f64_addmul, :1846)
__global__ void merge_axpy_kernel(double *out, const double *in, size_t n, double w)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
        out[e] += w * in[e];
}

// sum of the off-diagonal entries of a row slab                                                   (:1764-1772)
__global__ void merge_offdiag_sum_kernel(const double *in, int64_t row0, int64_t rows, int64_t N, double *sum)
{
    __shared__ double sh[8];
    double s = 0;
    const size_t n = (size_t)rows * (size_t)N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int64_t i = row0 + (int64_t)(e / (size_t)N), j = (int64_t)(e % (size_t)N);
        if (i != j) s += in[e];
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) unsafeAtomicAdd(sum, s);
}

// back-transform one file's slab to the M_ij scale and accumulate                                 (:1783-1793)
__global__ void merge_beta_acc_kernel(double *out, const double *in, int64_t row0, int64_t rows, int64_t N, double Mb,
                                      double Mb_inv, double avg, double w)
{
    const size_t n = (size_t)rows * (size_t)N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int64_t i = row0 + (int64_t)(e / (size_t)N), j = (int64_t)(e % (size_t)N);
        const double b = in[e];
        const double m = (j != i) ? (b * 0.5 - Mb) * Mb_inv * (1 - avg) + avg : (b - 1 - Mb) * Mb_inv * (1 - avg) + avg;
        out[e] += m * w;
    }
}

// off-diagonal sum and overall minimum of the merged M                                            (:1798-1808)
__global__ void merge_beta_stats_kernel(const double *out, int64_t N, double *sum, double *mn)
{
    __shared__ double sh[8];
    double s = 0, m = INFINITY;
    const size_t n = (size_t)N * (size_t)N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const double v = out[e];
        if ((int64_t)(e / (size_t)N) != (int64_t)(e % (size_t)N)) s += v;
        m = fmin(m, v);      // fmin drops NaN like the reference's `min > p[j]` comparison
    }
    s = block_sum(s, sh);
    m = block_min(m, sh);
    if (threadIdx.x == 0) { unsafeAtomicAdd(sum, s); atomic_min_f64(mn, m); }
}

// (M - min) * 2/(1-min); diagonal * 0.5 + 1                                                       (:1810-1819)
__global__ void merge_beta_final_kernel(double *out, int64_t N, double mn, double scale)
{
    const size_t n = (size_t)N * (size_t)N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        double v = (out[e] - mn) * scale;
        if ((int64_t)(e / (size_t)N) == (int64_t)(e % (size_t)N)) v = v * 0.5 + 1;
        out[e] = v;
    }
}

inline unsigned merge_grid(size_t n) { return (unsigned)std::min<size_t>((n + 255) / 256, 16384); }

}  // namespace

int snpgpu_gnrGRMMerge(int n_grm, int64_t N, const double *const *grm, const char *cmd, const double *avg_val,
                       const double *weight, double *out, int device)
{
    if (n_grm <= 0 || N <= 0 || !grm || !weight || !out) { set_error("snpgpu_gnrGRMMerge: invalid arguments"); return 1; }
    for (int k = 0; k < n_grm; k++)
        if (!grm[k]) { set_error("snpgpu_gnrGRMMerge: invalid arguments"); return 1; }
    const bool beta = cmd && strcmp(cmd, ":method = IndivBeta") == 0;       // src/genPCA.cpp:1744
    if (beta && !avg_val) { set_error("snpgpu_gnrGRMMerge: 'avg_val' of every input is needed for IndivBeta"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(device));
    const size_t nn = (size_t)N * (size_t)N;
    const int64_t slab_rows = std::max<int64_t>(1, std::min<int64_t>(N, (int64_t)(MERGE_SLAB_BYTES / (sizeof(double) * (size_t)N))));
    DevBuf M, slab, red;
    int rc = M.alloc(sizeof(double) * nn) | slab.alloc(sizeof(double) * (size_t)slab_rows * (size_t)N) | red.alloc(2 * sizeof(double));
    hipStream_t st = nullptr;
    do {
        if (rc) break;
        rc = 1;
        if (hipMemsetAsync(M.p, 0, sizeof(double) * nn, st) != hipSuccess) { set_error("snpgpu_gnrGRMMerge: memset failed"); break; }
        std::vector<double> Mb((size_t)n_grm, 0.0), Mb_inv((size_t)n_grm, 1.0);
        bool ok = true;
        for (int pass = beta ? 0 : 1; pass < 2 && ok; pass++) {
            for (int k = 0; k < n_grm && ok; k++) {
                if (pass == 0) ok = hipMemsetAsync(red.p, 0, sizeof(double), st) == hipSuccess;
                for (int64_t r0 = 0; r0 < N && ok; r0 += slab_rows) {
                    const int64_t rows = std::min(slab_rows, N - r0);
                    const size_t cnt = (size_t)rows * (size_t)N;
                    ok = hipMemcpyAsync(slab.p, grm[k] + (size_t)r0 * (size_t)N, sizeof(double) * cnt, hipMemcpyHostToDevice, st) == hipSuccess;
                    if (!ok) break;
                    double *dst = (double *)M.p + (size_t)r0 * (size_t)N;
                    if (pass == 0)
                        hipLaunchKernelGGL(merge_offdiag_sum_kernel, dim3(merge_grid(cnt)), dim3(256), 0, st,
                                           (const double *)slab.p, r0, rows, N, (double *)red.p);
                    else if (beta)
                        hipLaunchKernelGGL(merge_beta_acc_kernel, dim3(merge_grid(cnt)), dim3(256), 0, st, dst,
                                           (const double *)slab.p, r0, rows, N, Mb[(size_t)k], Mb_inv[(size_t)k], avg_val[k],
                                           weight[k]);
                    else
                        hipLaunchKernelGGL(merge_axpy_kernel, dim3(merge_grid(cnt)), dim3(256), 0, st, dst,
                                           (const double *)slab.p, cnt, weight[k]);
                    ok = hipStreamSynchronize(st) == hipSuccess;    // the slab buffer is reused
                }
                if (pass == 0 && ok) {
                    double s = 0;
                    ok = hipMemcpy(&s, red.p, sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
                    Mb[(size_t)k] = s / ((double)N * (double)(N - 1)) * 0.5;            // :1773
                    Mb_inv[(size_t)k] = 1 / (1 - Mb[(size_t)k]);
                }
            }
        }
        if (!ok) { set_error("snpgpu_gnrGRMMerge: device transfer or kernel failed"); break; }
        if (beta) {
            const double init[2] = {0.0, std::numeric_limits<double>::infinity()};
            if (hipMemcpy(red.p, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) { set_error("snpgpu_gnrGRMMerge: copy failed"); break; }
            hipLaunchKernelGGL(merge_beta_stats_kernel, dim3(merge_grid(nn)), dim3(256), 0, st, (const double *)M.p, N,
                               (double *)red.p, (double *)red.p + 1);
            double r[2];
            if (hipMemcpy(r, red.p, sizeof(r), hipMemcpyDeviceToHost) != hipSuccess) { set_error("snpgpu_gnrGRMMerge: copy failed"); break; }
            g_grm_avg_value = r[0] / ((double)N * (double)(N - 1));                    // :1809
            hipLaunchKernelGGL(merge_beta_final_kernel, dim3(merge_grid(nn)), dim3(256), 0, st, (double *)M.p, N, r[1],
                               2 / (1 - r[1]));
        }
        if (hipMemcpy(out, M.p, sizeof(double) * nn, hipMemcpyDeviceToHost) != hipSuccess) { set_error("snpgpu_gnrGRMMerge: copy failed"); break; }
        rc = 0;
    } while (0);
    M.release(); slab.release(); red.release();
    return rc;
}

int snpgpu_gnrGRM_avg_val(double *avg_val)
{
    if (avg_val) *avg_val = g_grm_avg_value;
    return 0;
}

// gnrIBD_PLINK, src/genIBS.cpp:558-639 with Init_EPrIBD_IBS, src/genIBD.cpp:253-338
int snpgpu_gnrIBD_PLINK(int, const double *allele_freq, int kinship_constraint, int use_matrix, int, double *k0,
                        double *k1, double *afreq_out)
{
    if (need_ws("snpgpu_gnrIBD_PLINK")) return 1;
    std::vector<int32_t> sum, num, het;
    if (ws_stats(sum, num, &het)) return 1;
    // E[IBS state | IBD state] per SNP as polynomials in the allele frequencies p, q = 1 - p (PLINK's method of moments;
    // the reference's Init_EPrIBD_IBS, src/genIBD.cpp:253-338).  Each monomial c p^a q^b stands for drawing a copies of
    // allele A and b of allele B; with frequencies estimated from the same sample the draws are without replacement, and
    // the unbiased estimate multiplies the monomial by   [x]_a / x^a * [y]_b / y^b * T^(a+b) / [T]_(a+b)
    // ([v]_k = v (v-1) ... (v-k+1); x, y = allele counts, T = x + y).  Caller-supplied frequencies take the plain monomials.
    struct Mono { double c; int a, b; };
    static const Mono E00[] = {{2, 2, 2}};
    static const Mono E01[] = {{4, 3, 1}, {4, 1, 3}};
    static const Mono E02[] = {{1, 0, 4}, {1, 4, 0}, {4, 2, 2}};
    static const Mono E11[] = {{2, 2, 1}, {2, 1, 2}};
    static const Mono E12[] = {{1, 3, 0}, {1, 0, 3}, {1, 2, 1}, {1, 1, 2}};
    static const struct { const Mono *m; int n; } POLY[5] = {{E00, 1}, {E01, 2}, {E02, 3}, {E11, 2}, {E12, 4}};
    auto falling = [](double v, int k) { double r = 1; for (int m = 1; m < k; m++) r *= (v - m) / v; return r; };   // [v]_k / v^k
    const double nan = std::numeric_limits<double>::quiet_NaN();
    double tot[5] = {0, 0, 0, 0, 0};
    long nValid = 0;
    for (size_t l = 0; l < sum.size(); l++) {
        const long nAB = het[l], nAA = (sum[l] - het[l]) / 2, nBB = num[l] - nAA - nAB;
        const double x = allele_freq ? 0.0 : 2.0 * nAA + nAB, y = allele_freq ? 0.0 : 2.0 * nBB + nAB, T = x + y;
        double p = allele_freq ? allele_freq[l] : (T > 0 ? x / T : nan);
        if (allele_freq && std::isfinite(p) && (p < 0 || p > 1)) p = nan;
        if (afreq_out) afreq_out[l] = p;
        const double q = 1 - p;
        double val[5];
        bool finite = true;
        for (int k = 0; k < 5; k++) {
            double v = 0;
            for (int t = 0; t < POLY[k].n; t++) {
                const Mono &m = POLY[k].m[t];
                double term = m.c;
                for (int r = 0; r < m.a; r++) term *= p;
                for (int r = 0; r < m.b; r++) term *= q;
                if (!allele_freq) term *= falling(x, m.a) * falling(y, m.b) / falling(T, m.a + m.b);
                v += term;
            }
            val[k] = v;
            finite = finite && std::isfinite(v);
        }
        if (finite) { for (int k = 0; k < 5; k++) tot[k] += val[k]; nValid++; }      // SNPs with a non-finite term are skipped
    }
    const double s00 = tot[0], s01 = tot[1], s02 = tot[2], s11 = tot[3], s12 = tot[4];
    const double e[5] = {s00 / nValid, s01 / nValid, s02 / nValid, s11 / nValid, s12 / nValid};
    CtxGuard g;
    if (run_stream(SNPGPU_IBS, 0, &g.c)) return 1;
    return snpgpu_ibd_mom(g.c, e, kinship_constraint, k0, k1, use_matrix ? 1 : 0, SNPGPU_HOST);
}

int snpgpu_gnrIBD_Beta(int inbreeding, int, int use_matrix, int, double *out, double *avg_val)
{
    if (need_ws("snpgpu_gnrIBD_Beta")) return 1;
    CtxGuard g;
    if (run_stream(SNPGPU_INDIV_BETA, 0, &g.c)) return 1;
    double avg = 0;
    if (snpgpu_indiv_beta(g.c, inbreeding ? 1 : 0, out, &avg, use_matrix ? 1 : 0, SNPGPU_HOST)) return 1;
    g_grm_avg_value = avg;
    if (avg_val) *avg_val = avg;
    return 0;
}

// gnrEigMix, src/genEIGMIX.cpp:656-740
int snpgpu_gnrEigMix(int eigen_cnt, int, int diagadj, int, double *ibd, double *eigval, double *eigvec, double *afreq)
{
    if (need_ws("snpgpu_gnrEigMix")) return 1;
    const int64_t n = g_ws.n_samp;
    if (afreq) {
        std::vector<int32_t> sum, num;
        if (ws_stats(sum, num)) return 1;
        for (size_t l = 0; l < sum.size(); l++) afreq[l] = (num[l] > 0) ? (0.5 * sum[l] / num[l]) : 0.0;   // 0.5*avg_geno, :116
    }
    CtxGuard g;
    if (run_stream(SNPGPU_EIGMIX, 0, &g.c)) return 1;
    if (ibd && snpgpu_eigmix(g.c, diagadj, 1.0, ibd, 0, SNPGPU_HOST)) return 1;
    int k = eigen_cnt;
    if (k < 0 || k > n) k = (int)n;          // :676
    if ((eigval || eigvec) && k > 0 && !dense_route(n, k)) {
        // beyond the dense solver: the coancestry matrix replaces the sums in place, block Krylov on the panel
        std::vector<double> w((size_t)k);
        snpgpu_ctx *panels[1] = {g.c};
        if (snpgpu_finalize_inplace(g.c, diagadj, 1.0) ||
            snpgpu_panels_topk_eigen(panels, 1, 1.0, k, nullptr, w.data(), eigvec, SNPGPU_HOST, nullptr))
            return 1;
        if (eigval) {
            for (int i = 0; i < k; i++) eigval[i] = w[(size_t)i];
            for (int64_t i = k; i < n; i++) eigval[i] = std::numeric_limits<double>::quiet_NaN();
        }
    } else if ((eigval || eigvec) && k > 0) {
        SNPGPU_HIP_CHECK(hipSetDevice(g.c->device));
        DevBuf A;
        if (A.alloc(sizeof(double) * (size_t)n * (size_t)n)) return 1;
        int rc = snpgpu_eigmix(g.c, diagadj, 1.0, (double *)A.p, 0, SNPGPU_DEVICE);
        std::vector<double> w((size_t)k);
        if (!rc) rc = dense_topk(g.c->device, g.c->stream, (double *)A.p, n, k, w.data(), eigvec, SNPGPU_HOST);
        A.release();
        if (rc == 2) {          // the dense solver declined this size
            snpgpu_ctx *panels[1] = {g.c};
            rc = snpgpu_finalize_inplace(g.c, diagadj, 1.0) ||
                 snpgpu_panels_topk_eigen(panels, 1, 1.0, k, nullptr, w.data(), eigvec, SNPGPU_HOST, nullptr);
        }
        if (rc) return 1;
        if (eigval) {
            for (int i = 0; i < k; i++) eigval[i] = w[(size_t)i];
            for (int64_t i = k; i < n; i++) eigval[i] = std::numeric_limits<double>::quiet_NaN();
        }
    }
    return 0;
}

// gnrPCA "exact", src/genPCA.cpp:1355-1452
int snpgpu_gnrPCA(int eigen_cnt, int, int bayesian, int, double *trace_xtx, double *genmat, double *eigval,
                  double *eigvec, double *trace_val)
{
    if (need_ws("snpgpu_gnrPCA")) return 1;
    const int64_t n = g_ws.n_samp;
    CtxGuard g;
    if (run_stream(SNPGPU_PCA_COV, bayesian, &g.c)) return 1;
    double tr = 0;
    if (snpgpu_pca_cov(g.c, genmat, 0, 1, 0.0, &tr, SNPGPU_HOST)) return 1;
    if (trace_xtx) *trace_xtx = tr;
    if (trace_val) *trace_val = (double)(n - 1);  // trace after the (n-1)/trace scaling, :1390
    if (eigval || eigvec) {
        if (eigen_cnt < 0) { set_error("Invalid 'eigen.cnt'."); return 1; }
        int k = eigen_cnt > n ? (int)n : eigen_cnt;
        if (k > 0) {
            std::vector<double> w((size_t)k);
            if (snpgpu_pca_eigen(g.c, k, w.data(), eigvec, SNPGPU_HOST)) return 1;
            if (eigval) {
                for (int i = 0; i < k; i++) eigval[i] = w[(size_t)i];
                for (int64_t i = k; i < n; i++) eigval[i] = std::numeric_limits<double>::quiet_NaN();  // :1343-1345
            }
        }
    }
    return 0;
}

struct ProjGuard {
    snpgpu_proj *p = nullptr;
    ~ProjGuard() { if (p) snpgpu_proj_destroy(p); }
};

static int proj_open(int n_eig, ProjGuard &g)
{
    snpgpu_opts o{};
    o.device = g_ws.device;
    o.max_block_snps = WS_BLOCK;
    return snpgpu_proj_create(g_ws.n_samp, n_eig, &o, &g.p);
}

// gnrPCA "randomized": CRandomPCA::Run, src/genPCA.cpp:672-792 (Galinsky's fast PCA), on the projection kernels.
//   Y = normalised genotypes (n_snp x n_samp), y = (g - avg) / sqrt(2 p (1 - p)), missing -> 0   (:503-519)
//   H_i = Y G_i (:528-579), G_{i+1} = Y^T H_i / n_snp (:581-613, 728-750): here both in ONE pass per iteration
//   (G_{i+1} = sum over blocks of Y_b^T (Y_b G_i)), an orthonormal basis Q of span(H_0 .. H_iter) (the
//   reference takes the right singular vectors of MatH, :757; any orthonormal basis gives the same T up to a
//   rotation: Householder QR), T = Q^T Y (:615-640, 763-781) and the SVD of T (:783-784).
int snpgpu_gnrPCA_randomized(int eigen_cnt, int aux_dim, int iter_num, const double *aux_mat, int, int, double *sigma,
                             double *eigvec, double *trace2)
{
    if (need_ws("snpgpu_gnrPCA")) return 1;
    const int64_t N = g_ws.n_samp, L = (int64_t)g_ws.sel.size();
    if (!aux_mat || aux_dim <= 0 || iter_num < 0) { set_error("snpgpu_gnrPCA: invalid 'aux.dim' / 'iter.num' / 'aux.mat'"); return 1; }
    const int A = aux_dim;
    const int64_t hs64 = (int64_t)A * (iter_num + 1);
    if (hs64 > 4096) { set_error("snpgpu_gnrPCA: aux.dim * (iter.num + 1) is limited to 4096"); return 1; }
    const int hs = (int)hs64;
    if (L < hs) { set_error("snpgpu_gnrPCA: the randomized algorithm needs at least aux.dim * (iter.num + 1) SNPs"); return 1; }
    if (eigen_cnt <= 0 || eigen_cnt > hs || eigen_cnt > N) { set_error("Invalid 'eigen.cnt'."); return 1; }

    // TraceXTX from the per-SNP genotype counts (:503-519)
    std::vector<int32_t> sum, num, het;
    if (ws_stats(sum, num, &het)) return 1;
    double trace = 0;
    for (int64_t l = 0; l < L; l++) {
        const double m = num[l], n1 = het[l], n2 = (sum[l] - het[l]) / 2, n0 = m - n1 - n2;
        const double avg = (m > 0) ? sum[l] / m : 0.0, p = avg * 0.5;
        const double s2 = (0 < p && p < 1) ? 1.0 / (2 * p * (1 - p)) : 0.0;
        trace += s2 * (n0 * avg * avg + n1 * (1 - avg) * (1 - avg) + n2 * (2 - avg) * (2 - avg));
    }
    if (trace2) *trace2 = 2 * trace;

    SNPGPU_HIP_CHECK(hipSetDevice(g_ws.device));
    ProjGuard g1, g2;
    if (proj_open(A, g1)) return 1;
    snpgpu_proj *P = g1.p;
    hipStream_t st = P->stream;
    const double rs2 = 1.0 / std::sqrt(2.0);      // loadings scale with 1/sqrt(p(1-p)), Y with 1/sqrt(2p(1-p))
    DevBuf Ht, blk, avgAll, scAll, tau, work, info, S, U, sl2;
    hipsolverHandle_t hh = nullptr;
    struct Cleanup {      // also on the early returns of SNPGPU_HIP_CHECK
        std::vector<DevBuf *> bufs;
        hipsolverHandle_t *h;
        ~Cleanup()
        {
            if (*h) hipsolverDestroy(*h);
            for (DevBuf *b : bufs) b->release();
        }
    } cleanup{{&Ht, &blk, &avgAll, &scAll, &tau, &work, &info, &S, &U, &sl2}, &hh};
    int rc = Ht.alloc(8 * (size_t)hs * (size_t)L) | blk.alloc(8 * (size_t)WS_BLOCK * (size_t)A) |
             avgAll.alloc(8 * (size_t)L) | scAll.alloc(8 * (size_t)L) | tau.alloc(8 * (size_t)hs) | info.alloc(sizeof(int));
    std::vector<uint8_t> buf;
    auto fail = [&](const char *m) { set_error(std::string("snpgpu_gnrPCA (randomized): ") + m); rc = 1; };
    do {
        if (rc) break;
        if (snpgpu_proj_set_eigvec(P, aux_mat, SNPGPU_HOST)) { rc = 1; break; }
        for (int it = 0; it <= iter_num && !rc; it++) {
            for (int64_t i0 = 0; i0 < L && !rc; i0 += WS_BLOCK) {
                const int64_t i1 = std::min(L, i0 + WS_BLOCK), nb = i1 - i0;
                gather_block(i0, i1, buf);
                double *af = (double *)avgAll.p + i0, *sc = (double *)scAll.p + i0;
                // H_i block = Y_b G_i
                if (snpgpu_proj_snp_loading(P, buf.data(), nb, SNPGPU_GENO_PACKED2, SNPGPU_HOST, 0, (double *)blk.p, af, sc,
                                            SNPGPU_DEVICE)) { rc = 1; break; }
                hipLaunchKernelGGL(rp_scatter_kernel, dim3((unsigned)((nb * A + 255) / 256)), dim3(256), 0, st,
                                   (double *)blk.p, nb, A, rs2, (double *)Ht.p, L, i0, A * it);
                hipLaunchKernelGGL(rp_scale_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, sc, (size_t)nb, rs2);
                // G_{i+1} += Y_b^T H_i block (same staged block)
                if (it < iter_num &&
                    snpgpu_proj_samp_loading_feed(P, nullptr, nb, SNPGPU_GENO_PACKED2, SNPGPU_DEVICE, (const double *)blk.p,
                                                  af, sc, SNPGPU_DEVICE)) { rc = 1; break; }
            }
            if (rc || it == iter_num) break;
            hipLaunchKernelGGL(rp_scale_kernel, dim3((unsigned)(((size_t)N * A + 255) / 256)), dim3(256), 0, st,
                               (double *)P->acc.p, (size_t)N * (size_t)A, 1.0 / (double)L);
            if (snpgpu_proj_set_eigvec(P, (const double *)P->acc.p, SNPGPU_DEVICE)) { rc = 1; break; }
            if (snpgpu_proj_samp_loading_reset(P)) { rc = 1; break; }
        }
        if (rc) break;
        // orthonormal basis of span(H): Householder QR of the L x hs matrix held in Ht (column-major, lda = L)
        if (hipsolverCreate(&hh) != HIPSOLVER_STATUS_SUCCESS) { fail("hipsolverCreate failed"); break; }
        hipsolverSetStream(hh, st);
        int lw1 = 0, lw2 = 0;
        if (hipsolverDnDgeqrf_bufferSize(hh, (int)L, hs, (double *)Ht.p, (int)L, &lw1) != HIPSOLVER_STATUS_SUCCESS ||
            hipsolverDnDorgqr_bufferSize(hh, (int)L, hs, hs, (double *)Ht.p, (int)L, (double *)tau.p, &lw2) != HIPSOLVER_STATUS_SUCCESS) {
            fail("QR workspace query failed"); break;
        }
        if (work.alloc(8 * (size_t)std::max(std::max(lw1, lw2), 1))) { rc = 1; break; }
        int hinfo = 0;
        if (hipsolverDnDgeqrf(hh, (int)L, hs, (double *)Ht.p, (int)L, (double *)tau.p, (double *)work.p, lw1, (int *)info.p) != HIPSOLVER_STATUS_SUCCESS ||
            hipsolverDnDorgqr(hh, (int)L, hs, hs, (double *)Ht.p, (int)L, (double *)tau.p, (double *)work.p, lw2, (int *)info.p) != HIPSOLVER_STATUS_SUCCESS) {
            fail("QR failed"); break;
        }
        // T^T = Y^T Q, accumulated as sample loadings with the Q rows of each block
        if (proj_open(hs, g2)) { rc = 1; break; }
        snpgpu_proj *P2 = g2.p;
        if (sl2.alloc(8 * (size_t)WS_BLOCK * (size_t)hs)) { rc = 1; break; }
        SNPGPU_HIP_CHECK(hipStreamSynchronize(st));
        for (int64_t i0 = 0; i0 < L && !rc; i0 += WS_BLOCK) {
            const int64_t i1 = std::min(L, i0 + WS_BLOCK), nb = i1 - i0;
            gather_block(i0, i1, buf);
            hipLaunchKernelGGL(rp_gather_kernel, dim3((unsigned)((nb * hs + 255) / 256)), dim3(256), 0, P2->stream,
                               (const double *)Ht.p, L, i0, nb, hs, (double *)sl2.p);
            if (snpgpu_proj_samp_loading_feed(P2, buf.data(), nb, SNPGPU_GENO_PACKED2, SNPGPU_HOST, (const double *)sl2.p,
                                              (const double *)avgAll.p + i0, (const double *)scAll.p + i0, SNPGPU_DEVICE))
                rc = 1;
        }
        if (rc) break;
        // SVD of T (hs x N): P2->acc holds T^T column-major (N x hs)
        hipsolverSetStream(hh, P2->stream);
        const int64_t mn = std::min<int64_t>(hs, N);
        if (S.alloc(8 * (size_t)mn)) { rc = 1; break; }
        int lw = 0;
        std::vector<double> ev((size_t)N * (size_t)eigen_cnt);
        if (N >= hs) {            // T^T = U' S V'^T (tall): the sample eigenvectors are the columns of U'
            if (U.alloc(8 * (size_t)N * (size_t)hs)) { rc = 1; break; }
            if (hipsolverDnDgesvd_bufferSize(hh, (int)N, hs, &lw) != HIPSOLVER_STATUS_SUCCESS) { fail("SVD workspace query failed"); break; }
            work.release();
            if (work.alloc(8 * (size_t)std::max(lw, 1))) { rc = 1; break; }
            if (hipsolverDnDgesvd(hh, 'S', 'N', (int)N, hs, (double *)P2->acc.p, (int)N, (double *)S.p, (double *)U.p, (int)N,
                                  nullptr, hs, (double *)work.p, lw, nullptr, (int *)info.p) != HIPSOLVER_STATUS_SUCCESS) {
                fail("LAPACK::DGESVD error"); break;
            }
            SNPGPU_HIP_CHECK(hipMemcpyAsync(ev.data(), U.p, 8 * (size_t)N * (size_t)eigen_cnt, hipMemcpyDeviceToHost, P2->stream));
        } else {                  // T (hs x N, tall): the sample eigenvectors are the rows of V^T (N x N)
            DevBuf Tt, VT;
            if (Tt.alloc(8 * (size_t)hs * (size_t)N) | VT.alloc(8 * (size_t)N * (size_t)N)) { rc = 1; break; }
            hipLaunchKernelGGL(rp_transpose_kernel, dim3((unsigned)(((size_t)N * hs + 255) / 256)), dim3(256), 0, P2->stream,
                               (const double *)P2->acc.p, N, (int64_t)hs, (double *)Tt.p);
            if (hipsolverDnDgesvd_bufferSize(hh, hs, (int)N, &lw) != HIPSOLVER_STATUS_SUCCESS) { fail("SVD workspace query failed"); Tt.release(); VT.release(); break; }
            work.release();
            if (work.alloc(8 * (size_t)std::max(lw, 1))) { rc = 1; Tt.release(); VT.release(); break; }
            hipsolverStatus_t ss = hipsolverDnDgesvd(hh, 'N', 'S', hs, (int)N, (double *)Tt.p, hs, (double *)S.p, nullptr, hs,
                                                     (double *)VT.p, (int)N, (double *)work.p, lw, nullptr, (int *)info.p);
            std::vector<double> vt((size_t)N * (size_t)N);
            hipError_t e = hipMemcpyAsync(vt.data(), VT.p, 8 * vt.size(), hipMemcpyDeviceToHost, P2->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(P2->stream);
            Tt.release(); VT.release();
            if (ss != HIPSOLVER_STATUS_SUCCESS || e != hipSuccess) { fail("LAPACK::DGESVD error"); break; }
            for (int r = 0; r < eigen_cnt; r++)
                for (int64_t i = 0; i < N; i++) ev[(size_t)r * N + i] = vt[(size_t)i * N + r];   // V^T(r, i), column-major N x N
        }
        std::vector<double> sg((size_t)mn);
        SNPGPU_HIP_CHECK(hipMemcpyAsync(sg.data(), S.p, 8 * (size_t)mn, hipMemcpyDeviceToHost, P2->stream));
        SNPGPU_HIP_CHECK(hipMemcpyAsync(&hinfo, info.p, sizeof(int), hipMemcpyDeviceToHost, P2->stream));
        SNPGPU_HIP_CHECK(hipStreamSynchronize(P2->stream));
        if (hinfo != 0) { set_error("LAPACK::DGESVD error (" + std::to_string(hinfo) + ")."); rc = 1; break; }
        if (sigma) {
            for (int64_t i = 0; i < N; i++) sigma[i] = (i < mn) ? sg[(size_t)i] : 0.0;     // vector<double> sigma(nSamp), :783
        }
        if (eigvec) memcpy(eigvec, ev.data(), 8 * ev.size());
    } while (0);
    return rc;
}

// gnrPCACorr, src/genPCA.cpp:1455-1484 (matrix result; the GDS-node variant Run2 appends the same blocks)
int snpgpu_gnrPCACorr(int len_eig, const double *eigvec, int, int, double *out)
{
    if (need_ws("snpgpu_gnrPCACorr")) return 1;
    if (!eigvec || !out || len_eig <= 0) { set_error("snpgpu_gnrPCACorr: invalid argument"); return 1; }
    ProjGuard g;
    if (proj_open(len_eig, g)) return 1;
    if (snpgpu_proj_set_eigvec(g.p, eigvec, SNPGPU_HOST)) return 1;
    std::vector<uint8_t> buf;
    const int64_t L = (int64_t)g_ws.sel.size();
    for (int64_t i0 = 0; i0 < L; i0 += WS_BLOCK) {
        const int64_t i1 = std::min(L, i0 + WS_BLOCK);
        gather_block(i0, i1, buf);
        if (snpgpu_proj_snp_corr(g.p, buf.data(), i1 - i0, SNPGPU_GENO_PACKED2, SNPGPU_HOST,
                                 out + (size_t)i0 * (size_t)len_eig, SNPGPU_HOST))
            return 1;
    }
    return 0;
}

// gnrPCASNPLoading, src/genPCA.cpp:1488-1531
int snpgpu_gnrPCASNPLoading(const double *eigval, const double *eigvec, int len_eig, double trace_xtx, int, int bayesian,
                            int, double *loading, double *afreq, double *scale)
{
    if (need_ws("snpgpu_gnrPCASNPLoading")) return 1;
    if (!eigval || !eigvec || !loading || !afreq || !scale || len_eig <= 0) {
        set_error("snpgpu_gnrPCASNPLoading: invalid argument");
        return 1;
    }
    const int64_t n = g_ws.n_samp;
    // scale eigenvectors with eigenvalues, :1499-1507
    std::vector<double> ev((size_t)n * (size_t)len_eig);
    const double sc = (double)(n - 1) / trace_xtx;
    for (int i = 0; i < len_eig; i++) {
        const double f = std::sqrt(sc / eigval[i]);
        for (int64_t j = 0; j < n; j++) ev[(size_t)i * n + j] = eigvec[(size_t)i * n + j] * f;
    }
    ProjGuard g;
    if (proj_open(len_eig, g)) return 1;
    if (snpgpu_proj_set_eigvec(g.p, ev.data(), SNPGPU_HOST)) return 1;
    std::vector<uint8_t> buf;
    const int64_t L = (int64_t)g_ws.sel.size();
    for (int64_t i0 = 0; i0 < L; i0 += WS_BLOCK) {
        const int64_t i1 = std::min(L, i0 + WS_BLOCK);
        gather_block(i0, i1, buf);
        if (snpgpu_proj_snp_loading(g.p, buf.data(), i1 - i0, SNPGPU_GENO_PACKED2, SNPGPU_HOST, bayesian,
                                    loading + (size_t)i0 * (size_t)len_eig, afreq + i0, scale + i0, SNPGPU_HOST))
            return 1;
    }
    return 0;
}

// gnrPCASampLoading, src/genPCA.cpp:1535-1562
int snpgpu_gnrPCASampLoading(int eigen_cnt, const double *snp_loadings, const double *avg_freq, const double *scale, int,
                             int, double *out)
{
    if (need_ws("snpgpu_gnrPCASampLoading")) return 1;
    if (!snp_loadings || !avg_freq || !scale || !out || eigen_cnt <= 0) {
        set_error("snpgpu_gnrPCASampLoading: invalid argument");
        return 1;
    }
    ProjGuard g;
    if (proj_open(eigen_cnt, g)) return 1;
    std::vector<uint8_t> buf;
    const int64_t L = (int64_t)g_ws.sel.size();
    for (int64_t i0 = 0; i0 < L; i0 += WS_BLOCK) {
        const int64_t i1 = std::min(L, i0 + WS_BLOCK);
        gather_block(i0, i1, buf);
        if (snpgpu_proj_samp_loading_feed(g.p, buf.data(), i1 - i0, SNPGPU_GENO_PACKED2, SNPGPU_HOST,
                                          snp_loadings + (size_t)i0 * (size_t)eigen_cnt, avg_freq + i0, scale + i0,
                                          SNPGPU_HOST))
            return 1;
    }
    return snpgpu_proj_samp_loading(g.p, out, SNPGPU_HOST);
}

// EIGMIX: y = (g - 2 p) / sqrt(sum_snp 4 p (1 - p)) over the called genotypes (CEigMix_SNPLoad / CEigMix_SampleLoad,
// src/genEIGMIX.cpp:440-512, 516-620)
static void eigmix_norm(const double *afreq, int64_t L, std::vector<double> &avg, std::vector<double> &scale)
{
    double sum = 0;
    for (int64_t i = 0; i < L; i++) sum += 4 * afreq[i] * (1 - afreq[i]);
    const double sc = 1 / std::sqrt(sum);
    avg.resize((size_t)L); scale.assign((size_t)L, sc);
    for (int64_t i = 0; i < L; i++) avg[(size_t)i] = 2 * afreq[i];
}

// gnrEigMixSNPLoading, src/genEIGMIX.cpp:739-775
int snpgpu_gnrEigMixSNPLoading(const double *eigval, const double *eigvec, int len_eig, const double *afreq, int, int,
                               double *loading)
{
    if (need_ws("snpgpu_gnrEigMixSNPLoading")) return 1;
    if (!eigval || !eigvec || !afreq || !loading || len_eig <= 0) { set_error("snpgpu_gnrEigMixSNPLoading: invalid argument"); return 1; }
    const int64_t n = g_ws.n_samp, L = (int64_t)g_ws.sel.size();
    std::vector<double> ev((size_t)n * (size_t)len_eig), avg, scale;
    for (int i = 0; i < len_eig; i++) {          // scale eigenvectors with eigenvalues, :750-758
        const double f = std::sqrt(1 / eigval[i]);
        for (int64_t j = 0; j < n; j++) ev[(size_t)i * n + j] = eigvec[(size_t)i * n + j] * f;
    }
    eigmix_norm(afreq, L, avg, scale);
    ProjGuard g;
    if (proj_open(len_eig, g)) return 1;
    if (snpgpu_proj_set_eigvec(g.p, ev.data(), SNPGPU_HOST)) return 1;
    std::vector<uint8_t> buf;
    for (int64_t i0 = 0; i0 < L; i0 += WS_BLOCK) {
        const int64_t i1 = std::min(L, i0 + WS_BLOCK);
        gather_block(i0, i1, buf);
        if (snpgpu_proj_snp_loading_ext(g.p, buf.data(), i1 - i0, SNPGPU_GENO_PACKED2, SNPGPU_HOST, avg.data() + i0,
                                        scale.data() + i0, SNPGPU_HOST, loading + (size_t)i0 * (size_t)len_eig, SNPGPU_HOST))
            return 1;
    }
    return 0;
}

// gnrEigMixSampLoading, src/genEIGMIX.cpp:777-803
int snpgpu_gnrEigMixSampLoading(int eigen_cnt, const double *snp_loadings, const double *afreq, int, int, double *out)
{
    if (need_ws("snpgpu_gnrEigMixSampLoading")) return 1;
    if (!snp_loadings || !afreq || !out || eigen_cnt <= 0) { set_error("snpgpu_gnrEigMixSampLoading: invalid argument"); return 1; }
    const int64_t L = (int64_t)g_ws.sel.size();
    std::vector<double> avg, scale;
    eigmix_norm(afreq, L, avg, scale);
    ProjGuard g;
    if (proj_open(eigen_cnt, g)) return 1;
    std::vector<uint8_t> buf;
    for (int64_t i0 = 0; i0 < L; i0 += WS_BLOCK) {
        const int64_t i1 = std::min(L, i0 + WS_BLOCK);
        gather_block(i0, i1, buf);
        if (snpgpu_proj_samp_loading_feed(g.p, buf.data(), i1 - i0, SNPGPU_GENO_PACKED2, SNPGPU_HOST,
                                          snp_loadings + (size_t)i0 * (size_t)eigen_cnt, avg.data() + i0, scale.data() + i0,
                                          SNPGPU_HOST))
            return 1;
    }
    return snpgpu_proj_samp_loading(g.p, out, SNPGPU_HOST);
}

}  // extern "C"
