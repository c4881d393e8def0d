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
int ws_stats(std::vector<int32_t> &sum, std::vector<int32_t> &num)
{
    const int64_t L = (int64_t)g_ws.sel.size();
    sum.assign((size_t)L, 0);
    num.assign((size_t)L, 0);
    if (L == 0) return 0;
    SNPGPU_HIP_CHECK(hipSetDevice(g_ws.device));
    const int64_t N = g_ws.n_samp, RB = (N + 255) / 256 * 64;
    DevBuf raw, packed, dsum, dnum, dmiss;
    int rc = raw.alloc((size_t)WS_BLOCK * g_ws.rb) | packed.alloc((size_t)WS_BLOCK * RB) |
             dsum.alloc(sizeof(int32_t) * WS_BLOCK) | dnum.alloc(sizeof(int32_t) * WS_BLOCK) | dmiss.alloc(8);
    std::vector<uint8_t> buf;
    for (int64_t i0 = 0; i0 < L && !rc; i0 += WS_BLOCK) {
        const int64_t i1 = std::min(L, i0 + WS_BLOCK), nb = i1 - i0;
        gather_block(i0, i1, buf);
        hipError_t e = hipMemcpy(raw.p, buf.data(), buf.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(dmiss.p, 0, 8);
        if (e != hipSuccess) { set_error(std::string("ws_stats: ") + hipGetErrorString(e)); rc = 1; break; }
        rc |= launch_repack(nullptr, raw.p, SNPGPU_GENO_PACKED2, nb, N, (uint8_t *)packed.p, RB);
        rc |= launch_snp_stats(nullptr, (const uint8_t *)packed.p, RB, nb, N, (int32_t *)dsum.p, (int32_t *)dnum.p,
                               (unsigned long long *)dmiss.p);
        if (rc) break;
        e = hipMemcpy(&sum[i0], dsum.p, sizeof(int32_t) * nb, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(&num[i0], dnum.p, sizeof(int32_t) * nb, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_error(std::string("ws_stats: ") + hipGetErrorString(e)); rc = 1; }
    }
    raw.release(); packed.release(); dsum.release(); dnum.release(); dmiss.release();
    return rc;
}

__global__ void negate_kernel(double *a, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = -a[i];
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
    set_error("Invalid 'method'!");  // src/genPCA.cpp:1710
    return 1;
}

// top-k eigenpairs of the normalised covariance: the reference negates the packed matrix and asks
// LAPACK dspevx for eigenvalues IL=1..IU=k (src/genPCA.cpp:1308-1341, :1419); here the full matrix is
// negated on the device and hipSOLVER's syevdx is asked for the same index range.
int snpgpu_pca_eigen(snpgpu_ctx *c, int k, double *eigval, double *eigvec, int mem)
{
    if (!c || c->kind != SNPGPU_PCA_COV || !c->full) { set_error("snpgpu_pca_eigen: needs a full PCA_COV context"); return 1; }
    const int64_t n = c->N;
    if (k <= 0 || k > n) { set_error("Invalid 'eigen.cnt'."); return 1; }
    if (n > 46340) { set_error("snpgpu_pca_eigen: dense eigen solver limited to n <= 46340"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(c->device));
    DevBuf A, W, work, info;
    int rc = A.alloc(sizeof(double) * (size_t)n * (size_t)n) | W.alloc(sizeof(double) * (size_t)n) | info.alloc(sizeof(int));
    hipsolverHandle_t h = nullptr;
    do {
        if (rc) break;
        rc = snpgpu_pca_cov(c, (double *)A.p, 0, 1, 0.0, nullptr, SNPGPU_DEVICE);
        if (rc) break;
        const size_t nn = (size_t)n * (size_t)n;
        hipLaunchKernelGGL(negate_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, c->stream, (double *)A.p, nn);
        if (hipStreamSynchronize(c->stream) != hipSuccess) { set_error("snpgpu_pca_eigen: negate failed"); rc = 1; break; }
        if (hipsolverCreate(&h) != HIPSOLVER_STATUS_SUCCESS) { set_error("hipsolverCreate failed"); rc = 1; break; }
        hipsolverSetStream(h, c->stream);
        int lwork = 0, nev = 0;
        if (hipsolverDnDsyevdx_bufferSize(h, HIPSOLVER_EIG_MODE_VECTOR, HIPSOLVER_EIG_RANGE_I, HIPBLAS_FILL_MODE_LOWER,
                                          (int)n, (double *)A.p, (int)n, 0.0, 0.0, 1, k, &nev, (double *)W.p,
                                          &lwork) != HIPSOLVER_STATUS_SUCCESS) {
            set_error("hipsolverDnDsyevdx_bufferSize failed"); rc = 1; break;
        }
        if (work.alloc(sizeof(double) * (size_t)std::max(lwork, 1))) { rc = 1; break; }
        hipsolverStatus_t s = hipsolverDnDsyevdx(h, HIPSOLVER_EIG_MODE_VECTOR, HIPSOLVER_EIG_RANGE_I,
                                                 HIPBLAS_FILL_MODE_LOWER, (int)n, (double *)A.p, (int)n, 0.0, 0.0, 1, k,
                                                 &nev, (double *)W.p, (double *)work.p, lwork, (int *)info.p);
        int hinfo = 0;
        hipError_t e = hipMemcpyAsync(&hinfo, info.p, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
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
        if (eigvec && hipMemcpy(eigvec, A.p, sizeof(double) * (size_t)n * (size_t)k, kv) != hipSuccess) { set_error("eigen copy failed"); rc = 1; break; }
    } while (0);
    if (h) hipsolverDestroy(h);
    A.release(); W.release(); work.release(); info.release();
    return rc;
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

}  // extern "C"
