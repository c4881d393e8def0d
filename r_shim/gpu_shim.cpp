// gpu_shim.cpp -- the `.Call` routines of SNPRelate's pairwise hot path bound to libsnpgpu (include/snpgpu.h).
//
// Added to the package's src/ directory next to the kept sources.  R headers and gdsfmt are not in this
// repository's build image, so this file is NOT compiled or run here; every libsnpgpu call below is exercised in
// exactly this order and with exactly these buffer shapes by tests/test_gpu_shim_order.py through ctypes.
//
// What stays (reference v1.46.0): R/*.R unchanged (same snpgds* signatures, same `.Call(gnrXxx, ...)` sites:
// R/IBS.R:36,68, R/IBD.R:383,399,594, R/PCA.R:70), `.InitFile2` and the working space (gnrSetGenoSpace /
// gnrSelSNP_Base, src/SNPRelate.cpp:76-210), the GDS block reader CGenoReadBySNP (src/dGenGWAS.cpp:1218-1397), the
// progress bar / interrupt polling, the `out.gds` row appends (GDS_Array_AppendData) and LAPACK-free results.
// What goes: the bodies of the seven routines below -- the CIBSCount / CKINGRobust / CKINGHomo / CGCTA_AlgArith /
// CExactPCA / CEigMix / CIndivBeta `Run` calls with their thread pools and SIMD loops -- and CalcEigen's dspevx call.
//
// Registration (src/SNPRelate.cpp:1154-1205): the seven entries of callMethods[] point at the functions below,
// names and arity unchanged -- see r_shim/registration.inc.  R_useDynamicSymbols(FALSE) stays.
//
//   routine replaced                                            reference body
//   gpu_gnrIBSNum(NumThread, Verbose)                           src/genIBS.cpp:500-550
//   gpu_gnrIBSAve(NumThread, useMatrix, Verbose)                src/genIBS.cpp:441-497
//   gpu_gnrIBD_KING_Robust(FamilyID, NumThread, useMatrix, V.)  src/genKING.cpp:576-679
//   gpu_gnrIBD_KING_Homo(NumThread, useMatrix, Verbose)         src/genKING.cpp:493-570
//   gpu_gnrGRM(NumThread, Method, GDS, useMatrix, Verbose)      src/genPCA.cpp:1614-1717 (+ grm_output :1586-1602,
//                                                                grm_save_to_gds :1571-1584)
//   gpu_gnrGRM_avg_val()                                        src/genPCA.cpp:1605-1611
//   gpu_gnrPCA(EigenCnt, Algorithm, NumThread, ParamList, V.)   src/genPCA.cpp:1355-1452 (+ CalcEigen :1262-1346)
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dGenGWAS.h"   // kept: MCWorkingGeno, CdBaseWorkSpace, CGenoReadBySNP, CProgress, CachingSNPData, SEXP_Verbose
#include <snpgpu.h>

using namespace GWAS;

namespace {

[[noreturn]] void gpu_fail() { throw ErrCoreArray("%s", snpgpu_last_error()); }

// additive knobs, never a signature change: options(snpgpu.device=) / options(snpgpu.block.snps=), else the environment
int opt_int(const char *r_option, const char *env, int dflt)
{
    SEXP v = Rf_GetOption1(Rf_install(r_option));
    if (v != R_NilValue && Rf_length(v) == 1) {
        const int x = Rf_asInteger(v);
        if (x != NA_INTEGER) return x;
    }
    const char *e = getenv(env);
    return (e && *e) ? atoi(e) : dflt;
}

// One accumulator context; released in the destructor, i.e. before gdsfmt's catch block turns the C++ exception
// into Rf_error (no longjmp ever crosses live device state).
struct Accumulator {
    snpgpu_ctx *ctx = nullptr;
    C_UInt8 *blk[2] = {nullptr, nullptr};      // page-locked block buffers of the reader
    ~Accumulator()
    {
        if (ctx) snpgpu_destroy(ctx);           // waits for the stream, so the buffers are idle afterwards
        for (int k = 0; k < 2; k++)
            if (blk[k]) snpgpu_host_free(blk[k]);
    }

    // Streams the working space (set up by .InitFile2 -> gnrSetGenoSpace / gnrSelSNP_Base, exactly as before)
    // through a context of `kind`: the reader loop of every CXxx::Run without its body.
    void stream(int kind, bool bayesian, size_t block_snps, bool verbose)
    {
        CdBaseWorkSpace &space = MCWorkingGeno.Space();
        const size_t n_samp = space.SampleNum();
        snpgpu_opts o;
        memset(&o, 0, sizeof(o));
        o.device = opt_int("snpgpu.device", "SNPGPU_DEVICE", 0);
        o.bayesian = bayesian ? 1 : 0;
        o.max_block_snps = (int64_t)block_snps;
        if (snpgpu_create(kind, (int64_t)n_samp, &o, &ctx)) gpu_fail();
        for (int k = 0; k < 2; k++)
            if (snpgpu_host_alloc(n_samp * block_snps, (void **)&blk[k])) gpu_fail();
        // kept reader: uint8 [B][n_samp], sample fastest, values > 3 clamped to 3; its one prefetch thread stays
        CGenoReadBySNP reader(1, space, block_snps, verbose ? -1 : 0, false);
        reader.Init();
        for (int k = 0;; k ^= 1) {
            if (snpgpu_host_wait(ctx, blk[k])) gpu_fail();       // the copy of two blocks ago has left this buffer
            if (!reader.Read(blk[k])) break;
            // asynchronous: the H2D copy of this block overlaps the kernels of the previous one
            if (snpgpu_feed(ctx, blk[k], (int64_t)reader.Count(), SNPGPU_GENO_U8, SNPGPU_HOST_PINNED)) gpu_fail();
            reader.ProgressForward(reader.Count());              // progress + R_CheckUserInterrupt as before
        }
        if (snpgpu_sync(ctx)) gpu_fail();
    }
};

// options(snpgpu.devices = c(0, 1, ...)) / SNPGPU_DEVICES="0,1,...": more than one entry selects the multi-device object
// (snpgpu_multi: ONE R process drives all listed GPUs -- row panels of the output triangle, the block copied to the
// first device once and forwarded over xGMI, packed slabs gathered into the R vector)
std::vector<int32_t> opt_devices()
{
    std::vector<int32_t> d;
    SEXP v = Rf_GetOption1(Rf_install("snpgpu.devices"));
    if (v != R_NilValue && Rf_length(v) > 0) {
        SEXP iv = PROTECT(Rf_coerceVector(v, INTSXP));
        for (R_xlen_t i = 0; i < Rf_xlength(iv); i++) d.push_back(INTEGER(iv)[i]);
        UNPROTECT(1);
    } else if (const char *e = getenv("SNPGPU_DEVICES")) {
        for (const char *p = e; *p;) { d.push_back((int32_t)strtol(p, (char **)&p, 10)); while (*p == ',' || *p == ' ') p++; }
    }
    return d;
}

// The same reader loop over a multi-device object (`pass` of `n_passes`: KING-robust's 20 B of counters per pair may need
// several walks over the SNPs, each with its own resident panels -- snpgpu_multi_opts).
struct MultiAccumulator {
    snpgpu_multi *m = nullptr;
    C_UInt8 *blk[2] = {nullptr, nullptr};
    ~MultiAccumulator()
    {
        if (m) snpgpu_multi_destroy(m);
        for (int k = 0; k < 2; k++)
            if (blk[k]) snpgpu_host_free(blk[k]);
    }
    // panels_per_device: 0 = options(snpgpu.panels.per.device=) / SNPGPU_PANELS_PER_DEVICE, default -1 (the library picks the fewest
    // that fit the devices' free memory).  A job of several passes must cut the triangle the SAME way in every pass -- the plan is
    // plan_rows(n, devices x panels_per_device x passes) --, and the automatic choice reads hipMemGetInfo, which may differ from
    // one pass to the next: pass 0 resolves it, chosen_ppd() hands the value to the later passes.
    int chosen_ppd() const
    {
        snpgpu_multi_status st;
        if (!m || snpgpu_multi_get_status(m, &st)) gpu_fail();
        return (int)st.panels_per_device;
    }
    void stream(int kind, bool bayesian, size_t block_snps, bool verbose, const std::vector<int32_t> &devices, int n_passes = 1, int pass = 0,
                int panels_per_device = 0)
    {
        CdBaseWorkSpace &space = MCWorkingGeno.Space();
        const size_t n_samp = space.SampleNum();
        snpgpu_opts o;
        memset(&o, 0, sizeof(o));
        o.bayesian = bayesian ? 1 : 0;
        o.max_block_snps = (int64_t)block_snps;
        snpgpu_multi_opts mo;
        memset(&mo, 0, sizeof(mo));
        mo.devices = &devices[0];
        mo.n_devices = (int32_t)devices.size();
        mo.panels_per_device = panels_per_device != 0 ? panels_per_device
                                                      : opt_int("snpgpu.panels.per.device", "SNPGPU_PANELS_PER_DEVICE", -1);   // -1: the fewest that fit (library)
        mo.n_passes = n_passes;
        mo.pass = pass;
        if (snpgpu_multi_create(kind, (int64_t)n_samp, &o, &mo, &m)) gpu_fail();
        for (int k = 0; k < 2; k++)
            if (!blk[k] && snpgpu_host_alloc(n_samp * block_snps, (void **)&blk[k])) gpu_fail();
        CGenoReadBySNP reader(1, space, block_snps, verbose ? -1 : 0, false);
        reader.Init();
        for (int k = 0;; k ^= 1) {
            if (snpgpu_multi_host_wait(m, blk[k])) gpu_fail();
            if (!reader.Read(blk[k])) break;
            if (snpgpu_multi_feed(m, blk[k], (int64_t)reader.Count(), SNPGPU_GENO_U8, SNPGPU_HOST_PINNED)) gpu_fail();
            reader.ProgressForward(reader.Count());
        }
        if (snpgpu_multi_sync(m)) gpu_fail();
    }
};

// packed upper triangle (row-major with diagonal) -> full symmetric column-major matrix, in place at the end of `full`
void tri_to_full(const std::vector<double> &tri, size_t n, double *full)
{
    for (size_t i = 0, k = 0; i < n; i++)
        for (size_t j = i; j < n; j++, k++) full[i * n + j] = full[j * n + i] = tri[k];
}

size_t counter_block() { return (size_t)opt_int("snpgpu.block.snps", "SNPGPU_BLOCK_SNPS", 65536); }   // IBS / KING / beta
size_t syrk_block() { return (size_t)opt_int("snpgpu.block.snps", "SNPGPU_BLOCK_SNPS", 32768); }      // GRM / PCA / EIGMIX (bench.py feeds 65536-SNP blocks of 2-bit rows; the byte genotypes of the kept reader make that 6.5 GB per buffer at N = 100 000, hence half of it here)

// n x n REALSXP matrix or the packed upper triangle as a plain numeric vector (useMatrix = TRUE: R wraps it with
// Matrix::dspMatrix(uplo = "L"), R/Internal.R:46-51 -- column-major lower == row-major upper, the CdMatTri order)
SEXP alloc_result(size_t n, bool packed)
{
    return packed ? Rf_allocVector(REALSXP, (R_xlen_t)(n * (n + 1) / 2)) : Rf_allocMatrix(REALSXP, (int)n, (int)n);
}

// snpgdsGRM(out.fn=): the finished matrix goes row by row into the "grm" node the R wrapper created
// (valdim = c(n, 0), R/IBD.R:586-589).  `tri` is the packed upper triangle; row i of the symmetric matrix is
// {tri(0,i) .. tri(i-1,i), tri(i,i) .. tri(i,n-1)}.
void append_rows(const std::vector<double> &tri, size_t n, PdGDSObj node, bool verbose)
{
    if (verbose) Rprintf("Saving to the GDS file:\n");
    std::vector<double> row(n);
    CProgress progress(verbose ? (C_Int64)n : -1);
    for (size_t i = 0; i < n; i++) {
        for (size_t k = 0; k < i; k++) row[k] = tri[i + k * (2 * n - k - 1) / 2];       // idx(k, i), k < i
        const double *p = &tri[i + i * (2 * n - i - 1) / 2];                            // idx(i, i)
        for (size_t j = i; j < n; j++) row[j] = p[j - i];
        GDS_Array_AppendData(node, (int)n, &row[0], svFloat64);
        progress.Forward(1);
    }
}

double grm_avg_val = 0;     // value of the last IndivBeta run (the reference keeps the same global, genPCA.cpp:1605)

}  // namespace

extern "C" {

// --------------------------------------------------------------------------------------------------------------
COREARRAY_DLL_EXPORT SEXP gpu_gnrIBSNum(SEXP NumThread, SEXP Verbose)
{
    const bool verbose = SEXP_Verbose(Verbose);
    COREARRAY_TRY
        CachingSNPData("IBS", verbose);
        const size_t n = MCWorkingGeno.Space().SampleNum();
        Accumulator acc;
        acc.stream(SNPGPU_IBS, false, counter_block(), verbose);
        PROTECT(rv_ans = Rf_allocVector(VECSXP, 3));
        for (int k = 0; k < 3; k++) SET_VECTOR_ELT(rv_ans, k, Rf_allocMatrix(INTSXP, (int)n, (int)n));
        // three full symmetric int32 matrices, written straight into R's memory
        if (snpgpu_ibs_num(acc.ctx, INTEGER(VECTOR_ELT(rv_ans, 0)), INTEGER(VECTOR_ELT(rv_ans, 1)),
                           INTEGER(VECTOR_ELT(rv_ans, 2)), 0, SNPGPU_HOST))
            gpu_fail();
        if (verbose) Rprintf("%s    Done.\n", TimeToStr());
        UNPROTECT(1);
    COREARRAY_CATCH
}

COREARRAY_DLL_EXPORT SEXP gpu_gnrIBSAve(SEXP NumThread, SEXP useMatrix, SEXP Verbose)
{
    const bool verbose = SEXP_Verbose(Verbose);
    COREARRAY_TRY
        CachingSNPData("IBS", verbose);
        const size_t n = MCWorkingGeno.Space().SampleNum();
        const bool packed = (Rf_asLogical(useMatrix) == TRUE);
        Accumulator acc;
        acc.stream(SNPGPU_IBS, false, counter_block(), verbose);
        PROTECT(rv_ans = alloc_result(n, packed));
        if (snpgpu_ibs_ave(acc.ctx, REAL(rv_ans), packed ? 1 : 0, SNPGPU_HOST)) gpu_fail();
        if (verbose) Rprintf("%s    Done.\n", TimeToStr());
        UNPROTECT(1);
    COREARRAY_CATCH
}

COREARRAY_DLL_EXPORT SEXP gpu_gnrIBD_KING_Robust(SEXP FamilyID, SEXP NumThread, SEXP useMatrix, SEXP Verbose)
{
    const bool verbose = SEXP_Verbose(Verbose);
    COREARRAY_TRY
        CachingSNPData("KING IBD", verbose);
        // SumSq <= 4 L must fit uint32 (src/genKING.cpp:598-602); snpgpu_feed enforces the same bound per stream,
        // checked here first so that no device memory is touched for an invalid request
        if (MCWorkingGeno.Space().SNPNum() >= 1073741824)
            throw ErrCoreArray("The number of SNPs should be less than 1,073,741,824.");
        const size_t n = MCWorkingGeno.Space().SampleNum();
        const bool packed = (Rf_asLogical(useMatrix) == TRUE);
        const std::vector<int32_t> devices = opt_devices();
        if (devices.size() > 1) {
            // several GPUs, and as many passes over the SNP stream as options(snpgpu.passes=) asks for: every pass gathers
            // the packed slabs of its own panels into the same two vectors
            const int n_passes = opt_int("snpgpu.passes", "SNPGPU_PASSES", 1);
            std::vector<double> t0, t1;
            PROTECT(rv_ans = Rf_allocVector(VECSXP, 2));
            SET_VECTOR_ELT(rv_ans, 0, alloc_result(n, packed));
            SET_VECTOR_ELT(rv_ans, 1, alloc_result(n, packed));
            double *o0 = REAL(VECTOR_ELT(rv_ans, 0)), *o1 = REAL(VECTOR_ELT(rv_ans, 1));
            if (!packed) { t0.resize(n * (n + 1) / 2); t1.resize(n * (n + 1) / 2); }
            int ppd = 0;                                   // resolved by pass 0, the same for every later pass
            for (int q = 0; q < n_passes; q++) {
                MultiAccumulator acc;
                acc.stream(SNPGPU_KING_ROBUST, false, counter_block(), verbose, devices, n_passes, q, ppd);
                if (q == 0) ppd = acc.chosen_ppd();
                if (snpgpu_multi_king_robust(acc.m, INTEGER(FamilyID), packed ? o0 : &t0[0], packed ? o1 : &t1[0], SNPGPU_HOST)) gpu_fail();
            }
            if (!packed) { tri_to_full(t0, n, o0); tri_to_full(t1, n, o1); }
            if (verbose) Rprintf("%s    Done.\n", TimeToStr());
            UNPROTECT(1);
            return rv_ans;
        }
        Accumulator acc;
        acc.stream(SNPGPU_KING_ROBUST, false, counter_block(), verbose);
        PROTECT(rv_ans = Rf_allocVector(VECSXP, 2));
        SET_VECTOR_ELT(rv_ans, 0, alloc_result(n, packed));      // IBS0
        SET_VECTOR_ELT(rv_ans, 1, alloc_result(n, packed));      // kinship
        // FamilyID: as.integer(as.factor(family.id)) from R/IBD.R:360-364 -- levels 1..k, NA_INTEGER (= INT_MIN,
        // negative) for "no family", which is the ABI's convention for NA
        if (snpgpu_king_robust(acc.ctx, INTEGER(FamilyID), REAL(VECTOR_ELT(rv_ans, 0)), REAL(VECTOR_ELT(rv_ans, 1)),
                               packed ? 1 : 0, SNPGPU_HOST))
            gpu_fail();
        if (verbose) Rprintf("%s    Done.\n", TimeToStr());
        UNPROTECT(1);
    COREARRAY_CATCH
}

COREARRAY_DLL_EXPORT SEXP gpu_gnrIBD_KING_Homo(SEXP NumThread, SEXP useMatrix, SEXP Verbose)
{
    const bool verbose = SEXP_Verbose(Verbose);
    COREARRAY_TRY
        CachingSNPData("KING IBD", verbose);
        const size_t n = MCWorkingGeno.Space().SampleNum();
        const bool packed = (Rf_asLogical(useMatrix) == TRUE);
        Accumulator acc;
        acc.stream(SNPGPU_KING_HOMO, false, syrk_block(), verbose);   // integer half + the two masked fp sums
        PROTECT(rv_ans = Rf_allocVector(VECSXP, 2));
        SET_VECTOR_ELT(rv_ans, 0, alloc_result(n, packed));      // k0
        SET_VECTOR_ELT(rv_ans, 1, alloc_result(n, packed));      // k1
        if (snpgpu_king_homo(acc.ctx, REAL(VECTOR_ELT(rv_ans, 0)), REAL(VECTOR_ELT(rv_ans, 1)), packed ? 1 : 0, SNPGPU_HOST))
            gpu_fail();
        if (verbose) Rprintf("%s    Done.\n", TimeToStr());
        UNPROTECT(1);
    COREARRAY_CATCH
}

// --------------------------------------------------------------------------------------------------------------
COREARRAY_DLL_EXPORT SEXP gpu_gnrGRM_avg_val() { return Rf_ScalarReal(grm_avg_val); }

COREARRAY_DLL_EXPORT SEXP gpu_gnrGRM(SEXP NumThread, SEXP Method, SEXP GDS, SEXP useMatrix, SEXP Verbose)
{
    const char *method = CHAR(STRING_ELT(Method, 0));
    const bool verbose = SEXP_Verbose(Verbose);
    COREARRAY_TRY
        PdGDSObj node = NULL;                                  // snpgdsGRM(out.fn=): the "grm" node of the output file
        if (!Rf_isNull(GDS)) node = GDS_Node_Path(GDS_R_SEXP2FileRoot(GDS), "grm", TRUE);
        CachingSNPData("GRM Calculation", verbose);
        const size_t n = MCWorkingGeno.Space().SampleNum();

        int kind;
        if (strcmp(method, "GCTA") == 0 || strcmp(method, "Corr") == 0) kind = SNPGPU_GRM_GCTA;
        else if (strcmp(method, "Eigenstrat") == 0) kind = SNPGPU_PCA_COV;
        else if (strcmp(method, "EIGMIX") == 0) kind = SNPGPU_EIGMIX;       // "Weighted" arrives as "EIGMIX" (R/IBD.R:551-555)
        else if (strcmp(method, "IndivBeta") == 0) kind = SNPGPU_INDIV_BETA;
        else throw ErrCoreArray("Invalid 'method'!");

        const std::vector<int32_t> devices = opt_devices();
        if (devices.size() > 1 && (kind == SNPGPU_GRM_GCTA || kind == SNPGPU_PCA_COV) && strcmp(method, "Corr") != 0) {
            MultiAccumulator acc;
            acc.stream(kind, false, syrk_block(), verbose, devices);
            const bool packed = node ? true : (Rf_asLogical(useMatrix) == TRUE);
            std::vector<double> tri;
            if (node || !packed) tri.resize(n * (n + 1) / 2);
            if (!node) PROTECT(rv_ans = alloc_result(n, packed));
            double *out = (node || !packed) ? &tri[0] : REAL(rv_ans);
            if (kind == SNPGPU_GRM_GCTA ? snpgpu_multi_grm_gcta(acc.m, out, SNPGPU_HOST)
                                        : snpgpu_multi_pca_cov(acc.m, out, 1, NULL, SNPGPU_HOST))
                gpu_fail();
            if (node) append_rows(tri, n, node, verbose);
            else if (!packed) tri_to_full(tri, n, REAL(rv_ans));
            if (verbose) Rprintf("%s    Done.\n", TimeToStr());
            if (!node) UNPROTECT(1);
            return rv_ans;
        }
        Accumulator acc;
        acc.stream(kind, false, kind == SNPGPU_INDIV_BETA ? counter_block() : syrk_block(), verbose);

        // "Corr" is always a full matrix (src/genPCA.cpp:1658-1685); a GDS target takes the packed triangle
        const bool corr = (strcmp(method, "Corr") == 0);
        const bool packed = node ? true : (!corr && Rf_asLogical(useMatrix) == TRUE);
        std::vector<double> tri;
        double *out;
        if (node) {
            tri.resize(n * (n + 1) / 2);
            out = &tri[0];
        } else {
            PROTECT(rv_ans = alloc_result(n, packed));
            out = REAL(rv_ans);
        }
        int rc;
        switch (kind) {
        case SNPGPU_GRM_GCTA: rc = snpgpu_grm_gcta(acc.ctx, out, packed ? 1 : 0, SNPGPU_HOST); break;
        case SNPGPU_PCA_COV:   // Eigenstrat: covariance times (n - 1) / trace
            rc = snpgpu_pca_cov(acc.ctx, out, packed ? 1 : 0, 1, 0.0, NULL, SNPGPU_HOST); break;
        case SNPGPU_EIGMIX:    // CalcEigMixGRM: no diagonal adjustment, times 2 (src/genEIGMIX.cpp:645-653)
            rc = snpgpu_eigmix(acc.ctx, 0, 2.0, out, packed ? 1 : 0, SNPGPU_HOST); break;
        default:               // CalcIndivBetaGRM: min-based transform, keeps the average for gnrGRM_avg_val
            rc = snpgpu_indiv_beta(acc.ctx, 2, out, &grm_avg_val, packed ? 1 : 0, SNPGPU_HOST); break;
        }
        if (rc) gpu_fail();

        if (corr) {            // scaled GRM: unit diagonal, g_ij / sqrt(g_ii g_jj)
            std::vector<double> sd(n);
            for (size_t i = 0; i < n; i++) sd[i] = sqrt(out[i * n + i]);
            for (size_t i = 0; i < n; i++) {
                out[i * n + i] = 1;
                for (size_t j = i + 1; j < n; j++) out[i * n + j] = out[j * n + i] = out[j * n + i] / (sd[i] * sd[j]);
            }
        }
        if (node) append_rows(tri, n, node, verbose);
        if (verbose) Rprintf("%s    Done.\n", TimeToStr());
        if (!node) UNPROTECT(1);
    COREARRAY_CATCH
}

// --------------------------------------------------------------------------------------------------------------
COREARRAY_DLL_EXPORT SEXP gpu_gnrPCA(SEXP EigenCnt, SEXP Algorithm, SEXP NumThread, SEXP ParamList, SEXP Verbose)
{
    const bool verbose = SEXP_Verbose(Verbose);
    COREARRAY_TRY
        CachingSNPData("PCA", verbose);
        const char *alg = CHAR(STRING_ELT(Algorithm, 0));
        const size_t n = MCWorkingGeno.Space().SampleNum();

        if (strcmp(alg, "exact") == 0) {
            const bool bayesian = (Rf_asLogical(RGetListElement(ParamList, "bayesian")) == TRUE);
            const bool need_genmat = (Rf_asLogical(RGetListElement(ParamList, "need.genmat")) == TRUE);
            const bool genmat_only = (Rf_asLogical(RGetListElement(ParamList, "genmat.only")) == TRUE);
            const std::vector<int32_t> devices = opt_devices();
            // eigen.method is validated on every route ("DSPEV" = all eigenvalues, src/genPCA.cpp:1262-1346): a request for the
            // whole spectrum is a dense problem -- it takes the single-device route below, which returns all n eigenvalues as the
            // reference does, also when several devices are configured
            const char *em_all = CHAR(STRING_ELT(RGetListElement(ParamList, "eigen.method"), 0));
            if (strcmp(em_all, "DSPEV") != 0 && strcmp(em_all, "DSPEVX") != 0) throw ErrCoreArray("Unknown 'eigen.method'.");
            if (devices.size() > 1 && (genmat_only || strcmp(em_all, "DSPEV") != 0)) {
                // configs[3]: the covariance stays distributed over the GPUs as row panels; only the trace, (on request) the
                // matrix, and the top eigenpairs -- block Krylov over all devices, snpgpu_multi_topk_eigen -- come back
                MultiAccumulator acc;
                acc.stream(SNPGPU_PCA_COV, bayesian, syrk_block(), verbose, devices);
                PROTECT(rv_ans = Rf_allocVector(VECSXP, 5));
                double trace_xtx = 0;
                if (need_genmat) {
                    std::vector<double> tri(n * (n + 1) / 2);
                    if (snpgpu_multi_pca_cov(acc.m, &tri[0], 1, &trace_xtx, SNPGPU_HOST)) gpu_fail();
                    SET_VECTOR_ELT(rv_ans, 1, Rf_allocMatrix(REALSXP, (int)n, (int)n));
                    tri_to_full(tri, n, REAL(VECTOR_ELT(rv_ans, 1)));
                } else if (snpgpu_multi_pca_cov(acc.m, NULL, 1, &trace_xtx, SNPGPU_HOST)) gpu_fail();
                SET_VECTOR_ELT(rv_ans, 0, Rf_ScalarReal(trace_xtx));
                SET_VECTOR_ELT(rv_ans, 4, Rf_ScalarReal((double)(n - 1)));
                if (!genmat_only) {
                    int n_eig = Rf_asInteger(EigenCnt);
                    if (n_eig < 0) throw ErrCoreArray("Invalid 'eigen.cnt'.");
                    if ((size_t)n_eig > n) n_eig = (int)n;
                    if (n_eig > 0) {
                        SET_VECTOR_ELT(rv_ans, 2, Rf_allocVector(REALSXP, (R_xlen_t)n));
                        SET_VECTOR_ELT(rv_ans, 3, Rf_allocMatrix(REALSXP, (int)n, n_eig));
                        double *val = REAL(VECTOR_ELT(rv_ans, 2));
                        // scale <= 0: (n - 1) / trace over all panels, src/genPCA.cpp:1386-1390
                        if (snpgpu_multi_topk_eigen(acc.m, 0.0, n_eig, NULL, val, REAL(VECTOR_ELT(rv_ans, 3)), SNPGPU_HOST, NULL)) gpu_fail();
                        for (size_t i = (size_t)n_eig; i < n; i++) val[i] = R_NaN;
                    }
                }
                if (verbose) Rprintf("%s    Done.\n", TimeToStr());
                UNPROTECT(1);
                return rv_ans;
            }
            Accumulator acc;
            acc.stream(SNPGPU_PCA_COV, bayesian, syrk_block(), verbose);

            PROTECT(rv_ans = Rf_allocVector(VECSXP, 5));         // TraceXTX, genmat, eigenval, eigenvect, TraceVal
            double trace_xtx = 0;
            double *genmat = NULL;
            if (need_genmat) {
                SET_VECTOR_ELT(rv_ans, 1, Rf_allocMatrix(REALSXP, (int)n, (int)n));
                genmat = REAL(VECTOR_ELT(rv_ans, 1));
            }
            // covariance times (n - 1) / trace; genmat == NULL only returns the trace
            if (snpgpu_pca_cov(acc.ctx, genmat, 0, 1, 0.0, &trace_xtx, SNPGPU_HOST)) gpu_fail();
            SET_VECTOR_ELT(rv_ans, 0, Rf_ScalarReal(trace_xtx));
            // trace after the scaling: n - 1 up to rounding; taken from the matrix when it is there, as the reference does
            double trace_val = (double)(n - 1);
            if (genmat) { trace_val = 0; for (size_t i = 0; i < n; i++) trace_val += genmat[i * n + i]; }
            SET_VECTOR_ELT(rv_ans, 4, Rf_ScalarReal(trace_val));

            if (!genmat_only) {
                if (verbose) Rprintf("%s    Begin (eigenvalues and eigenvectors)\n", TimeToStr());
                int n_eig = Rf_asInteger(EigenCnt);
                if (n_eig < 0) throw ErrCoreArray("Invalid 'eigen.cnt'.");
                if ((size_t)n_eig > n) n_eig = (int)n;
                if (n_eig > 0) {
                    const char *em = CHAR(STRING_ELT(RGetListElement(ParamList, "eigen.method"), 0));
                    const bool all = (strcmp(em, "DSPEV") == 0);               // all eigenvalues, first n_eig vectors
                    if (!all && strcmp(em, "DSPEVX") != 0) throw ErrCoreArray("Unknown 'eigen.method'.");
                    const int k = all ? (int)n : n_eig;
                    SET_VECTOR_ELT(rv_ans, 2, Rf_allocVector(REALSXP, (R_xlen_t)n));
                    SET_VECTOR_ELT(rv_ans, 3, Rf_allocMatrix(REALSXP, (int)n, n_eig));
                    double *val = REAL(VECTOR_ELT(rv_ans, 2));
                    std::vector<double> vec_all;
                    double *vec = REAL(VECTOR_ELT(rv_ans, 3));
                    if (all) { vec_all.resize(n * n); vec = &vec_all[0]; }
                    // top-k eigenpairs of the normalised covariance on the device (descending, as -dspevx(-C) gives
                    // them): hipSOLVER's dense solver up to SNPGPU_EIG_DENSE_MAX samples, the block-Krylov solver of
                    // csrc/eigen.hip on the resident panel beyond (any n); a failure carries LAPACK's wording, "... infinite or missing values in the genetic
                    // covariance matrix!"
                    if (snpgpu_pca_eigen(acc.ctx, k, val, vec, SNPGPU_HOST)) gpu_fail();
                    for (size_t i = (size_t)k; i < n; i++) val[i] = R_NaN;      // CalcEigen :1343-1345
                    if (all) memcpy(REAL(VECTOR_ELT(rv_ans, 3)), vec, sizeof(double) * n * (size_t)n_eig);
                }
            }
            UNPROTECT(1);

        } else if (strcmp(alg, "randomized") == 0) {
            // CRandomPCA::Run makes iter.num + 2 passes over the SNPs; here the selected genotypes are read ONCE
            // through the kept reader, kept as 2-bit rows in host memory (N L / 4 bytes) and handed to the
            // workspace-level routine, which runs every pass on the device.
            const int aux_dim = Rf_asInteger(RGetListElement(ParamList, "aux.dim"));
            const int iter_num = Rf_asInteger(RGetListElement(ParamList, "iter.num"));
            const double *aux_mat = REAL(RGetListElement(ParamList, "aux.mat"));    // rnorm(aux.dim * n.samp)
            const int n_eig = Rf_asInteger(EigenCnt);
            CdBaseWorkSpace &space = MCWorkingGeno.Space();
            const size_t n_snp = space.SNPNum(), rb = (n + 3) / 4, blk = 4096;
            std::vector<C_UInt8> packed(n_snp * rb), buf(n * blk);
            {
                CGenoReadBySNP reader(1, space, blk, verbose ? -1 : 0, false);
                reader.Init();
                size_t at = 0;
                while (reader.Read(&buf[0])) {
                    for (size_t s = 0; s < reader.Count(); s++, at++) {
                        const C_UInt8 *g = &buf[s * n];
                        C_UInt8 *q = &packed[at * rb];
                        for (size_t i = 0; i < n; i++) {
                            if ((i & 3) == 0) q[i >> 2] = 0;
                            q[i >> 2] |= (C_UInt8)((g[i] > 3 ? 3 : g[i]) << (2 * (i & 3)));
                        }
                        if (n & 3) q[rb - 1] |= (C_UInt8)(0xFF << (2 * (n & 3)));      // padding = missing
                    }
                    reader.ProgressForward(reader.Count());
                }
            }
            const int device = opt_int("snpgpu.device", "SNPGPU_DEVICE", 0);
            struct WsGuard { ~WsGuard() { snpgpu_ws_clear(); } } guard;
            if (snpgpu_ws_set_geno(&packed[0], (int64_t)n_snp, (int64_t)n, SNPGPU_GENO_PACKED2, device)) gpu_fail();
            const int hsize = aux_dim * (iter_num + 1);
            std::vector<double> vecs(n * (size_t)n_eig);
            double trace2 = 0;
            PROTECT(rv_ans = Rf_allocVector(VECSXP, 3));         // sigma [n], V^T [hsize x n], 2 * TraceXTX
            SET_VECTOR_ELT(rv_ans, 0, Rf_allocVector(REALSXP, (R_xlen_t)n));
            SET_VECTOR_ELT(rv_ans, 1, Rf_allocMatrix(REALSXP, hsize, (int)n));
            if (snpgpu_gnrPCA_randomized(n_eig, aux_dim, iter_num, aux_mat, 1, verbose ? 1 : 0,
                                         REAL(VECTOR_ELT(rv_ans, 0)), &vecs[0], &trace2))
                gpu_fail();
            // R/PCA.R:84 uses the first eigen.cnt ROWS of V^T: element (r, s) of an hsize x n column-major matrix
            double *vt = REAL(VECTOR_ELT(rv_ans, 1));
            memset(vt, 0, sizeof(double) * (size_t)hsize * n);
            for (int r = 0; r < n_eig; r++)
                for (size_t s = 0; s < n; s++) vt[(size_t)r + (size_t)hsize * s] = vecs[s + n * (size_t)r];
            SET_VECTOR_ELT(rv_ans, 2, Rf_ScalarReal(trace2));
            UNPROTECT(1);
        } else
            throw "Invalid 'algorithm'.";

        if (verbose) Rprintf("%s    Done.\n", TimeToStr());
    COREARRAY_CATCH
}

}  // extern "C"
