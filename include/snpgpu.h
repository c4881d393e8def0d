/*
 * snpgpu.h -- C ABI of libsnpgpu: MI355X (gfx950) pairwise relatedness kernels
 * behind SNPRelate's snpgds* hot path.
 *
 * Plain C, no SEXP / no torch types: pointers, sizes, status codes.  Every
 * entry point names the reference interface it replaces (paths relative to
 * the SNPRelate source tree, v1.46.0).  The R-side `.Call` shim that binds
 * these for a drop-in build of the package is shown in INTEGRATION.md.
 *
 * Two levels:
 *   (1) streaming accumulators  (snpgpu_create / _feed / finalisers / _destroy)
 *       replace the `CXxx::Run(CdMatTri<T>&, NumThread, verbose)` algorithm
 *       classes; the caller keeps the reference's GDS block reader
 *       (CGenoReadBySNP, src/dGenGWAS.cpp:1218-1397) and hands each block over;
 *   (2) workspace calls (snpgpu_ws_* / snpgpu_gnr*) mirror the registered
 *       `.Call` routines one to one (src/SNPRelate.cpp:1154-1205) on an
 *       in-memory genotype matrix, for hosts without gdsfmt (tests, Python).
 *
 * Conventions
 *   genotypes : SNP-major blocks, sample fastest (what CGenoReadBySNP::Read
 *               returns): SNPGPU_GENO_U8      uint8 [n_snp][n_samp], >2 = missing
 *                         SNPGPU_GENO_PACKED2 uint8 [n_snp][ceil(n_samp/4)],
 *                           4 genotypes/byte LSB first, 3 = missing (GDS bit2)
 *   triangles : packed upper, row-major with diagonal (CdMatTri,
 *               src/dGenGWAS.h:511-583): idx(i,j) = j + i(2N-i-1)/2, i <= j
 *   matrices  : full symmetric n x n, column-major == row-major
 *   status    : 0 = ok, non-zero = error; message via snpgpu_last_error()
 *   memory    : `mem` says whether a caller pointer is host or device memory
 *               (device pointers must belong to the context's device)
 *   threading : one context is used from one host thread at a time; calls
 *               block until the result is in the caller's buffer.
 */
#ifndef SNPGPU_H
#define SNPGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNPGPU_ABI_VERSION 2

typedef struct snpgpu_ctx snpgpu_ctx;

enum snpgpu_kind {
    SNPGPU_IBS         = 1, /* CIBSCount        src/genIBS.cpp:145-329  */
    SNPGPU_KING_ROBUST = 2, /* CKINGRobust      src/genKING.cpp:283-482 */
    SNPGPU_KING_HOMO   = 3, /* CKINGHomo        src/genKING.cpp:58-266  */
    SNPGPU_GRM_GCTA    = 4, /* CGCTA_AlgArith   src/genPCA.cpp:1131-1238 */
    SNPGPU_PCA_COV     = 5, /* CExactPCA        src/genPCA.cpp:378-465 (also GRM "Eigenstrat") */
    SNPGPU_EIGMIX      = 6, /* CEigMix_AlgArith src/genEIGMIX.cpp:43-160 (also GRM "EIGMIX") */
    SNPGPU_INDIV_BETA  = 7  /* CIndivBeta       src/genBeta.cpp:57-252 (also GRM "IndivBeta") */
};

enum snpgpu_geno_format { SNPGPU_GENO_U8 = 0, SNPGPU_GENO_PACKED2 = 1 };
enum snpgpu_mem {
    SNPGPU_HOST = 0,        /* pageable host memory: the call returns when the block has been copied   */
    SNPGPU_DEVICE = 1,      /* device memory of the context's device                                   */
    SNPGPU_HOST_PINNED = 2  /* memory from snpgpu_host_alloc: snpgpu_feed only enqueues an asynchronous
                               copy on a second stream (double-buffered on the device), so the copy of
                               block k+1 overlaps the kernels of block k; call snpgpu_host_wait before
                               refilling the same buffer                                               */
};

typedef struct snpgpu_opts {
    int32_t device;         /* HIP device ordinal                                   */
    int32_t bayesian;       /* PCA_COV only: Bayesian normalisation, genPCA.cpp:441-453 */
    int64_t row_begin;      /* output panel = sample rows [row_begin,row_end) x cols>=row */
    int64_t row_end;        /*   0,0 = whole triangle. row_begin must be a multiple of 256 */
    int64_t max_block_snps; /* largest n_snp a single snpgpu_feed will pass (0 = 32768) */
    void   *stream;         /* hipStream_t to run on, or NULL for the context's own  */
} snpgpu_opts;

/* ---- library ---------------------------------------------------------- */
int         snpgpu_abi_version(void);
const char *snpgpu_last_error(void);  /* thread-local; replaces gnrErrMsg, src/SNPRelate.cpp:1099 */
int         snpgpu_device_count(int *count);

/* Synthetic genotype blocks for benchmarks and full-size parity tests (SURVEY.md 8(d) generator; no reference
 * counterpart): writes SNPs [snp_begin, snp_begin + n_snp) of the seeded data set as SNPGPU_GENO_PACKED2 rows
 * [n_snp][ceil(n_samp/4)] into DEVICE memory `dst`.  Counter-based: every cell is a pure integer function of
 * (seed, snp, sample), identical on every GPU and re-computable for single samples on the CPU
 * (oracle/synth.py).  spectrum 0: per-SNP p ~ U(0.05, 0.95); 1: p = u^3 / 2 (rare variants); 2: p ~ U(0.01, 0.5).
 * 3: three sub-populations (sample % 3) with Fst ~ 0.1 around an ancestral p ~ U(0.05, 0.95); 4: linkage disequilibrium --
 * blocks of 48 consecutive SNPs copied from 6 founder haplotypes per block, 2 % of the haplotype alleles drawn independently.
 * missing: iid missing-call rate.  special != 0 plants monomorphic / all-missing SNPs (snp % 997 in {3, 5, 7}).
 * stream: hipStream_t or NULL (the call then synchronises the device before -- an earlier asynchronous snpgpu_feed may
 * still be reading `dst` -- and after writing the block). */
int snpgpu_synth_block(void *dst, int64_t n_samp, int64_t snp_begin, int64_t n_snp, uint32_t seed, double missing,
                       int spectrum, int special, int device, void *stream);

/* ---- (1) streaming accumulators --------------------------------------- */
/* replaces the construction + memset part of CXxx::Run
 * (e.g. src/genIBS.cpp:280-299) */
int snpgpu_create(int kind, int64_t n_samp, const snpgpu_opts *opts, snpgpu_ctx **out);
int snpgpu_destroy(snpgpu_ctx *ctx);

/* replaces one iteration of `while (WS.Read(Geno))` { pack / centre ; BatchWork }
 * (src/genIBS.cpp:312-327, src/genKING.cpp:465-480, src/genPCA.cpp:428-462, :1185-1230).
 * Asynchronous w.r.t. the device when `mem` is SNPGPU_DEVICE: the block must stay allocated and unchanged until the pre-pass
 * has read it -- snpgpu_sync, or any call that orders after the context's stream. */
int snpgpu_feed(snpgpu_ctx *ctx, const void *geno, int64_t n_snp, int format, int mem);
/* The block's per-SNP statistics (sum of called genotypes, number of calls over ALL n_samp samples; vec_u8_geno_count,
 * src/dVect.cpp:30-117) computed ONCE per node instead of once per rank: every rank calls snpgpu_block_stats on its share of the
 * block's SNP rows (device memory, rows of the given format; sum / num: device arrays of n_snp int32), all-gathers the two arrays
 * (8 bytes per SNP) and feeds the whole block with snpgpu_feed_stats, which then only re-lays the rows out (sum / num: device
 * memory).  Same results as snpgpu_feed bit for bit (integer statistics).  snprelate_amd/multigpu.py: shared_stats=True. */
int snpgpu_block_stats(snpgpu_ctx *ctx, const void *geno, int64_t n_snp, int format, int32_t *sum, int32_t *num);
int snpgpu_feed_stats(snpgpu_ctx *ctx, const void *geno, int64_t n_snp, int format, int mem, const int32_t *sum, const int32_t *num);
int snpgpu_sync(snpgpu_ctx *ctx);
/* page-locked host buffers for SNPGPU_HOST_PINNED feeds (the R shim allocates the reader's two
 * block buffers with this instead of VEC_AUTO_PTR, src/genIBS.cpp:305) */
int snpgpu_host_alloc(size_t bytes, void **out);
int snpgpu_host_free(void *p);
/* block until the last asynchronous copy out of `host_buf` issued by this context is complete */
int snpgpu_host_wait(snpgpu_ctx *ctx, const void *host_buf);
/* number of SNPs fed so far, and of those the polymorphic ones (GCTA's nLocus,
 * src/genPCA.cpp:1206; maintained by GRM_GCTA contexts only -- the IBS / KING-robust counters need no per-SNP
 * statistics and their pre-pass does not compute any) */
int snpgpu_counts(snpgpu_ctx *ctx, int64_t *n_snp_total, int64_t *n_locus);
/* Optional HIP-event timing of the dominant pair kernel launches inside snpgpu_feed
 * (events are recorded on the context's stream around each launch).  `which`: 0 = bit-plane
 * pair kernel, 1 = MFMA SYRK kernel.  Returns the summed kernel time and launch count since
 * timing was (re-)enabled. */
int snpgpu_set_timing(snpgpu_ctx *ctx, int enable);
int snpgpu_get_timing(snpgpu_ctx *ctx, int which, double *ms_sum, int64_t *launches);
/* size (elements) of the packed slab this context's panel produces */
int64_t snpgpu_slab_size(const snpgpu_ctx *ctx);

/* Finalisers.  `packed` != 0: write the packed-triangle slab of the panel
 * (rows row_begin..row_end-1; the whole triangle for a full context) --
 * the `useMatrix=TRUE` outputs.  `packed` == 0: write the full symmetric
 * n x n matrix (full contexts only). */

/* gnrIBSNum finaliser, src/genIBS.cpp:521-543: three int32 matrices */
int snpgpu_ibs_num(snpgpu_ctx *ctx, int32_t *ibs0, int32_t *ibs1, int32_t *ibs2,
                   int packed, int mem);
/* gnrIBSAve finaliser, src/genIBS.cpp:463-490 */
int snpgpu_ibs_ave(snpgpu_ctx *ctx, double *out, int packed, int mem);
/* raw KING-robust counters {IBS0,nLoci,SumSq,N1_Aa,N2_Aa} (TS_KINGRobust,
 * src/genKING.cpp:274-281), packed slab, uint32 [npair][5] */
int snpgpu_king_robust_counts(snpgpu_ctx *ctx, uint32_t *out5, int mem);
/* gnrIBD_KING_Robust finaliser, src/genKING.cpp:614-667.
 * family: int32 [n_samp] (host), negative = NA; NULL = all NA */
int snpgpu_king_robust(snpgpu_ctx *ctx, const int32_t *family, double *ibs0, double *kinship,
                       int packed, int mem);
/* gnrIBD_KING_Homo finaliser, src/genKING.cpp:516-560 */
int snpgpu_king_homo(snpgpu_ctx *ctx, double *k0, double *k1, int packed, int mem);
/* GCTA GRM: numerator / (2 (nLocus - Denom)), src/genPCA.cpp:1232-1236 + grm_output :1586-1602 */
int snpgpu_grm_gcta(snpgpu_ctx *ctx, double *out, int packed, int mem);
/* PCA covariance.  normalize != 0 applies C *= (n-1)/trace (src/genPCA.cpp:1386-1390;
 * needs the whole diagonal, i.e. a full context, unless `trace_in` > 0 is supplied);
 * trace_xtx receives the trace of this panel's diagonal BEFORE scaling (may be NULL). */
int snpgpu_pca_cov(snpgpu_ctx *ctx, double *out, int packed, int normalize, double trace_in,
                   double *trace_xtx, int mem);
/* PLINK method of moments on an IBS context: per-pair k0/k1 (Est_PLINK_Kinship,
 * src/genIBD.cpp:341-390; loops of gnrIBD_PLINK, src/genIBS.cpp:590-628).
 * e[5] = {E00, E01, E02, E11, E12} of EPrIBS_IBD (Init_EPrIBD_IBS, src/genIBD.cpp:253-338). */
int snpgpu_ibd_mom(snpgpu_ctx *ctx, const double *e, int kinship_constraint, double *k0, double *k1,
                   int packed, int mem);
/* EIGMIX coancestry: numerator / (SumDenominator - Denom), optional diagonal adjustment, times
 * `scale` (2 for snpgdsGRM(method="EIGMIX"), src/genEIGMIX.cpp:146-155, :645-653) */
int snpgpu_eigmix(snpgpu_ctx *ctx, int diagadj, double scale, double *out, int packed, int mem);
/* individual beta.  mode 0/1: gnrIBD_Beta with inbreeding = FALSE/TRUE (src/genBeta.cpp:384-452),
 * mode 2: CalcIndivBetaGRM (min-based transform, src/genBeta.cpp:263-357).  avg_val receives
 * grm_avg_value.  Full contexts only (the transform needs all pairs). */
int snpgpu_indiv_beta(snpgpu_ctx *ctx, int mode, double *out, double *avg_val, int packed, int mem);
/* top-k eigenpairs of the (normalised, full-context) PCA covariance:
 * replaces CalcEigen / LAPACK dspevx (src/genPCA.cpp:1262-1346).
 * eigval: double [k] descending, eigvec: double [n_samp][k] column-major (n x k).
 * n <= SNPGPU_EIG_DENSE_MAX (default 2048): hipSOLVER's dense syevdx on the finalised matrix, index range 1..k as the
 * reference asks LAPACK; larger n: the block-Krylov solver below on the resident panel (no n x n copy). */
int snpgpu_pca_eigen(snpgpu_ctx *ctx, int k, double *eigval, double *eigvec, int mem);

/* Building block of the distributed top-k eigen solver (snprelate_amd/eigen.py) that replaces
 * LAPACK dspevx at sizes where a dense solve is impossible: with the symmetric covariance held as
 * row panels on several devices,   Y += scale * C Q   is the sum over panels of
 *     Y[I]     += scale * P[I, r0:N]   Q[r0:N]
 *     Y[r1:N]  += scale * P[I, r1:N]^T Q[I]          (I = [r0,r1) = the panel's rows)
 * This call adds ONE panel's contribution (one pass over the fp64 panel accumulator on fp64 MFMAs, every
 * tile used for both triangles, after mirroring the panel's diagonal block; SNPGPU_EIG_BLAS=1: two rocBLAS
 * dgemms).  Q, Y: device pointers, column-major n_samp x m
 * (leading dimension n_samp).  PCA_COV contexts only; no feeds may follow. */
int snpgpu_pca_panel_matmul(snpgpu_ctx *ctx, double scale, const double *Q, int m, double *Y);
/* The same product with the panel values and the vectors rounded to fp32 and fp32 matrix instructions: half the time of
 * the fp64 form (which is bound by the fp64 matrix rate and its atomics); sums of more than 1024 terms and the result stay
 * fp64.  Relative error of a product ~5e-7 rms: what the Krylov solver runs most of its products on (snpgpu_eig_opts). */
int snpgpu_pca_panel_matmul_f32(snpgpu_ctx *ctx, double scale, const double *Q, int m, double *Y);
/* trace of this panel's diagonal (raw sums, before any scaling) */
int snpgpu_pca_panel_trace(snpgpu_ctx *ctx, double *trace);

/* Turn the accumulators into the FINAL matrix in place (the panel rectangle of fp64 sums becomes the result itself), so that
 * the panel product / the eigen solver below can work on matrices that are more than raw sums: GRM_GCTA
 * (numerator / (2 (nLocus - Denom)), src/genPCA.cpp:1232-1236) and EIGMIX (src/genEIGMIX.cpp:146-155 with `diagadj`, times
 * `scale`).  For PCA_COV it only settles pending terms (the (n-1)/trace factor travels with the products).  No block may be
 * fed afterwards; the kind's own finaliser (snpgpu_grm_gcta / snpgpu_eigmix) keeps working and copies the stored matrix out. */
int snpgpu_finalize_inplace(snpgpu_ctx *ctx, int diagadj, double scale);
/* Sampled reads of the panel's fp64 result plane (host arrays; sample indices with row_begin <= rows[k] < row_end, cols[k] >=
 * rows[k]): the FINAL matrix entries after snpgpu_finalize_inplace (GRM_GCTA / EIGMIX), the settled raw sums of a PCA_COV
 * context otherwise.  For parity checks at sizes where no slab can be copied out whole (sampled tiles of the
 * 500 000-sample job recomputed in fp64 on the host); no reference counterpart. */
int snpgpu_panel_entries(snpgpu_ctx *ctx, const int64_t *rows, const int64_t *cols, int64_t n_entries, double *out);

/* Top-k eigenpairs of the symmetric matrix held as row panels: replaces CalcEigen / LAPACK dspevx for ANY n
 * (src/genPCA.cpp:1262-1346; the same call behind gnrEigMix, src/genEIGMIX.cpp:700-702).  Thick-restarted block Krylov +
 * Rayleigh-Ritz in C++ / HIP (csrc/eigen.hip): the O(n^2) product runs on the panels, the tall-skinny algebra on their
 * device.  `panels`: contexts on ONE device -- PCA_COV, or GRM_GCTA / EIGMIX after snpgpu_finalize_inplace -- that tile
 * [0, n) of the triangle, or (one process per GPU) this rank's share of it: then opts->reduce must sum opts->y_buf
 * (double [block][n], device memory owned by the caller) over the ranks in place, e.g. one RCCL all-reduce.
 * The matrix is `scale` times the panels' contents (PCA: (n-1) / trace, src/genPCA.cpp:1386-1390).
 * eigval: HOST double [k] descending; eigvec: double [n][k] column-major (n x k) in `mem` (host, or the panels' device). */
typedef int (*snpgpu_reduce_fn)(void *user);
typedef struct snpgpu_eig_opts {
    double   tol;            /* largest relative residual |C v - theta v| / |theta| accepted (0 = 1e-9)            */
    int32_t  block;          /* vectors per Krylov block (0 = k + 8 rounded up to a multiple of 16)               */
    int32_t  depth;          /* blocks per restart cycle (0 = 24, fewer while the device lacks the memory)        */
    int32_t  max_restarts;   /* 0 = 60                                                                             */
    uint32_t seed;           /* start block (0 = 20240601); identical on every rank                               */
    double  *y_buf;          /* with `reduce`: the buffer every product is formed in before it is reduced         */
    snpgpu_reduce_fn reduce; /* NULL: the panels are the whole matrix                                              */
    void    *user;
    double   fp32_until;     /* mixed precision.  Restart cycles run entirely on fp32 products until the residual falls
                                below this (or stops falling); from then on only the product of a cycle's first block --
                                the vectors the previous cycle returned -- is fp64, which also yields their true
                                residual: that fp64 figure is what accepts the result (and is `max_rel_residual`).
                                0 = 1e-5, < 0 = fp64 products only (also: SNPGPU_EIG_FP32=0)                         */
} snpgpu_eig_opts;
typedef struct snpgpu_eig_info {
    int32_t restarts, matmuls, block, depth;
    double  max_rel_residual;
    int32_t matmuls_fp32;    /* how many of `matmuls` were fp32 products */
    int32_t reserved;
} snpgpu_eig_info;
int snpgpu_panels_topk_eigen(snpgpu_ctx *const *panels, int n_panels, double scale, int k, const snpgpu_eig_opts *opts,
                             double *eigval, double *eigvec, int mem, snpgpu_eig_info *info);

/* ---- (1c) several GPUs driven by ONE host process (an R session) ----------------------------------------------------
 * north_star: "the N x N output triangle is row-block partitioned across the 8 GPUs of one node with a final gather over
 * xGMI".  The object cuts the packed triangle into equal-area row panels (the device-level analogue of Array_SplitJobs,
 * src/dGenGWAS.cpp:2202-2216), `panels_per_device` per device (several even out the memory: the last equal-area panel is
 * a square holding a triangle), and creates one accumulator context per panel.  snpgpu_multi_feed moves a block across
 * PCIe ONCE, to devices[0], and forwards it to the other devices over xGMI (hipMemcpyPeerAsync, double-buffered, under the
 * kernels of the previous block); there is no collective on the data path.  The gathers below write the packed triangle
 * (CdMatTri order) -- every panel's slab is a contiguous range of it -- into host memory, or into device memory of
 * devices[0] through peer copies.  n_passes > 1: only the panels of pass `pass` are resident (output-stationary: KING-robust
 * keeps 20 B per pair, 2.5 TB at N = 500 000); the caller walks the SNP stream once per pass and every pass's gather fills
 * its own ranges of the same output.  A device may be listed more than once (tests: several "devices" on one GPU). */
typedef struct snpgpu_multi snpgpu_multi;
typedef struct snpgpu_multi_opts {
    const int32_t *devices;      /* HIP device ordinals                                  */
    int32_t n_devices;
    int32_t panels_per_device;   /* 0 = 1; -1 = the fewest that fit the devices' free memory (accumulators + per-panel scratch).
                                    With n_passes > 1 every pass must use the SAME value (the plan is cut into devices x panels x passes
                                    panels): resolve -1 once, in pass 0, and give the later passes what snpgpu_multi_get_status reports */
    int32_t n_passes;            /* 0 = 1                                                */
    int32_t pass;                /* 0 .. n_passes - 1                                    */
} snpgpu_multi_opts;
/* opts: bayesian and max_block_snps are used (device, rows and stream are set per panel) */
int snpgpu_multi_create(int kind, int64_t n_samp, const snpgpu_opts *opts, const snpgpu_multi_opts *mopts, snpgpu_multi **out);
int snpgpu_multi_destroy(snpgpu_multi *m);
/* number of resident panels; whether the eigen solver's broadcast / reduce go through RCCL (distinct devices and librccl
 * loadable; SNPGPU_MULTI_COMM=peer|rccl overrides) or through peer copies */
int snpgpu_multi_info(const snpgpu_multi *m, int *n_panels, int *uses_rccl);
/* one broadcast + one sum-reduction of a known pattern over the object's devices through the exchange path the eigen solver
 * uses (RCCL communicator, or peer copies when none could be built -- which snpgpu_multi_create reports on stderr and
 * SNPGPU_MULTI_COMM=rccl turns into an error): non-zero, with a message, if any device returns the wrong sum */
int snpgpu_multi_comm_selftest(snpgpu_multi *m, int *uses_rccl);
/* After a successful snpgpu_multi_comm_selftest the two data paths have been exercised as well (round 6): a known 2-bit block forwarded
 * from the first device to every other one the way snpgpu_multi_feed does it, verified on each receiving device, and a known slab
 * from every device written into its range of one buffer on the first device the way the gathers do it (one host thread per device,
 * asynchronous peer copies), verified there.  What the object found out about its devices: */
typedef struct snpgpu_multi_status {
    int32_t n_devices, n_distinct_devices, n_panels;
    int32_t panels_per_device;   /* of the plan: the resolved value when snpgpu_multi_opts.panels_per_device was -1                 */
    int32_t uses_rccl;           /* the eigen solver's broadcast / reduce go through an RCCL communicator                            */
    int32_t peer_pairs;          /* ordered pairs (a, b) of distinct devices of the list ...                                        */
    int32_t peer_pairs_enabled;  /* ... of which hipDeviceCanAccessPeer said yes and hipDeviceEnablePeerAccess succeeded (the others:
                                    a line on stderr at create; their copies are staged through host memory by the runtime)         */
    int32_t selftest_comm, selftest_feed, selftest_gather;   /* snpgpu_multi_comm_selftest: -1 not run, 0 failed, 1 passed           */
    int32_t reserved[6];
} snpgpu_multi_status;
int snpgpu_multi_get_status(const snpgpu_multi *m, snpgpu_multi_status *out);
/* panel i: its context (any level-1 call may be made on it), rows and device */
int snpgpu_multi_panel(const snpgpu_multi *m, int i, snpgpu_ctx **ctx, int64_t *row_begin, int64_t *row_end, int *device);
/* as snpgpu_feed; SNPGPU_DEVICE = memory of devices[0], which must stay untouched until snpgpu_multi_sync */
int snpgpu_multi_feed(snpgpu_multi *m, const void *geno, int64_t n_snp, int format, int mem);
int snpgpu_multi_host_wait(snpgpu_multi *m, const void *host_buf);
int snpgpu_multi_sync(snpgpu_multi *m);
int snpgpu_multi_counts(snpgpu_multi *m, int64_t *n_snp_total, int64_t *n_locus);
/* gathers of the packed triangle; `mem`: SNPGPU_HOST, or SNPGPU_DEVICE = memory of devices[0] */
int snpgpu_multi_ibs_num(snpgpu_multi *m, int32_t *ibs0, int32_t *ibs1, int32_t *ibs2, int mem);
int snpgpu_multi_ibs_ave(snpgpu_multi *m, double *out, int mem);
int snpgpu_multi_king_robust(snpgpu_multi *m, const int32_t *family, double *ibs0, double *kinship, int mem);
int snpgpu_multi_king_robust_counts(snpgpu_multi *m, uint32_t *out5, int mem);
int snpgpu_multi_king_homo(snpgpu_multi *m, double *k0, double *k1, int mem);
int snpgpu_multi_grm_gcta(snpgpu_multi *m, double *out, int mem);
int snpgpu_multi_eigmix(snpgpu_multi *m, int diagadj, double scale, double *out, int mem);
int snpgpu_multi_pca_trace(snpgpu_multi *m, double *trace);
/* out may be NULL (trace only); normalize != 0: C *= (n-1)/trace with the trace of ALL panels */
int snpgpu_multi_pca_cov(snpgpu_multi *m, double *out, int normalize, double *trace_xtx, int mem);
/* snpgpu_finalize_inplace on every panel */
int snpgpu_multi_finalize_inplace(snpgpu_multi *m, int diagadj, double scale);
/* top-k eigenpairs over all devices (a one-pass plan; GRM_GCTA / EIGMIX after snpgpu_multi_finalize_inplace): the
 * tall-skinny algebra runs on devices[0], every product Y = C Q on all devices -- the vector block is broadcast, the
 * partial products are reduced (RCCL ncclBroadcast / ncclReduce, or peer copies).  PCA_COV with scale <= 0: the
 * (n-1)/trace factor of gnrPCA.  eigval: host; eigvec: n x k column-major in `mem` (host / devices[0]); opts->reduce must
 * be NULL. */
int snpgpu_multi_topk_eigen(snpgpu_multi *m, double scale, int k, const snpgpu_eig_opts *opts, double *eigval, double *eigvec,
                            int mem, snpgpu_eig_info *info);

/* ---- (1b) PCA projections: SNP correlations, SNP loadings, sample loadings ---
 * A projector holds the sample-side matrix and per-block scratch; the caller keeps its block reader
 * (CGenoReadBySNP) and hands over one block at a time, as for the accumulators.  All arithmetic is fp64.
 * Matrices use R's layouts: eigvec = n_samp x n_eig column-major ([n_eig][n_samp]); per-block outputs
 * = n_eig x n_snp column-major ([n_snp][n_eig]); sample loadings = n_samp x n_eig column-major. */
typedef struct snpgpu_proj snpgpu_proj;
int snpgpu_proj_create(int64_t n_samp, int n_eig, const snpgpu_opts *opts, snpgpu_proj **out);
int snpgpu_proj_destroy(snpgpu_proj *p);
int snpgpu_proj_sync(snpgpu_proj *p);
/* eigenvectors of the samples (for gnrPCASNPLoading already multiplied by sqrt((n-1)/TraceXTX/eigenval),
 * src/genPCA.cpp:1499-1507) */
int snpgpu_proj_set_eigvec(snpgpu_proj *p, const double *eigvec, int mem);
/* body of CPCA_SNPCorr::Run (src/genPCA.cpp:860-899): Pearson correlation of each SNP of the block with
 * each eigenvector over the called genotypes; NaN for < 2 calls or zero variance */
int snpgpu_proj_snp_corr(snpgpu_proj *p, const void *geno, int64_t n_snp, int format, int mem,
                         double *out, int out_mem);
/* body of CPCA_SNPLoad::Run (src/genPCA.cpp:1000-1035): loading [n_snp][n_eig], afreq [n_snp] (mean
 * genotype), scale [n_snp] */
int snpgpu_proj_snp_loading(snpgpu_proj *p, const void *geno, int64_t n_snp, int format, int mem, int bayesian,
                            double *loading, double *afreq, double *scale, int out_mem);
/* the same product with the caller's centring and scaling, loading = sum_i (g_i - avg) * scale * eigvec_i over the
 * called genotypes: body of CEigMix_SNPLoad::Run (src/genEIGMIX.cpp:440-512) with avg = 2 * afreq and
 * scale = 1 / sqrt(sum 4 p (1 - p)) */
int snpgpu_proj_snp_loading_ext(snpgpu_proj *p, const void *geno, int64_t n_snp, int format, int mem, const double *avg,
                                const double *scale, int in_mem, double *loading, int out_mem);
/* body of CPCA_SampleLoad::Run (src/genPCA.cpp:1070-1110): accumulate one block; sload [n_snp][n_eig]
 * (SNP loadings times sqrt(ss/eigenval), R/PCA.R:283-285), afreq / scale as returned above */
int snpgpu_proj_samp_loading_feed(snpgpu_proj *p, const void *geno, int64_t n_snp, int format, int mem,
                                  const double *sload, const double *afreq, const double *scale, int in_mem);
int snpgpu_proj_samp_loading(snpgpu_proj *p, double *out, int out_mem);
int snpgpu_proj_samp_loading_reset(snpgpu_proj *p);
/* In the three block calls geno == NULL reuses the block staged by the previous call on this projector
 * (same n_snp): the randomised PCA multiplies every block by Y and by Y^T in one pass. */

/* ---- (2) workspace level: mirrors of the registered .Call routines ------ */
/* gnrSetGenoSpace(Node, SelSamp, SelSNP), src/SNPRelate.cpp:76-114: install an
 * in-memory genotype matrix (host, copied) as the process-global working space */
int snpgpu_ws_set_geno(const void *geno, int64_t n_snp, int64_t n_samp, int format, int device);
/* gnrSelSNP_Base(remove_mono, maf, missrate), src/SNPRelate.cpp:184-210 ->
 * CdBaseWorkSpace::Select_SNP_Base, src/dGenGWAS.cpp:361-397.
 * sel_out: uint8 [n_snp of the current selection] (may be NULL) */
int snpgpu_ws_sel_snp_base(int remove_mono, double maf, double missrate,
                           int32_t *n_excluded, uint8_t *sel_out);
/* gnrSelSNP_Base_Ex(afreq, remove_mono, maf, missrate), src/SNPRelate.cpp:215-239 ->
 * CdBaseWorkSpace::Select_SNP_Base_Ex, src/dGenGWAS.cpp:399-469: as above, but the monomorphic / MAF tests use the
 * caller's allele frequencies afreq [n_snp of the current selection] (non-finite = excluded) */
int snpgpu_ws_sel_snp_base_ex(const double *afreq, int remove_mono, double maf, double missrate,
                              int32_t *n_excluded, uint8_t *sel_out);
/* gnrGetGenoDim(), src/SNPRelate.cpp:158-181: {n_snp, n_samp} after selection */
int snpgpu_ws_get_geno_dim(int64_t *n_snp, int64_t *n_samp);
/* per-SNP allele frequency / missing rate over the working space
 * (Get_AF_MR_perSNP, src/dGenGWAS.cpp:472-552); any pointer may be NULL */
int snpgpu_ws_snp_rate_freq(double *af, double *maf, double *missrate);
int snpgpu_ws_clear(void);

/* gnrIBSNum(NumThread, Verbose), src/genIBS.cpp:500-550 */
int snpgpu_gnrIBSNum(int num_thread, int verbose, int32_t *ibs0, int32_t *ibs1, int32_t *ibs2);
/* gnrIBSAve(NumThread, useMatrix, Verbose), src/genIBS.cpp:441-497 */
int snpgpu_gnrIBSAve(int num_thread, int use_matrix, int verbose, double *out);
/* gnrIBD_KING_Robust(FamilyID, NumThread, useMatrix, Verbose), src/genKING.cpp:576-679 */
int snpgpu_gnrIBD_KING_Robust(const int32_t *family, int num_thread, int use_matrix, int verbose,
                              double *ibs0, double *kinship);
/* gnrIBD_KING_Homo(NumThread, useMatrix, Verbose), src/genKING.cpp:493-570 */
int snpgpu_gnrIBD_KING_Homo(int num_thread, int use_matrix, int verbose, double *k0, double *k1);
/* gnrGRM(NumThread, Method, GDS, useMatrix, Verbose), src/genPCA.cpp:1614-1717;
 * methods on this path: "GCTA", "Eigenstrat", "Corr", "EIGMIX", "IndivBeta" */
int snpgpu_gnrGRM(int num_thread, const char *method, int use_matrix, int verbose, double *out);
/* gnrGRMMerge(OutGDS, GDSList, Cmd, Weight, Verbose), src/genPCA.cpp:1721-1853 (caller R/IBD.R:624-741):
 * weighted combination of n_grm GRMs of the same N samples.  grm[k]: host, N x N doubles (symmetric; the rows the
 * kept GDS reader delivers).  cmd = the second element of the files' "command" node; ":method = IndivBeta" selects
 * the beta merge (back-transform with avg_val[k], re-baseline to the new minimum, :1744-1833), anything else the plain
 * weighted sum (:1835-1851).  out: host, N x N.  After a beta merge snpgpu_gnrGRM_avg_val returns the merged
 * average, as the reference's gnrGRM_avg_val does. */
int snpgpu_gnrGRMMerge(int n_grm, int64_t N, const double *const *grm, const char *cmd, const double *avg_val,
                       const double *weight, double *out, int device);
/* gnrIBD_PLINK(NumThread, AlleleFreq, UseSpecificAFreq, KinshipConstrict, useMatrix, Verbose),
 * src/genIBS.cpp:558-639.  allele_freq may be NULL (then the allele-count correction is used);
 * afreq_out: double [n_snp] */
int snpgpu_gnrIBD_PLINK(int num_thread, const double *allele_freq, int kinship_constraint, int use_matrix,
                        int verbose, double *k0, double *k1, double *afreq_out);
/* gnrIBD_Beta(Inbreeding, NumThread, useMatrix, Verbose), src/genBeta.cpp:361-460 */
int snpgpu_gnrIBD_Beta(int inbreeding, int num_thread, int use_matrix, int verbose, double *out, double *avg_val);
/* gnrGRM_avg_val(), src/genPCA.cpp:1605-1611 */
int snpgpu_gnrGRM_avg_val(double *avg_val);
/* gnrEigMix(EigenCnt, NumThread, ParamList{diagadj, ibdmat}, Verbose), src/genEIGMIX.cpp:656-740.
 * ibd (n x n), eigval [n] (NaN beyond eigen_cnt), eigvec (n x eigen_cnt), afreq [n_snp]: any may be NULL */
int snpgpu_gnrEigMix(int eigen_cnt, int num_thread, int diagadj, int verbose, double *ibd, double *eigval,
                     double *eigvec, double *afreq);
/* gnrPCA(EigenCnt, "exact", NumThread, ParamList, Verbose), src/genPCA.cpp:1355-1452.
 * genmat (n x n) may be NULL; eigval: double [n] (entries >= eigen_cnt are NaN as in
 * CalcEigen :1343-1345), eigvec: n x eigen_cnt; both may be NULL (genmat.only). */
int snpgpu_gnrPCA(int eigen_cnt, int num_thread, int bayesian, int verbose, double *trace_xtx,
                  double *genmat, double *eigval, double *eigvec, double *trace_val);

/* gnrPCA(EigenCnt, "randomized", NumThread, ParamList{aux.dim, iter.num, aux.mat}, Verbose): CRandomPCA::Run,
 * src/genPCA.cpp:472-803.  aux_mat = aux_dim x n_samp as R's rnorm(aux.dim * n.samp) is read ([aux_dim][n_samp]).
 * Returns what R/PCA.R:80-89 uses of the routine's list: sigma [n_samp] (zero beyond min(hsize, n_samp),
 * hsize = aux_dim * (iter_num + 1)), the first eigen_cnt rows of V^T as eigvec (n_samp x eigen_cnt
 * column-major) and trace2 = 2 * TraceXTX. */
int snpgpu_gnrPCA_randomized(int eigen_cnt, int aux_dim, int iter_num, const double *aux_mat, int num_thread,
                             int verbose, double *sigma, double *eigvec, double *trace2);
/* gnrPCACorr(LenEig, EigenVect, NumThread, GDSNode=NULL, Verbose), src/genPCA.cpp:1455-1484:
 * out = LenEig x n_snp column-major */
int snpgpu_gnrPCACorr(int len_eig, const double *eigvec, int num_thread, int verbose, double *out);
/* gnrPCASNPLoading(EigenVal, EigenVect, TraceXTX, NumThread, Bayesian, Verbose), src/genPCA.cpp:1488-1531:
 * eigvec = n_samp x len_eig; loading = len_eig x n_snp, afreq / scale = [n_snp] */
int snpgpu_gnrPCASNPLoading(const double *eigval, const double *eigvec, int len_eig, double trace_xtx,
                            int num_thread, int bayesian, int verbose, double *loading, double *afreq, double *scale);
/* gnrPCASampLoading(EigenCnt, SNPLoadings, AvgFreq, Scale, NumThread, Verbose), src/genPCA.cpp:1535-1562:
 * snp_loadings = eigen_cnt x n_snp; out = n_samp x eigen_cnt */
int snpgpu_gnrPCASampLoading(int eigen_cnt, const double *snp_loadings, const double *avg_freq, const double *scale,
                             int num_thread, int verbose, double *out);

/* gnrEigMixSNPLoading(EigenVal, EigenVect, AFreq, NumThread, Verbose), src/genEIGMIX.cpp:739-775:
 * loading = len_eig x n_snp */
int snpgpu_gnrEigMixSNPLoading(const double *eigval, const double *eigvec, int len_eig, const double *afreq,
                               int num_thread, int verbose, double *loading);
/* gnrEigMixSampLoading(SNPLoadings, AFreq, NumThread, Verbose), src/genEIGMIX.cpp:777-803: out = n_samp x eigen_cnt */
int snpgpu_gnrEigMixSampLoading(int eigen_cnt, const double *snp_loadings, const double *afreq, int num_thread,
                                int verbose, double *out);

/* ---- diagnostics (no reference counterpart) ---------------------------------------------------------------------------
 * What THIS device's matrix pipe sustains right now: a register-only stream of one MFMA instruction (never waiting on memory,
 * two waves per SIMD) run for `seconds`, rate taken over the second half.  The kernels of this library run against the socket
 * power cap, which depends on the operands' bit patterns and differs by a few per cent from box to box: bench.py measures it in
 * the run it reports (roofline.sustained_peak_measured).  tflops: TFLOP/s of the instruction; implied_mhz (may be NULL): the shader
 * clock that rate corresponds to (rate / flop per clock of the whole device). */
enum snpgpu_diag_mode {
    SNPGPU_DIAG_F16_ZERO = 0,      /* v_mfma_f32_32x32x16_f16, all operands zero: the unthrottled rate                              */
    SNPGPU_DIAG_F16_EXACT_ROW = 1, /* row operand small integers, column operand real-valued: the exact-row SYRK's operand classes */
    SNPGPU_DIAG_F16_UV = 2,        /* both operands (g - c) x fp16 factor: the single-product SYRK's operand classes                */
    SNPGPU_DIAG_FP4 = 3,           /* v_mfma_scale_f32_32x32x64_f8f6f4 on e2m1 operands {0, 1/2, 1} x {0, +-1}: the pair counters'  */
    SNPGPU_DIAG_F16_UV_16X16X32 = 4, /* the operands of mode 2 through v_mfma_f32_16x16x32_f16 (same flops, a quarter of the
                                      accumulator registers per instruction): what the other instruction shape sustains        */
    SNPGPU_DIAG_FP4_16X16X128 = 5, /* the operands of mode 3 through v_mfma_scale_f32_16x16x128_f8f6f4                            */
    SNPGPU_DIAG_F16_EXACT_ROW_16X16X32 = 6 /* the operands of mode 1 through v_mfma_f32_16x16x32_f16                            */
};
int snpgpu_diag_mfma_rate(int device, int mode, double seconds, double *tflops, double *implied_mhz);
/* PCI address "dddd:bb:dd.f" of HIP device `device` (hipDeviceGetPCIBusId): which physical GPU a rank really drives */
int snpgpu_diag_device_pci(int device, char *buf, int len);

#ifdef __cplusplus
}
#endif
#endif /* SNPGPU_H */
