/*
 * snp_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path)
 *
 * A plain-C restatement of the algorithms behind the reference's pairwise
 * hot path (SNPRelate v1.46.0).  It exists only so that tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() can check / time the HIP
 * path against an independent CPU answer.  Nothing under snprelate_amd/ may
 * import, link or call this file.
 *
 * Parity status ("pinning"), see tests/test_oracle_golden.py:
 *   - IBS counts       -> pinned by the reference's Validate.IBS.RData
 *   - KING-robust/homo -> pinned by Validate.KING.RData (both estimators)
 *   - PCA covariance   -> pinned by Validate.PCA.RData$genmat
 *   - GCTA GRM         -> no golden in the reference's tests; pinned by the
 *                         known answers recorded in SURVEY.md 8(c) (produced by
 *                         the reference's own classes during the survey) and by
 *                         the reference's self-consistency test (test_GRM.R).
 *
 * Every function cites the reference file:line it restates (paths relative
 * to the reference tree).  No reference source text is copied: the code below
 * was written from the algorithm description in SURVEY.md section 8(a).
 *
 * Genotype input convention (same as CGenoReadBySNP::Read output,
 * src/dGenGWAS.cpp:1218-1397): uint8 geno[L][N], SNP-major, sample fastest;
 * 0/1/2 = number of A alleles, any value > 2 = missing.
 *
 * Packed upper-triangle convention (CdMatTri, src/dGenGWAS.h:511-583):
 * row-major with diagonal, idx(i,j) = j + i*(2N-i-1)/2 for i <= j.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef int64_t i64;

static inline i64 tri_index(i64 n, i64 i, i64 j) { return j + i * (2 * n - i - 1) / 2; }

/* ------------------------------------------------------------------ */
/* per-SNP sum / count of non-missing genotypes                        */
/* restates vec_u8_geno_count (src/dVect.cpp:30-117)                   */
void orc_snp_stats(const uint8_t *g, i64 L, i64 N, int32_t *sum, int32_t *num)
{
#pragma omp parallel for schedule(static)
    for (i64 l = 0; l < L; l++) {
        const uint8_t *p = g + l * N;
        int32_t s = 0, c = 0;
        for (i64 i = 0; i < N; i++)
            if (p[i] <= 2) { s += p[i]; c++; }
        sum[l] = s; num[l] = c;
    }
}

/* ------------------------------------------------------------------ */
/* SNP filter of .InitFile2 -> gnrSelSNP_Base                          */
/* restates CdBaseWorkSpace::Select_SNP_Base (src/dGenGWAS.cpp:361-397)*/
/* and Get_AF_MR_perSNP (src/dGenGWAS.cpp:472-552).                    */
/* sel_out[l] = 1 keeps the SNP.  Returns the number of excluded SNPs. */
int orc_select_snp_base(const uint8_t *g, i64 L, i64 N, int remove_mono,
                        double maf, double missrate, uint8_t *sel_out)
{
    int32_t *sum = (int32_t *)malloc(sizeof(int32_t) * (size_t)L);
    int32_t *num = (int32_t *)malloc(sizeof(int32_t) * (size_t)L);
    orc_snp_stats(g, L, N, sum, num);
    int excluded = 0;
    for (i64 l = 0; l < L; l++) {
        int keep;
        if (num[l] > 0) {
            double F = (double)sum[l] / (2 * num[l]);
            double MAF = (F < 1 - F) ? F : (1 - F);
            double MR = 1 - ((double)num[l]) / (double)N;
            keep = 1;
            if (remove_mono && MAF <= 0) keep = 0;
            if (keep && MAF < maf) keep = 0;
            if (keep && MR > missrate) keep = 0;
        } else
            keep = 0; /* MAF is NaN */
        sel_out[l] = (uint8_t)keep;
        if (!keep) excluded++;
    }
    free(sum); free(num);
    return excluded;
}

/* ------------------------------------------------------------------ */
/* bit-plane packing: one sample, one block of SNPs                    */
/* restates PackSNPGeno1b (src/dGenGWAS.cpp:1429-1475):                */
/*   g -> (p1,p2): 0->(0,0) 1->(1,0) 2->(1,1) NA->(0,1); SNP k is bit   */
/*   k%64 of word k/64; tail / padding SNPs are NA.                     */
static void pack_block_1b(const uint8_t *gblock, i64 nsnp, i64 N, i64 nwords,
                          uint64_t *plane /* [N][2][nwords] */)
{
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < N; i++) {
        uint64_t *p1 = plane + (size_t)i * 2 * nwords, *p2 = p1 + nwords;
        for (i64 w = 0; w < nwords; w++) {
            uint64_t a = 0, b = 0;
            for (int k = 0; k < 64; k++) {
                i64 l = w * 64 + k;
                unsigned gg = (l < nsnp) ? gblock[l * N + i] : 3u;
                if (gg > 3) gg = 3;
                uint64_t b1 = (gg == 1 || gg == 2), b2 = (gg == 2 || gg == 3);
                a |= b1 << k; b |= b2 << k;
            }
            p1[w] = a; p2[w] = b;
        }
    }
}

#define POP64(x) ((uint32_t)__builtin_popcountll(x))

/* block size (SNPs) for the bit-plane kernels; the reference derives it  */
/* from the CPU cache (src/genIBS.cpp:286-289); any multiple of 64 gives  */
/* the same integer counts.                                               */
#define ORC_BITBLOCK 4096

/* ------------------------------------------------------------------ */
/* IBS0/IBS1/IBS2 counts per pair                                      */
/* restates CIBSCount::Run + thread_ibs_num (src/genIBS.cpp:154-328)   */
/* out: uint32 [N(N+1)/2][3] = {IBS0, IBS1, IBS2}, packed triangle     */
void orc_ibs_count(const uint8_t *g, i64 L, i64 N, uint32_t *out)
{
    const i64 nw = ORC_BITBLOCK / 64;
    uint64_t *plane = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)N * 2 * nw);
    memset(out, 0, sizeof(uint32_t) * 3 * (size_t)(N * (N + 1) / 2));
    for (i64 l0 = 0; l0 < L; l0 += ORC_BITBLOCK) {
        i64 nsnp = (L - l0 < ORC_BITBLOCK) ? (L - l0) : ORC_BITBLOCK;
        pack_block_1b(g + l0 * N, nsnp, N, nw, plane);
        const i64 nwb = (nsnp + 63) / 64;   /* words that hold SNPs (the rest is all-missing padding) */
#pragma omp parallel for schedule(dynamic, 4)
        for (i64 i = 0; i < N; i++) {
            const uint64_t *a1 = plane + (size_t)i * 2 * nw, *a2 = a1 + nw;
            uint32_t *po = out + 3 * tri_index(N, i, i);
            for (i64 j = i; j < N; j++, po += 3) {
                const uint64_t *b1 = plane + (size_t)j * 2 * nw, *b2 = b1 + nw;
                uint32_t c0 = 0, c2 = 0, cm = 0;
                for (i64 w = 0; w < nwb; w++) {
                    uint64_t mask = (a1[w] | ~a2[w]) & (b1[w] | ~b2[w]);
                    uint64_t ibs0 = ~((a1[w] ^ ~b1[w]) | (a2[w] ^ ~b2[w])) & mask;
                    uint64_t ibs2 = ~((a1[w] ^ b1[w]) | (a2[w] ^ b2[w])) & mask;
                    c0 += POP64(ibs0); c2 += POP64(ibs2); cm += POP64(mask);
                }
                po[0] += c0; po[1] += cm - c0 - c2; po[2] += c2;
            }
        }
    }
    free(plane);
}

/* finaliser of gnrIBSAve (src/genIBS.cpp:463-490):                    */
/* (0.5*IBS1 + IBS2) / (IBS0+IBS1+IBS2), packed triangle of doubles     */
void orc_ibs_ave(const uint32_t *cnt, i64 N, double *out_tri)
{
    i64 np = N * (N + 1) / 2;
    for (i64 k = 0; k < np; k++) {
        const uint32_t *p = cnt + 3 * k;
        out_tri[k] = (0.5 * p[1] + p[2]) / (double)(p[0] + p[1] + p[2]);
    }
}

/* ------------------------------------------------------------------ */
/* KING-robust counters                                                */
/* restates CKINGRobust::thread_ibs_num (src/genKING.cpp:292-426)      */
/* out: uint32 [npair][5] = {IBS0, nLoci, SumSq, N1_Aa, N2_Aa};         */
/* N1_Aa belongs to the ROW sample i (i <= j).                          */
void orc_king_robust_count(const uint8_t *g, i64 L, i64 N, uint32_t *out)
{
    const i64 nw = ORC_BITBLOCK / 64;
    uint64_t *plane = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)N * 2 * nw);
    memset(out, 0, sizeof(uint32_t) * 5 * (size_t)(N * (N + 1) / 2));
    for (i64 l0 = 0; l0 < L; l0 += ORC_BITBLOCK) {
        i64 nsnp = (L - l0 < ORC_BITBLOCK) ? (L - l0) : ORC_BITBLOCK;
        pack_block_1b(g + l0 * N, nsnp, N, nw, plane);
        const i64 nwb = (nsnp + 63) / 64;
#pragma omp parallel for schedule(dynamic, 4)
        for (i64 i = 0; i < N; i++) {
            const uint64_t *a1 = plane + (size_t)i * 2 * nw, *a2 = a1 + nw;
            uint32_t *po = out + 5 * tri_index(N, i, i);
            for (i64 j = i; j < N; j++, po += 5) {
                const uint64_t *b1 = plane + (size_t)j * 2 * nw, *b2 = b1 + nw;
                uint32_t c0 = 0, cn = 0, ch = 0, n1 = 0, n2 = 0;
                for (i64 w = 0; w < nwb; w++) {
                    uint64_t mask = (a1[w] | ~a2[w]) & (b1[w] | ~b2[w]);
                    uint64_t ibs0 = ~((a1[w] ^ ~b1[w]) | (a2[w] ^ ~b2[w])) & mask;
                    uint64_t het = ((a1[w] ^ a2[w]) ^ (b1[w] ^ b2[w])) & mask;
                    uint64_t Aa1 = a1[w] & ~a2[w] & mask;
                    uint64_t Aa2 = b1[w] & ~b2[w] & mask;
                    c0 += POP64(ibs0); cn += POP64(mask); ch += POP64(het);
                    n1 += POP64(Aa1); n2 += POP64(Aa2);
                }
                po[0] += c0; po[1] += cn; po[2] += ch + 4 * c0; po[3] += n1; po[4] += n2;
            }
        }
    }
    free(plane);
}

/* finaliser of gnrIBD_KING_Robust (src/genKING.cpp:614-667)           */
/* family[i] < 0 stands for NA_INTEGER.  Outputs packed triangles.      */
void orc_king_robust_final(const uint32_t *cnt, i64 N, const int32_t *family,
                           double *ibs0_tri, double *kin_tri)
{
    i64 k = 0;
    for (i64 i = 0; i < N; i++) {
        ibs0_tri[k] = 0; kin_tri[k] = 0.5; k++;
        for (i64 j = i + 1; j < N; j++, k++) {
            const uint32_t *p = cnt + 5 * k;
            ibs0_tri[k] = (p[1] > 0) ? ((double)p[0] / p[1]) : NAN;
            int f1 = family ? family[i] : -1, f2 = family ? family[j] : -1;
            double v;
            if (f1 == f2 && f1 >= 0)
                v = 0.5 - p[2] / (2.0 * (uint32_t)(p[3] + p[4]));
            else
                v = 0.5 - p[2] / (4.0 * (p[3] < p[4] ? p[3] : p[4]));
            if (!isfinite(v)) v = NAN;
            kin_tri[k] = v;
        }
    }
}

/* ------------------------------------------------------------------ */
/* KING-homo accumulators                                              */
/* restates CKINGHomo::Run / thread_ibs_num (src/genKING.cpp:58-266)   */
/* cnt: uint32 [npair][2] = {IBS0, SumSq}; fsum: double [npair][2] =    */
/* {SumAFreq, SumAFreq2}, p = 0.5*sum/num over the given samples.       */
void orc_king_homo_count(const uint8_t *g, i64 L, i64 N, uint32_t *cnt, double *fsum)
{
    const i64 nw = ORC_BITBLOCK / 64;
    i64 np = N * (N + 1) / 2;
    uint64_t *plane = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)N * 2 * nw);
    double *af = (double *)malloc(sizeof(double) * ORC_BITBLOCK * 2), *af2 = af + ORC_BITBLOCK;
    int32_t *sum = (int32_t *)malloc(sizeof(int32_t) * ORC_BITBLOCK * 2), *num = sum + ORC_BITBLOCK;
    memset(cnt, 0, sizeof(uint32_t) * 2 * (size_t)np);
    memset(fsum, 0, sizeof(double) * 2 * (size_t)np);
    for (i64 l0 = 0; l0 < L; l0 += ORC_BITBLOCK) {
        i64 nsnp = (L - l0 < ORC_BITBLOCK) ? (L - l0) : ORC_BITBLOCK;
        orc_snp_stats(g + l0 * N, nsnp, N, sum, num);
        for (i64 k = 0; k < ORC_BITBLOCK; k++) {
            double s = 0;
            if (k < nsnp) {
                double p = (num[k] > 0) ? 0.5 * sum[k] / num[k] : 0;
                s = p * (1 - p);
            }
            af[k] = s; af2[k] = s * s;
        }
        pack_block_1b(g + l0 * N, nsnp, N, nw, plane);
#pragma omp parallel for schedule(dynamic, 4)
        for (i64 i = 0; i < N; i++) {
            const uint64_t *a1 = plane + (size_t)i * 2 * nw, *a2 = a1 + nw;
            i64 k0 = tri_index(N, i, i);
            for (i64 j = i; j < N; j++) {
                const uint64_t *b1 = plane + (size_t)j * 2 * nw, *b2 = b1 + nw;
                uint32_t c0 = 0, ch = 0;
                double s1 = 0, s2 = 0;
                for (i64 w = 0; w < nw; w++) {
                    uint64_t mask = (a1[w] | ~a2[w]) & (b1[w] | ~b2[w]);
                    uint64_t ibs0 = ~((a1[w] ^ ~b1[w]) | (a2[w] ^ ~b2[w])) & mask;
                    uint64_t het = ((a1[w] ^ a2[w]) ^ (b1[w] ^ b2[w])) & mask;
                    c0 += POP64(ibs0); ch += POP64(het);
                    const double *pa = af + w * 64, *pb = af2 + w * 64;
                    while (mask) {
                        int b = __builtin_ctzll(mask);
                        s1 += pa[b]; s2 += pb[b];
                        mask &= mask - 1;
                    }
                }
                uint32_t *pc = cnt + 2 * (k0 + (j - i));
                double *pf = fsum + 2 * (k0 + (j - i));
                pc[0] += c0; pc[1] += ch + 4 * c0; pf[0] += s1; pf[1] += s2;
            }
        }
    }
    free(plane); free(af); free(sum);
}

/* finaliser of gnrIBD_KING_Homo (src/genKING.cpp:493-570) */
void orc_king_homo_final(const uint32_t *cnt, const double *fsum, i64 N,
                         double *k0_tri, double *k1_tri)
{
    i64 k = 0;
    for (i64 i = 0; i < N; i++) {
        k0_tri[k] = 0; k1_tri[k] = 0; k++;
        for (i64 j = i + 1; j < N; j++, k++) {
            double theta = 0.5 - cnt[2 * k + 1] / (8 * fsum[2 * k]);
            double k0 = cnt[2 * k] / (2 * fsum[2 * k + 1]);
            double k1 = 2 - 2 * k0 - 4 * theta;
            k0_tri[k] = isfinite(k0) ? k0 : NAN;
            k1_tri[k] = isfinite(k1) ? k1 : NAN;
        }
    }
}

/* ------------------------------------------------------------------ */
/* centred/scaled genotype block and pairwise dot products             */
/* restates CProdMat_Base::SummarizeGeno_SampxSNP / DivideGeno /       */
/* rsqrt_prod (src/genPCA.cpp:84-181), TransposeGenotype               */
/* (src/genPCA.h:93-108), GenoSub/GenoMul (src/genPCA.cpp:315-368) and */
/* CProdMat_AlgArith::MulAdd (src/genPCA.cpp:229-312).                 */
#define ORC_COVBLOCK 256

/* mode 0: scale = 1/sqrt(s(1-s)), s = avg/2, zero unless 0<s<1        */
/* mode 1: Bayesian scale (src/genPCA.cpp:441-453)                     */
static void build_z_block(const uint8_t *gb, i64 nsnp, i64 N, int mode,
                          int32_t *sum, int32_t *num, double *Z /* [N][ORC_COVBLOCK] */)
{
    double avg[ORC_COVBLOCK], scale[ORC_COVBLOCK];
    orc_snp_stats(gb, nsnp, N, sum, num);
    for (i64 k = 0; k < nsnp; k++) {
        avg[k] = (num[k] > 0) ? ((double)sum[k] / num[k]) : 0;
        if (mode == 0) {
            double s = avg[k] * 0.5;
            scale[k] = (0 < s && s < 1) ? (1.0 / sqrt(s * (1 - s))) : 0;
        } else {
            double s = (sum[k] + 1.0) / (2 * num[k] + 2);
            scale[k] = 1.0 / sqrt(s * (1 - s));
        }
    }
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < N; i++) {
        double *z = Z + (size_t)i * ORC_COVBLOCK;
        for (i64 k = 0; k < nsnp; k++) {
            uint8_t gg = gb[k * N + i];
            double v = (gg <= 2) ? (double)gg : avg[k];
            z[k] = (v - avg[k]) * scale[k];
        }
        for (i64 k = nsnp; k < ORC_COVBLOCK; k++) z[k] = 0;
    }
}

static void muladd_block(const double *Z, i64 N, double *cov_tri)
{
#pragma omp parallel for schedule(dynamic, 4)
    for (i64 i = 0; i < N; i++) {
        const double *zi = Z + (size_t)i * ORC_COVBLOCK;
        double *po = cov_tri + tri_index(N, i, i);
        for (i64 j = i; j < N; j++) {
            const double *zj = Z + (size_t)j * ORC_COVBLOCK;
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            for (i64 k = 0; k < ORC_COVBLOCK; k += 4) {
                s0 += zi[k] * zj[k]; s1 += zi[k + 1] * zj[k + 1];
                s2 += zi[k + 2] * zj[k + 2]; s3 += zi[k + 3] * zj[k + 3];
            }
            po[j - i] += (s0 + s1) + (s2 + s3);
        }
    }
}

/* raw covariance numerator of CExactPCA::Run (src/genPCA.cpp:395-464) */
void orc_pca_cov(const uint8_t *g, i64 L, i64 N, int bayesian, double *cov_tri)
{
    double *Z = (double *)malloc(sizeof(double) * (size_t)N * ORC_COVBLOCK);
    int32_t sum[ORC_COVBLOCK], num[ORC_COVBLOCK];
    memset(cov_tri, 0, sizeof(double) * (size_t)(N * (N + 1) / 2));
    for (i64 l0 = 0; l0 < L; l0 += ORC_COVBLOCK) {
        i64 nsnp = (L - l0 < ORC_COVBLOCK) ? (L - l0) : ORC_COVBLOCK;
        build_z_block(g + l0 * N, nsnp, N, bayesian ? 1 : 0, sum, num, Z);
        muladd_block(Z, N, cov_tri);
    }
    free(Z);
}

/* trace normalisation of gnrPCA (src/genPCA.cpp:1386-1390) and the     */
/* Eigenstrat GRM branch (src/genPCA.cpp:1633-1640): C *= (N-1)/trace   */
/* returns TraceXTX (the trace before scaling)                          */
double orc_trace_normalize(double *cov_tri, i64 N)
{
    double tr = 0;
    for (i64 i = 0; i < N; i++) tr += cov_tri[tri_index(N, i, i)];
    double scale = (double)(N - 1) / tr;
    i64 np = N * (N + 1) / 2;
    for (i64 k = 0; k < np; k++) cov_tri[k] *= scale;
    return tr;
}

/* GCTA GRM: restates CGCTA_AlgArith::Run (src/genPCA.cpp:1148-1237)    */
/* including the per-pair missing denominator loop (:1201-1224) and the */
/* unguarded final division (:1232-1236).                               */
void orc_grm_gcta(const uint8_t *g, i64 L, i64 N, double *cov_tri)
{
    i64 np = N * (N + 1) / 2;
    double *Z = (double *)malloc(sizeof(double) * (size_t)N * ORC_COVBLOCK);
    int32_t *denom = (int32_t *)calloc((size_t)np, sizeof(int32_t));
    int32_t sum[ORC_COVBLOCK], num[ORC_COVBLOCK];
    i64 nLocus = 0;
    memset(cov_tri, 0, sizeof(double) * (size_t)np);
    for (i64 l0 = 0; l0 < L; l0 += ORC_COVBLOCK) {
        i64 nsnp = (L - l0 < ORC_COVBLOCK) ? (L - l0) : ORC_COVBLOCK;
        const uint8_t *gb = g + l0 * N;
        build_z_block(gb, nsnp, N, 0, sum, num, Z);
        for (i64 k = 0; k < nsnp; k++) {
            if (0 < sum[k] && sum[k] < 2 * num[k]) {
                nLocus++;
                const uint8_t *gg = gb + k * N;
                for (i64 j = 0; j < N; j++) {
                    if (gg[j] > 2) {
                        int32_t *row = denom + tri_index(N, j, j);
                        for (i64 c = 0; c < N - j; c++) row[c]++;
                        for (i64 r = j - 1; r >= 0; r--)
                            if (gg[r] <= 2) denom[tri_index(N, r, j)]++;
                    }
                }
            }
        }
        muladd_block(Z, N, cov_tri);
    }
    for (i64 k = 0; k < np; k++)
        cov_tri[k] /= (double)(2 * (nLocus - (i64)denom[k]));
    free(Z); free(denom);
}

/* expand a packed upper triangle to a full symmetric N x N matrix      */
/* (CdMatTri::SaveTo, src/dGenGWAS.h:563-572)                           */
void orc_tri_to_full_f64(const double *tri, i64 N, double *full)
{
    i64 k = 0;
    for (i64 i = 0; i < N; i++)
        for (i64 j = i; j < N; j++, k++)
            full[i * N + j] = full[j * N + i] = tri[k];
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}
