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
 *   - PLINK MoM        -> pinned by Validate.MoM.RData (k0, k1, afreq)
 *   - Individual beta  -> pinned by Validate.Beta.RData
 *   - EIGMIX           -> pinned by Validate.EIGMIX.RData
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
#define ORC_TILE 128      /* samples per tile of the pair loops: 128 x 1 KB of bit planes per side */

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
        /* sample tiles of ORC_TILE x ORC_TILE pairs, one task per tile: the bit planes of a tile's samples stay in the  */
        /* core's cache while its pairs are counted (a row-by-row walk streams all N planes per row and is bound by     */
        /* memory bandwidth long before 128 cores are busy); every pair is still counted word by word as above          */
        const i64 nt = (N + ORC_TILE - 1) / ORC_TILE;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
        for (i64 ti = 0; ti < nt; ti++)
        for (i64 tj = 0; tj < nt; tj++) {
            if (tj < ti) continue;
            const i64 i1 = (ti + 1) * ORC_TILE < N ? (ti + 1) * ORC_TILE : N, j1 = (tj + 1) * ORC_TILE < N ? (tj + 1) * ORC_TILE : N;
            for (i64 i = ti * ORC_TILE; i < i1; i++) {
                const uint64_t *a1 = plane + (size_t)i * 2 * nw, *a2 = a1 + nw;
                const i64 j0 = (tj * ORC_TILE > i) ? tj * ORC_TILE : i;
                uint32_t *po = out + 3 * tri_index(N, i, j0);
                for (i64 j = j0; j < j1; j++, po += 3) {
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
        const i64 nt = (N + ORC_TILE - 1) / ORC_TILE;        /* sample tiles, as in orc_ibs_count */
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
        for (i64 ti = 0; ti < nt; ti++)
        for (i64 tj = 0; tj < nt; tj++) {
            if (tj < ti) continue;
            const i64 i1 = (ti + 1) * ORC_TILE < N ? (ti + 1) * ORC_TILE : N, j1 = (tj + 1) * ORC_TILE < N ? (tj + 1) * ORC_TILE : N;
            for (i64 i = ti * ORC_TILE; i < i1; i++) {
                const uint64_t *a1 = plane + (size_t)i * 2 * nw, *a2 = a1 + nw;
                const i64 j0 = (tj * ORC_TILE > i) ? tj * ORC_TILE : i;
                uint32_t *po = out + 5 * tri_index(N, i, j0);
                for (i64 j = j0; j < j1; j++, po += 5) {
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
    /* sample tiles, one task per tile (see orc_ibs_count): the rows of a tile are 64 x 2 KB; each pair's dot product is  */
    /* formed exactly as before, so the sums do not depend on the tiling                                                    */
    const i64 T = 64, nt = (N + T - 1) / T;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (i64 ti = 0; ti < nt; ti++)
    for (i64 tj = 0; tj < nt; tj++) {
        if (tj < ti) continue;
        const i64 i1 = (ti + 1) * T < N ? (ti + 1) * T : N, j1 = (tj + 1) * T < N ? (tj + 1) * T : N;
        for (i64 i = ti * T; i < i1; i++) {
            const double *zi = Z + (size_t)i * ORC_COVBLOCK;
            double *po = cov_tri + tri_index(N, i, i);
            for (i64 j = (tj * T > i) ? tj * T : i; j < j1; j++) {
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
    int32_t *miss_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    int32_t sum[ORC_COVBLOCK], num[ORC_COVBLOCK];
    i64 nLocus = 0;
    memset(cov_tri, 0, sizeof(double) * (size_t)np);
    for (i64 l0 = 0; l0 < L; l0 += ORC_COVBLOCK) {
        i64 nsnp = (L - l0 < ORC_COVBLOCK) ? (L - l0) : ORC_COVBLOCK;
        const uint8_t *gb = g + l0 * N;
        build_z_block(gb, nsnp, N, 0, sum, num, Z);
        /* The reference walks the missing calls of every polymorphic SNP on ONE thread (src/genPCA.cpp:1209-1219): pair (r, j)  */
        /* counts a SNP where r or j is missing.  Same counts here, shared out by ROW of the triangle so that the baseline  */
        /* scales with the cores: row r takes +1 in every column when r itself is missing, else +1 in the columns j > r of   */
        /* the SNP's missing samples.                                                                                       */
        for (i64 k = 0; k < nsnp; k++) {
            if (0 < sum[k] && sum[k] < 2 * num[k]) {
                nLocus++;
                const uint8_t *gg = gb + k * N;
                i64 nm = 0;
                for (i64 j = 0; j < N; j++) if (gg[j] > 2) miss_idx[nm++] = (int32_t)j;
                if (nm == 0) continue;
#pragma omp parallel for schedule(static) if (nm * N > 200000)
                for (i64 r = 0; r < N; r++) {
                    int32_t *row = denom + tri_index(N, r, r);
                    if (gg[r] > 2) { for (i64 c = 0; c < N - r; c++) row[c]++; }
                    else for (i64 t = 0; t < nm; t++) { const i64 j = miss_idx[t]; if (j > r) row[j - r]++; }
                }
            }
        }
        muladd_block(Z, N, cov_tri);
    }
    for (i64 k = 0; k < np; k++)
        cov_tri[k] /= (double)(2 * (nLocus - (i64)denom[k]));
    free(Z); free(denom); free(miss_idx);
}

/* ------------------------------------------------------------------ */
/* PLINK method of moments (snpgdsIBDMoM)                              */
/* expectations: restates IBD::Init_EPrIBD_IBS (src/genIBD.cpp:253-338) */
/* e[5] = {E00, E01, E02, E11, E12}; AA = #(g==2), AB = #(g==1), BB = #(g==0) */
/* (GetABNumPerSNP, src/dGenGWAS.cpp:314-360).  in_afreq may be NULL.   */
void orc_mom_expect(const uint8_t *g, i64 L, i64 N, const double *in_afreq, double *e, double *out_afreq)
{
    double s00 = 0, s01 = 0, s02 = 0, s11 = 0, s12 = 0;
    i64 nValid = 0;
    for (i64 l = 0; l < L; l++) {
        const uint8_t *p8 = g + l * N;
        long AA = 0, AB = 0, BB = 0;
        if (!in_afreq)
            for (i64 i = 0; i < N; i++) { if (p8[i] == 0) BB++; else if (p8[i] == 1) AB++; else if (p8[i] == 2) AA++; }
        long n = 2 * (AA + AB + BB);
        double p = (n > 0) ? ((double)(2 * AA + AB) / n) : NAN;
        if (in_afreq) {
            p = in_afreq[l];
            if (isfinite(p) && (p < 0 || p > 1)) p = NAN;
        }
        if (out_afreq) out_afreq[l] = p;
        double q = 1 - p, Na = n, x = 2 * AA + AB, y = 2 * BB + AB;
        double a00, a01, a02, a11, a12;
        if (!in_afreq) {   /* CorrectFactor */
            a00 = 2*p*p*q*q * ((x-1)/x * (y-1)/y * (Na/(Na-1)) * (Na/(Na-2)) * (Na/(Na-3)));
            a01 = 4*p*p*p*q * ((x-1)/x * (x-2)/x * (Na/(Na-1)) * (Na/(Na-2)) * (Na/(Na-3))) +
                  4*p*q*q*q * ((y-1)/y * (y-2)/y * (Na/(Na-1)) * (Na/(Na-2)) * (Na/(Na-3)));
            a02 = q*q*q*q * ((y-1)/y * (y-2)/y * (y-3)/y * (Na/(Na-1)) * (Na/(Na-2)) * (Na/(Na-3))) +
                  p*p*p*p * ((x-1)/x * (x-2)/x * (x-3)/x * (Na/(Na-1)) * (Na/(Na-2)) * (Na/(Na-3))) +
                  4*p*p*q*q * ((x-1)/x * (y-1)/y * (Na/(Na-1)) * (Na/(Na-2)) * (Na/(Na-3)));
            a11 = 2*p*p*q * ((x-1)/x * Na/(Na-1) * Na/(Na-2)) + 2*p*q*q * ((y-1)/y * Na/(Na-1) * Na/(Na-2));
            a12 = p*p*p * ((x-1)/x * (x-2)/x * Na/(Na-1) * Na/(Na-2)) + q*q*q * ((y-1)/y * (y-2)/y * Na/(Na-1) * Na/(Na-2)) +
                  p*p*q * ((x-1)/x * Na/(Na-1) * Na/(Na-2)) + p*q*q * ((y-1)/y * Na/(Na-1) * Na/(Na-2));
        } else {
            a00 = 2*p*p*q*q; a01 = 4*p*p*p*q + 4*p*q*q*q; a02 = q*q*q*q + p*p*p*p + 4*p*p*q*q;
            a11 = 2*p*p*q + 2*p*q*q; a12 = p*p*p + q*q*q + p*p*q + p*q*q;
        }
        if (isfinite(a00) && isfinite(a01) && isfinite(a02) && isfinite(a11) && isfinite(a12)) {
            s00 += a00; s01 += a01; s02 += a02; s11 += a11; s12 += a12; nValid++;
        }
    }
    e[0] = s00 / nValid; e[1] = s01 / nValid; e[2] = s02 / nValid; e[3] = s11 / nValid; e[4] = s12 / nValid;
}

/* per-pair estimate: restates IBD::Est_PLINK_Kinship (src/genIBD.cpp:341-390) and the  */
/* gnrIBD_PLINK loops (src/genIBS.cpp:590-628); cnt = {IBS0, IBS1, IBS2} per pair        */
void orc_mom_final(const uint32_t *cnt, i64 N, const double *e, int constraint, double *k0_tri, double *k1_tri)
{
    i64 k = 0;
    for (i64 i = 0; i < N; i++) {
        k0_tri[k] = 0; k1_tri[k] = 0; k++;
        for (i64 j = i + 1; j < N; j++, k++) {
            int IBS0 = (int)cnt[3*k], IBS1 = (int)cnt[3*k+1], IBS2 = (int)cnt[3*k+2];
            int n012 = IBS0 + IBS1 + IBS2;
            double e00 = e[0]*n012, e01 = e[1]*n012, e11 = e[3]*n012, e02 = e[2]*n012, e12 = e[4]*n012, e22 = 1.0*n012;
            double k0 = IBS0 / e00;
            double k1 = (IBS1 - k0 * e01) / e11;
            double k2 = (IBS2 - k0*e02 - k1*e12) / e22;
            if (k0 > 1) { k0 = 1; k1 = k2 = 0; }
            if (k1 > 1) { k1 = 1; k0 = k2 = 0; }
            if (k2 > 1) { k2 = 1; k0 = k1 = 0; }
            if (k0 < 0) { double S = k1+k2; k1 /= S; k2 /= S; k0 = 0; }
            if (k1 < 0) { double S = k0+k2; k0 /= S; k2 /= S; k1 = 0; }
            if (k2 < 0) { double S = k0+k1; k0 /= S; k1 /= S; k2 = 0; }
            if (constraint) {
                k2 = 1 - k0 - k1;
                double pihat = k1 / 2 + k2;
                if (pihat*pihat < k2) { k0 = (1-pihat)*(1-pihat); k1 = 2*pihat*(1-pihat); }
            }
            k0_tri[k] = k0; k1_tri[k] = k1;
        }
    }
}

/* ------------------------------------------------------------------ */
/* individual beta counters: restates CIndivBeta::thread_ibs_num       */
/* (src/genBeta.cpp:65-183): out[npair][2] = {ibscnt, num}             */
void orc_beta_count(const uint8_t *g, i64 L, i64 N, uint32_t *out)
{
    const i64 nw = ORC_BITBLOCK / 64;
    uint64_t *plane = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)N * 2 * nw);
    memset(out, 0, sizeof(uint32_t) * 2 * (size_t)(N * (N + 1) / 2));
    for (i64 l0 = 0; l0 < L; l0 += ORC_BITBLOCK) {
        i64 nsnp = (L - l0 < ORC_BITBLOCK) ? (L - l0) : ORC_BITBLOCK;
        pack_block_1b(g + l0 * N, nsnp, N, nw, plane);
        const i64 nwb = (nsnp + 63) / 64;
#pragma omp parallel for schedule(dynamic, 4)
        for (i64 i = 0; i < N; i++) {
            const uint64_t *a1 = plane + (size_t)i * 2 * nw, *a2 = a1 + nw;
            uint32_t *po = out + 2 * tri_index(N, i, i);
            for (i64 j = i; j < N; j++, po += 2) {
                const uint64_t *b1 = plane + (size_t)j * 2 * nw, *b2 = b1 + nw;
                uint32_t c = 0, m = 0;
                for (i64 w = 0; w < nwb; w++) {
                    uint64_t mask = (a1[w] | ~a2[w]) & (b1[w] | ~b2[w]);
                    uint64_t het = (a1[w] ^ a2[w]) | (b1[w] ^ b2[w]);
                    uint64_t ibs2 = ~(het | (a1[w] ^ b1[w]));
                    c += POP64(het & mask) + 2 * POP64(ibs2 & mask);
                    m += POP64(mask);
                }
                po[0] += c; po[1] += m;
            }
        }
    }
    free(plane);
}

/* gnrIBD_Beta finaliser (src/genBeta.cpp:384-452); returns avg (grm_avg_value) */
double orc_beta_final_ibd(const uint32_t *cnt, i64 N, int inbreeding, double *out_tri)
{
    i64 k = 0; double avg = 0;
    for (i64 i = 0; i < N; i++) {
        out_tri[k] = inbreeding ? ((double)cnt[2*k] / cnt[2*k+1] - 1) : ((0.5 * cnt[2*k]) / cnt[2*k+1]);
        k++;
        for (i64 j = i + 1; j < N; j++, k++) {
            double s = (0.5 * cnt[2*k]) / cnt[2*k+1];
            out_tri[k] = s; avg += s;
        }
    }
    avg /= (double)(N * (N - 1) / 2);
    double bt = 1.0 / (1 - avg);
    i64 np = N * (N + 1) / 2;
    for (k = 0; k < np; k++) out_tri[k] = (out_tri[k] - avg) * bt;
    return avg;
}

/* CalcIndivBetaGRM_Mat (src/genBeta.cpp:263-303): snpgdsGRM(method="IndivBeta") */
double orc_beta_final_grm(const uint32_t *cnt, i64 N, double *out_tri)
{
    i64 k = 0; double avg = 0;
    double mn = (double)cnt[0] / cnt[1] - 1, r;
    for (i64 i = 0; i < N; i++) {
        out_tri[k] = r = (double)cnt[2*k] / cnt[2*k+1] - 1; k++;
        if (mn > r) mn = r;
        for (i64 j = i + 1; j < N; j++, k++) {
            out_tri[k] = r = (0.5 * cnt[2*k]) / cnt[2*k+1];
            avg += r;
            if (mn > r) mn = r;
        }
    }
    avg /= (double)(N * (N - 1) / 2);
    double scale = 2.0 / (1 - mn);
    k = 0;
    for (i64 i = 0; i < N; i++) {
        out_tri[k] = (out_tri[k] - mn) * scale * 0.5 + 1; k++;
        for (i64 j = i + 1; j < N; j++, k++) out_tri[k] = (out_tri[k] - mn) * scale;
    }
    return avg;
}

/* ------------------------------------------------------------------ */
/* EIGMIX: restates CEigMix_AlgArith::Run (src/genEIGMIX.cpp:43-160)   */
void orc_eigmix(const uint8_t *g, i64 L, i64 N, int diagadj, double *ibd_tri, double *afreq_out)
{
    i64 np = N * (N + 1) / 2;
    double *Z = (double *)malloc(sizeof(double) * (size_t)N * ORC_COVBLOCK);
    double *denom = (double *)calloc((size_t)np, sizeof(double));
    int *diagv = (int *)calloc((size_t)N, sizeof(int));
    int32_t sum[ORC_COVBLOCK], num[ORC_COVBLOCK];
    double sumden = 0;
    memset(ibd_tri, 0, sizeof(double) * (size_t)np);
    for (i64 l0 = 0; l0 < L; l0 += ORC_COVBLOCK) {
        i64 nsnp = (L - l0 < ORC_COVBLOCK) ? (L - l0) : ORC_COVBLOCK;
        const uint8_t *gb = g + l0 * N;
        orc_snp_stats(gb, nsnp, N, sum, num);
        for (i64 k = 0; k < nsnp; k++) {
            double avg = (num[k] > 0) ? ((double)sum[k] / num[k]) : 0;
            for (i64 i = 0; i < N; i++) {
                uint8_t gg = gb[k * N + i];
                Z[(size_t)i * ORC_COVBLOCK + k] = (gg <= 2) ? ((double)gg - avg) : 0.0;
            }
            double af = 0.5 * avg;
            if (afreq_out) afreq_out[l0 + k] = af;
            double d = 4 * af * (1 - af);
            sumden += d;
            const uint8_t *gg = gb + k * N;
            for (i64 j = 0; j < N; j++) {
                if (gg[j] == 1) diagv[j]++;
                else if (gg[j] > 2) {
                    double *row = denom + tri_index(N, j, j);
                    for (i64 c = 0; c < N - j; c++) row[c] += d;
                    for (i64 r = j - 1; r >= 0; r--)
                        if (gg[r] <= 2) denom[tri_index(N, r, j)] += d;
                }
            }
        }
        for (i64 i = 0; i < N; i++)
            for (i64 k = nsnp; k < ORC_COVBLOCK; k++) Z[(size_t)i * ORC_COVBLOCK + k] = 0;
        muladd_block(Z, N, ibd_tri);
    }
    if (diagadj)
        for (i64 i = 0; i < N; i++) ibd_tri[tri_index(N, i, i)] -= diagv[i];
    for (i64 k = 0; k < np; k++) ibd_tri[k] /= (sumden - denom[k]);
    free(Z); free(denom); free(diagv);
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

/* ---------------------------------------------------------------------------
 * Counter-based synthetic genotypes: C twin of oracle/synth.py:synth_hash_geno for spectra 0, 1, 2 (test / bench utility, no
 * reference counterpart; the generator itself is snpgpu_synth_block, kernels_prep.hip).  The numpy twin defines it and
 * tests/test_cpu_host.py compares the two cell by cell; this form only makes the fp64 anchors of the full-size checks
 * (tests/fp64_anchor.py: ~650 samples x 1e6 SNPs) a matter of seconds.  out: uint8 [n_snp][n_samples].  */
static inline uint32_t orc_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

void orc_synth_hash_geno(const i64 *samples, i64 n_samples, i64 snp_begin, i64 n_snp, uint32_t seed, uint32_t miss32,
                         int spectrum, int special, uint8_t *out)
{
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < n_snp; k++) {
        const uint32_t snp = (uint32_t)(snp_begin + k);
        const uint32_t ks = orc_mix32(seed ^ orc_mix32(snp + 0x9E3779B9u));
        const uint64_t u = orc_mix32(ks ^ 0xA5A5A5A5u) >> 16;
        uint32_t t;
        if (spectrum == 1) t = (uint32_t)((u * u * u) >> 33);
        else if (spectrum == 2) t = (uint32_t)(655u + ((u * 32113u) >> 16));
        else t = (uint32_t)(3277u + ((u * 58982u) >> 16));
        const i64 m = (snp_begin + k) % 997;
        uint8_t *row = out + k * n_samples;
        for (i64 j = 0; j < n_samples; j++) {
            const uint32_t h = orc_mix32(ks ^ ((uint32_t)samples[j] * 0x9E3779B1u));
            uint8_t g = (uint8_t)(((h & 0xFFFFu) < t) + ((h >> 16) < t));
            if (miss32 && orc_mix32(h ^ 0x68E31DA4u) < miss32) g = 3;
            if (special) { if (m == 3) g = 0; else if (m == 5) g = 2; else if (m == 7) g = 3; }
            row[j] = g;
        }
    }
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
