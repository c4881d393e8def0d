// Finalisers: rectangular panel accumulators -> the reference's output layouts
// (packed upper triangle slab, CdMatTri src/dGenGWAS.h:511-583, or the full symmetric matrix that
// gnrIBSNum / gnrIBSAve / grm_output build, src/genIBS.cpp:463-543, src/genPCA.cpp:1586-1602).
#include "snpgpu_internal.h"

#include <math.h>

namespace snpgpu {

struct OutPos {
    int64_t a, b;  // primary index and mirror index (-1 = none)
};

__device__ __forceinline__ int64_t tri_idx(int64_t N, int64_t i, int64_t j) { return j + i * (2 * N - i - 1) / 2; }

// grid: (ceil((N-col0)/256), groups of FIN_ROWS panel rows); F::apply(rel, relf, i, j, pos): element offsets in the uint32 / fp64 planes.
// A workgroup walks FIN_ROWS consecutive rows of its 256 columns (round 4: one row per workgroup left the finaliser latency-bound --
// 25 million workgroups of one or two elements per thread at N = 100 000: 30 ms for the 100 GB of a GCTA panel).
constexpr int FIN_ROWS = 32;
template <class F>
__global__ __launch_bounds__(256) void fin_kernel(PanelGeom g, int packed, F f)
{
    const int64_t j = g.col0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= g.N) return;
    for (int64_t ib = g.row0 + (int64_t)blockIdx.y * FIN_ROWS; ib < g.row1; ib += (int64_t)gridDim.y * FIN_ROWS) {   // grid.y is capped at 65535
        const int64_t ie = (ib + FIN_ROWS < g.row1) ? (ib + FIN_ROWS) : g.row1;
        if (j < ib) continue;                                   // (the whole group lies right of this column only from row j on)
#pragma unroll 4
        for (int64_t i = ib; i < ie; i++) {
            if (j < i) break;
            const int64_t rel = (i - g.row0) * g.ncols_pad + (j - g.col0);                         // uint32 planes: row-major
            const int64_t relf = acc_off(g.ncols_pad, g.f64_tiles_c, i - g.row0, j - g.col0);      // fp64 planes
            OutPos pos;
            if (packed == 2) {          // the panel rectangle itself (in-place finalisation: out == the fp64 accumulator plane)
                pos.a = relf;
                pos.b = -1;
            } else if (packed) {
                pos.a = tri_idx(g.N, i, j) - tri_idx(g.N, g.row0, g.row0);
                pos.b = -1;
            } else {
                pos.a = i * g.N + j;
                pos.b = (i == j) ? -1 : (j * g.N + i);
            }
            f.apply(rel, relf, i, j, pos);
        }
    }
}

template <class F>
static int run_fin(hipStream_t st, const PanelGeom &g, int packed, const F &f)
{
    const int64_t nrows = g.row1 - g.row0;
    if (nrows <= 0) return 0;
    const int64_t ngrp = (nrows + FIN_ROWS - 1) / FIN_ROWS;
    dim3 grid((unsigned)((g.N - g.col0 + 255) / 256), (unsigned)(ngrp < 65535 ? ngrp : 65535));
    hipLaunchKernelGGL(fin_kernel<F>, grid, dim3(256), 0, st, g, packed, f);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- IBS -------------------------------------------------------------------
struct FinIbsNum {
    const uint32_t *acc; int64_t plane; int32_t *o0, *o1, *o2;
    __device__ void apply(int64_t rel, int64_t relf, int64_t, int64_t, OutPos p) const
    {
        const uint32_t n = acc[rel], c1 = acc[plane + rel], c0 = acc[2 * plane + rel] >> 1;   // the plane holds 2 ibs0
        const int32_t v0 = (int32_t)c0, v1 = (int32_t)c1, v2 = (int32_t)(n - c0 - c1);
        o0[p.a] = v0; o1[p.a] = v1; o2[p.a] = v2;
        if (p.b >= 0) { o0[p.b] = v0; o1[p.b] = v1; o2[p.b] = v2; }
    }
};
int launch_fin_ibs_num(hipStream_t st, const PanelGeom &g, const uint32_t *acc, int32_t *o0, int32_t *o1,
                       int32_t *o2, int packed)
{
    FinIbsNum f{acc, g.rows_pad * g.ncols_pad, o0, o1, o2};
    return run_fin(st, g, packed, f);
}

struct FinIbsAve {
    const uint32_t *acc; int64_t plane; double *out;
    __device__ void apply(int64_t rel, int64_t relf, int64_t, int64_t, OutPos p) const
    {
        const uint32_t n = acc[rel], c1 = acc[plane + rel], c0 = acc[2 * plane + rel] >> 1;   // the plane holds 2 ibs0
        const uint32_t c2 = n - c0 - c1;
        // (0.5*IBS1 + IBS2) / (IBS0+IBS1+IBS2) with the reference's uint32 sum, genIBS.cpp:475
        const double v = (0.5 * c1 + c2) / (double)(uint32_t)(c0 + c1 + c2);
        out[p.a] = v;
        if (p.b >= 0) out[p.b] = v;
    }
};
int launch_fin_ibs_ave(hipStream_t st, const PanelGeom &g, const uint32_t *acc, double *out, int packed)
{
    FinIbsAve f{acc, g.rows_pad * g.ncols_pad, out};
    return run_fin(st, g, packed, f);
}

// ---- KING robust -------------------------------------------------------------
// kernel counters {nLoci, ibs1, ibs0, N1, N2} -> TS_KINGRobust {IBS0, nLoci, SumSq, N1_Aa, N2_Aa}
struct FinKingCounts {
    const uint32_t *acc; int64_t plane; uint32_t *out;
    __device__ void apply(int64_t rel, int64_t relf, int64_t, int64_t, OutPos p) const
    {
        const uint32_t n = acc[rel], c1 = acc[plane + rel], c0 = acc[2 * plane + rel] >> 1;   // the plane holds 2 ibs0
        uint32_t *o = out + 5 * p.a;
        o[0] = c0; o[1] = n; o[2] = c1 + 4u * c0; o[3] = acc[3 * plane + rel]; o[4] = acc[4 * plane + rel];
    }
};
int launch_fin_king_counts(hipStream_t st, const PanelGeom &g, const uint32_t *acc, uint32_t *out5)
{
    FinKingCounts f{acc, g.rows_pad * g.ncols_pad, out5};
    return run_fin(st, g, 1, f);
}

struct FinKingRobust {
    const uint32_t *acc; int64_t plane; const int32_t *fam; double *ibs0, *kin;
    __device__ void apply(int64_t rel, int64_t relf, int64_t i, int64_t j, OutPos p) const
    {
        double vi, vk;
        if (i == j) {           // genKING.cpp:623
            vi = 0; vk = 0.5;
        } else {
            const uint32_t n = acc[rel], c1 = acc[plane + rel], c0 = acc[2 * plane + rel] >> 1;   // the plane holds 2 ibs0
            const uint32_t n1 = acc[3 * plane + rel], n2 = acc[4 * plane + rel];
            const uint32_t sumsq = c1 + 4u * c0;
            vi = (n > 0) ? ((double)c0 / n) : (double)NAN;
            const int f1 = fam ? fam[i] : -1, f2 = fam ? fam[j] : -1;
            double v = (f1 == f2 && f1 >= 0) ? (0.5 - sumsq / (2.0 * (uint32_t)(n1 + n2)))
                                             : (0.5 - sumsq / (4.0 * (n1 < n2 ? n1 : n2)));
            if (!isfinite(v)) v = (double)NAN;
            vk = v;
        }
        ibs0[p.a] = vi; kin[p.a] = vk;
        if (p.b >= 0) { ibs0[p.b] = vi; kin[p.b] = vk; }
    }
};
int launch_fin_king_robust(hipStream_t st, const PanelGeom &g, const uint32_t *acc, const int32_t *family,
                           double *ibs0, double *kin, int packed)
{
    FinKingRobust f{acc, g.rows_pad * g.ncols_pad, family, ibs0, kin};
    return run_fin(st, g, packed, f);
}

// ---- KING homo ---------------------------------------------------------------
struct FinKingHomo {
    const uint32_t *acc; const double *facc; int64_t plane; double fscale; double *k0, *k1; const double *wc;
    // round 5: blocks with missing calls leave B_ij = sum c mu_i mu_j in the planes and per-sample sums M in msum[2][ncols_pad]:
    // masked sum = C - M_i - M_j + B_ij with C in wc (the totals of ALL blocks then); msum == nullptr: the planes hold the masked sums
    const double *msum; int64_t col0, ncols_pad;
    __device__ void apply(int64_t rel, int64_t relf, int64_t i, int64_t j, OutPos p) const
    {
        double a = 0, b = 0;
        if (i != j) {           // genKING.cpp:526-537
            const uint32_t c1 = acc[rel], c0 = acc[plane + rel] >> 1;   // the plane holds 2 ibs0
            const uint32_t sumsq = c1 + 4u * c0;
            // tables may be pre-scaled; blocks without missing calls contribute the same sum to every pair (wc)
            double saf = facc[relf] * fscale + (wc ? wc[0] : 0.0), saf2 = facc[plane + relf] * fscale + (wc ? wc[1] : 0.0);
            if (msum) {
                saf -= msum[i - col0] + msum[j - col0];
                saf2 -= msum[ncols_pad + i - col0] + msum[ncols_pad + j - col0];
            }
            const double theta = 0.5 - sumsq / (8 * saf);
            const double v0 = c0 / (2 * saf2);
            const double v1 = 2 - 2 * v0 - 4 * theta;
            a = isfinite(v0) ? v0 : (double)NAN;
            b = isfinite(v1) ? v1 : (double)NAN;
        }
        k0[p.a] = a; k1[p.a] = b;
        if (p.b >= 0) { k0[p.b] = a; k1[p.b] = b; }
    }
};
int launch_fin_king_homo(hipStream_t st, const PanelGeom &g, const uint32_t *acc, const double *facc, double fscale,
                         double *k0, double *k1, int packed, const double *w_const, const double *msum)
{
    FinKingHomo f{acc, facc, g.rows_pad * g.ncols_pad, fscale, k0, k1, w_const, msum, g.col0, g.ncols_pad};
    return run_fin(st, g, packed, f);
}

// ---- GCTA / covariance ---------------------------------------------------------
// Denom(i,j) = #polymorphic SNPs where i or j is missing = M(i,i) + M(j,j) - M(i,j) with
// M = both-missing counts; result = num / (2 (nLocus - Denom)), no guard (genPCA.cpp:1232-1236).
struct FinGcta {
    const double *num; const uint32_t *miss; const unsigned long long *nlocus;
    const uint32_t *diag;  // M(s,s) for every sample s (absolute index)
    double *out;
    // the pending column / row / constant terms of the fp16 SYRK kernels (colterm_settle_kernel's job, folded in here: the
    // sums stay as they are, so further blocks may follow): num - (T[c] + Q[c] + R[r] - K), panel-relative r, c
    const double *colterm, *uvterm; int64_t row0, col0, ncols_pad;
    __device__ void apply(int64_t rel, int64_t relf, int64_t i, int64_t j, OutPos p) const
    {
        const long long nl = (long long)*nlocus;
        const long long den = (long long)diag[i] + (long long)diag[j] - (long long)miss[rel];
        double s = num[relf];
        if (colterm) {
            const int64_t r = i - row0, c = j - col0;
            s -= colterm[c] + (uvterm ? uvterm[ncols_pad + c] + uvterm[r] - uvterm[2 * ncols_pad] : 0.0);
        }
        const double v = s / (double)(2 * (nl - den));
        out[p.a] = v;
        if (p.b >= 0) out[p.b] = v;
    }
};

struct FinCov {
    const double *num; double scale; double *out;
    __device__ void apply(int64_t rel, int64_t relf, int64_t, int64_t, OutPos p) const
    {
        const double v = num[relf] * scale;
        out[p.a] = v;
        if (p.b >= 0) out[p.b] = v;
    }
};
int launch_fin_cov(hipStream_t st, const PanelGeom &g, const double *num, double scale, double *out, int packed)
{
    FinCov f{num, scale, out};
    return run_fin(st, g, packed, f);
}

// ---- PLINK method of moments ------------------------------------------------------
// Est_PLINK_Kinship, src/genIBD.cpp:341-390; kernel counters {n, ibs1, ibs0}
struct FinMom {
    const uint32_t *acc; int64_t plane; double e00, e01, e02, e11, e12; int constraint; double *k0, *k1;
    __device__ void apply(int64_t rel, int64_t relf, int64_t i, int64_t j, OutPos p) const
    {
        double a = 0, b = 0;
        if (i != j) {
            const int n012 = (int)acc[rel], IBS1 = (int)acc[plane + rel], IBS0 = (int)(acc[2 * plane + rel] >> 1);
            const int IBS2 = n012 - IBS0 - IBS1;
            const double f00 = e00 * n012, f01 = e01 * n012, f11 = e11 * n012, f02 = e02 * n012, f12 = e12 * n012,
                         f22 = 1.0 * n012;
            double v0 = IBS0 / f00;
            double v1 = (IBS1 - v0 * f01) / f11;
            double v2 = (IBS2 - v0 * f02 - v1 * f12) / f22;
            if (v0 > 1) { v0 = 1; v1 = v2 = 0; }
            if (v1 > 1) { v1 = 1; v0 = v2 = 0; }
            if (v2 > 1) { v2 = 1; v0 = v1 = 0; }
            if (v0 < 0) { const double S = v1 + v2; v1 /= S; v2 /= S; v0 = 0; }
            if (v1 < 0) { const double S = v0 + v2; v0 /= S; v2 /= S; v1 = 0; }
            if (v2 < 0) { const double S = v0 + v1; v0 /= S; v1 /= S; v2 = 0; }
            if (constraint) {
                v2 = 1 - v0 - v1;
                const double pihat = v1 / 2 + v2;
                if (pihat * pihat < v2) { v0 = (1 - pihat) * (1 - pihat); v1 = 2 * pihat * (1 - pihat); }
            }
            a = v0; b = v1;
        }
        k0[p.a] = a; k1[p.a] = b;
        if (p.b >= 0) { k0[p.b] = a; k1[p.b] = b; }
    }
};
int launch_fin_mom(hipStream_t st, const PanelGeom &g, const uint32_t *acc, const double *e, int constraint,
                   double *k0, double *k1, int packed)
{
    FinMom f{acc, g.rows_pad * g.ncols_pad, e[0], e[1], e[2], e[3], e[4], constraint, k0, k1};
    return run_fin(st, g, packed, f);
}

// ---- EIGMIX -------------------------------------------------------------------------
// ibd = (num - diagadj*het_i [i==j]) / (SumDenominator - Denom),  Denom(i,j) = sum of 4p(1-p) over
// the SNPs where i or j is missing = dmiss[i] + dmiss[j] - dd(i,j)   (src/genEIGMIX.cpp:113-155)
struct FinEigmix {
    const double *num, *dd; const uint32_t *het; const double *dmiss, *dsq; const double *sumden; int diagadj; double scale;
    double *out;
    __device__ void apply(int64_t rel, int64_t relf, int64_t i, int64_t j, OutPos p) const
    {
        double v = (i == j) ? dsq[i] : num[relf];     // diagonal numerator from the fp64 per-sample sums
        if (i == j && diagadj) v -= (double)het[i];
        v = v / (*sumden - (dmiss[i] + dmiss[j] - dd[relf])) * scale;
        out[p.a] = v;
        if (p.b >= 0) out[p.b] = v;
    }
};
int launch_fin_eigmix(hipStream_t st, const PanelGeom &g, const double *num, const double *dd, const uint32_t *het,
                      const double *dmiss, const double *dsq, const double *d_sumden, int diagadj, double scale,
                      double *out, int packed)
{
    FinEigmix f{num, dd, het, dmiss, dsq, d_sumden, diagadj, scale, out};
    return run_fin(st, g, packed, f);
}

// ---- individual beta ------------------------------------------------------------------
// kernel counters {num, x1, x2}: ibscnt = x1 + 2*x2 (src/genBeta.cpp:170-176)
__device__ __forceinline__ double beta_raw(const uint32_t *acc, int64_t plane, int64_t rel, bool diag_m1)
{
    const uint32_t num = acc[rel], ibscnt = acc[plane + rel] + 2u * acc[2 * plane + rel];
    return diag_m1 ? ((double)ibscnt / num - 1) : ((0.5 * ibscnt) / num);
}

// min over all entries and sum over the off-diagonal entries of the raw values
__global__ __launch_bounds__(256) void beta_reduce_kernel(PanelGeom g, const uint32_t *__restrict__ acc,
                                                          int diag_inbreeding, double *__restrict__ pmin,
                                                          double *__restrict__ psum)
{
    const int64_t plane = g.rows_pad * g.ncols_pad;
    double mn = 1e300, sm = 0;
    for (int64_t i = g.row0 + blockIdx.x; i < g.row1; i += gridDim.x)
        for (int64_t j = i + threadIdx.x; j < g.N; j += 256) {
            const int64_t rel = (i - g.row0) * g.ncols_pad + (j - g.col0);
            const double v = beta_raw(acc, plane, rel, (i == j) && diag_inbreeding);
            if (v < mn) mn = v;
            if (i != j) sm += v;
        }
    __shared__ double rmin[256], rsum[256];
    rmin[threadIdx.x] = mn; rsum[threadIdx.x] = sm;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            rsum[threadIdx.x] += rsum[threadIdx.x + off];
            if (rmin[threadIdx.x + off] < rmin[threadIdx.x]) rmin[threadIdx.x] = rmin[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { pmin[blockIdx.x] = rmin[0]; psum[blockIdx.x] = rsum[0]; }
}
int launch_beta_reduce(hipStream_t st, const PanelGeom &g, const uint32_t *acc, int diag_inbreeding, double *partial_min,
                       double *partial_sum, int nblocks)
{
    hipLaunchKernelGGL(beta_reduce_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, g, acc, diag_inbreeding,
                       partial_min, partial_sum);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

struct FinBeta {
    const uint32_t *acc; int64_t plane; int mode; double avg, mn; double *out;
    __device__ void apply(int64_t rel, int64_t relf, int64_t i, int64_t j, OutPos p) const
    {
        double v;
        if (mode == 2) {                 // CalcIndivBetaGRM, src/genBeta.cpp:263-357
            const double r = beta_raw(acc, plane, rel, i == j);
            const double scale = 2.0 / (1 - mn);
            v = (i == j) ? ((r - mn) * scale * 0.5 + 1) : ((r - mn) * scale);
        } else {                         // gnrIBD_Beta, src/genBeta.cpp:384-452
            const double r = beta_raw(acc, plane, rel, (i == j) && mode == 1);
            v = (r - avg) * (1.0 / (1 - avg));
        }
        out[p.a] = v;
        if (p.b >= 0) out[p.b] = v;
    }
};
int launch_fin_beta(hipStream_t st, const PanelGeom &g, const uint32_t *acc, int mode, double avg, double mn, double *out,
                    int packed)
{
    FinBeta f{acc, g.rows_pad * g.ncols_pad, mode, avg, mn, out};
    return run_fin(st, g, packed, f);
}

__global__ __launch_bounds__(256) void trace_kernel(PanelGeom g, const double *__restrict__ num,
                                                    double *__restrict__ d_trace)
{
    double s = 0;
    for (int64_t i = g.row0 + threadIdx.x; i < g.row1; i += 256)
        s += num[acc_off(g.ncols_pad, g.f64_tiles_c, i - g.row0, i - g.col0)];
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *d_trace = red[0];
}
int launch_trace(hipStream_t st, const PanelGeom &g, const double *num, double *d_trace)
{
    hipLaunchKernelGGL(trace_kernel, dim3(1), dim3(256), 0, st, g, num, d_trace);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// copy the upper triangle of the panel's diagonal block (rows I x columns I) onto its lower
// triangle, so that the block can be used as a dense symmetric operand
__global__ __launch_bounds__(256) void mirror_diag_kernel(PanelGeom g, double *__restrict__ num)
{
    const int64_t nI = g.row1 - g.row0;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;      // column within the block
    for (int64_t i = blockIdx.y; i < nI; i += gridDim.y)            // row within the block
        if (j < i) num[acc_off(g.ncols_pad, g.f64_tiles_c, i, j)] = num[acc_off(g.ncols_pad, g.f64_tiles_c, j, i)];
}
// the same inside the T x T tiles on the diagonal only (all the one-pass symmetric panel product reads below
// the diagonal)
__global__ __launch_bounds__(256) void mirror_diag_tiles_kernel(PanelGeom g, double *__restrict__ num, int T)
{
    const int64_t nI = g.row1 - g.row0;
    const int64_t t0 = (int64_t)blockIdx.x * T;
    for (int e = threadIdx.x; e < T * T; e += 256) {
        const int64_t i = t0 + e / T, j = t0 + e % T;
        if (j < i && i < nI) num[acc_off(g.ncols_pad, g.f64_tiles_c, i, j)] = num[acc_off(g.ncols_pad, g.f64_tiles_c, j, i)];
    }
}
int launch_mirror_diag_tiles(hipStream_t st, const PanelGeom &g, double *num, int T)
{
    const int64_t nI = g.row1 - g.row0;
    if (nI <= 1) return 0;
    hipLaunchKernelGGL(mirror_diag_tiles_kernel, dim3((unsigned)((nI + T - 1) / T)), dim3(256), 0, st, g, num, T);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_mirror_diag(hipStream_t st, const PanelGeom &g, double *num)
{
    const int64_t nI = g.row1 - g.row0;
    if (nI <= 1) return 0;
    dim3 grid((unsigned)((nI + 255) / 256), (unsigned)(nI < 65535 ? nI : 65535));
    hipLaunchKernelGGL(mirror_diag_kernel, grid, dim3(256), 0, st, g, num);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// GCTA needs M(s,s) for every column sample, which lies outside a non-full panel's rows: a
// per-sample count vector `diag` (absolute sample index) is accumulated by miss_diag_kernel below.
int launch_fin_gcta(hipStream_t st, const PanelGeom &g, const double *num, const uint32_t *miss,
                    const uint32_t *diag, const unsigned long long *d_nlocus, double *out, int packed,
                    const double *colterm, const double *uvterm)
{
    FinGcta f{num, miss, d_nlocus, diag, out, colterm, uvterm, g.row0, g.col0, g.ncols_pad};
    return run_fin(st, g, packed, f);
}

// Rank-one terms of the two-product pair kernel (blocks without missing calls): ibs1 += H_r + H_c, 2 ibs0 += 2 (T_r + T_c), and for KING-robust
// N1_Aa += H_r, N2_Aa += H_c, over the whole panel rectangle; then the counts start over.
__global__ __launch_bounds__(256) void het_settle_kernel(uint32_t *__restrict__ acc, int64_t plane, int64_t rows_pad,
                                                         int64_t ncols_pad, const uint32_t *__restrict__ het, int king,
                                                         int p1, int p0)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= ncols_pad) return;
    const uint32_t hc = het[c];
    for (int64_t r = blockIdx.y; r < rows_pad; r += gridDim.y) {
        const uint32_t hr = het[r];           // panel-relative rows and columns start at the same sample
        const int64_t e = r * ncols_pad + c;
        const uint32_t tr = het[ncols_pad + r], tc = het[ncols_pad + c];                    // #(g == 2) of row and column sample
        if (hr + hc) acc[p1 * plane + e] += hr + hc;                                         // ibs1 += H_r + H_c
        if (tr + tc) acc[p0 * plane + e] += 2u * (tr + tc);                                  // 2 ibs0 += 2 (T_r + T_c)
        if (king) {
            if (hr) acc[3 * plane + e] += hr;
            if (hc) acc[4 * plane + e] += hc;
        }
    }
}

int launch_het_settle(hipStream_t st, uint32_t *acc, int64_t plane, int64_t rows_pad, int64_t ncols_pad, uint32_t *het,
                      int king, int plane_ibs1, int plane_ibs0x2)
{
    dim3 grid((unsigned)((ncols_pad + 255) / 256), (unsigned)std::min<int64_t>(rows_pad, 4096));
    hipLaunchKernelGGL(het_settle_kernel, grid, dim3(256), 0, st, acc, plane, rows_pad, ncols_pad, het, king, plane_ibs1,
                       plane_ibs0x2);
    SNPGPU_HIP_CHECK(hipGetLastError());
    SNPGPU_HIP_CHECK(hipMemsetAsync(het, 0, sizeof(uint32_t) * (size_t)(2 * ncols_pad), st));
    return 0;
}

// per-sample popcount of the missing plane (uint2 words, word-major colp layout) added to diag[s]
__global__ __launch_bounds__(256) void miss_diag_kernel(const uint2 *__restrict__ colp, int KWv, int64_t ncols_pad,
                                                        int64_t col0, uint32_t *__restrict__ diag,
                                                        const unsigned long long *__restrict__ skip)
{
    if (skip && *skip == 0ull) return;
    const int64_t sc = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (sc >= ncols_pad) return;
    uint32_t c = 0;
    for (int k = 0; k < KWv; k++) {
        const uint2 v = colp[(int64_t)k * ncols_pad + sc];
        c += __popc(v.x) + __popc(v.y);
    }
    diag[col0 + sc] += c;
}
int launch_miss_diag(hipStream_t st, const uint2 *colp, int KWv, int64_t ncols_pad, int64_t col0, uint32_t *diag,
                     const unsigned long long *skip)
{
    hipLaunchKernelGGL(miss_diag_kernel, dim3((unsigned)((ncols_pad + 255) / 256)), dim3(256), 0, st, colp, KWv,
                       ncols_pad, col0, diag, skip);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// Exact-row SYRK: the kernel accumulates sum_s (g_is - c_s) w_js; the part - sum_s (avg_s - c_s) w_js = - T[j] is the same
// for every row i and is applied here, once per result request: acc[i][j] -= T[j] for the real rows i and the stored
// columns j >= 256 floor(i / 256) (the tiles that touch the upper trapezoid), then T is cleared.
__global__ __launch_bounds__(256) void colterm_settle_kernel(double *__restrict__ acc, int64_t ld, int64_t tiles_c,
                                                             int64_t n_rows_real, int64_t ncols_pad, int64_t n_cols_real,
                                                             const double *__restrict__ colterm,
                                                             const double *__restrict__ uvterm)
{
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // (the padding columns stay zero: the row terms R[i] are not zero there, and the eigen solver's panel product reads the
    // columns up to the next multiple of 16)
    if (col >= n_cols_real) return;
    // single-product blocks (syrk_uv_kernel): acc[i][j] -= R[i] + Q[j] - K, uvterm = {R[ncols_pad], Q[ncols_pad], K}
    const double t = colterm[col] + (uvterm ? uvterm[ncols_pad + col] - uvterm[2 * ncols_pad] : 0.0);
    if (!uvterm && t == 0.0) return;
    const int64_t r_end_all = (col / 256 + 1) * 256;                 // rows whose tile row starts at or left of this column
    const int64_t r_end = r_end_all < n_rows_real ? r_end_all : n_rows_real;
    const int64_t per = (r_end + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * per, r1 = (r0 + per < r_end) ? (r0 + per) : r_end;
    if (uvterm) for (int64_t r = r0; r < r1; r++) acc[acc_off(ld, tiles_c, r, col)] -= t + uvterm[r];
    else for (int64_t r = r0; r < r1; r++) acc[acc_off(ld, tiles_c, r, col)] -= t;
}

int launch_colterm_settle(hipStream_t st, double *acc, int64_t ld, int64_t tiles_c, int64_t n_rows_real, int64_t ncols_pad,
                          int64_t n_cols_real, double *colterm, double *uvterm)
{
    if (n_rows_real <= 0) return 0;
    int gy = (int)((n_rows_real + 255) / 256);
    if (gy > 256) gy = 256;
    hipLaunchKernelGGL(colterm_settle_kernel, dim3((unsigned)((ncols_pad + 255) / 256), (unsigned)gy), dim3(256), 0, st, acc, ld,
                       tiles_c, n_rows_real, ncols_pad, n_cols_real, colterm, uvterm);
    SNPGPU_HIP_CHECK(hipGetLastError());
    SNPGPU_HIP_CHECK(hipMemsetAsync(colterm, 0, sizeof(double) * (size_t)ncols_pad, st));
    if (uvterm) SNPGPU_HIP_CHECK(hipMemsetAsync(uvterm, 0, sizeof(double) * (size_t)(2 * ncols_pad + 2), st));
    return 0;
}

}  // namespace snpgpu
