// Top-k eigenpairs of a symmetric matrix that exists only as fp64 row panels on one or several GPUs: the scalable
// replacement for the reference's dense LAPACK call (CalcEigen -> dspevx('V','I',IL=1..IU=k), src/genPCA.cpp:1262-1346;
// the same call behind gnrEigMix, src/genEIGMIX.cpp:700-702), which is impossible beyond a few 10^4 samples.
//
// Method: thick-restarted block Krylov (block Lanczos with full re-orthogonalisation) + Rayleigh-Ritz, all in C++ / HIP
// behind the C ABI (round 2 had it in Python + torch, out of reach of the R shim).  The only O(N^2) work, Y = C Q, is the
// one-pass symmetric panel product (kernels_eig.hip) behind an `EigOperator`; the tall-skinny algebra runs on the
// operator's device: Gram matrices as split-K batched rocBLAS products (a plain GEMM with a 48 x 576 output and an inner
// dimension of 5e5 occupies a handful of workgroups), CholeskyQR2 with a Householder (hipSOLVER geqrf / orgqr) fallback,
// hipSOLVER syevd for the projected matrix.  Blocks of vectors are stored vector-major, double [b][n] (= column-major
// n x b), which is also the ABI's eigenvector layout.  Eigenvector signs are arbitrary, as with LAPACK.
// Round 3 (krylov_topk): the solver keeps W = C K for its whole basis, so a restart carries a third of the basis as Ritz
// vectors WITH their products (thick restart at no panel product); all products but the first of a cycle run on fp32 matrix
// instructions, and the fp64 product of the vectors a cycle starts from -- their true residual -- is what accepts the result.
#include <hipsolver/hipsolver.h>
#include <rocblas/rocblas.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "eigen.h"

namespace snpgpu {

namespace {

// below CHOLQR_MIN_N samples the Householder QR is cheap enough (and rank-robust); GRAM_CHUNK = split-K chunk of the Gram
// products.  SNPGPU_EIG_CHOLQR_MIN_N / SNPGPU_EIG_GRAM_CHUNK override them (tests run the large-n algebra at small n).
static int64_t env_i64(const char *name, int64_t dflt, int64_t lo, int64_t hi)
{
    if (const char *e = getenv(name)) { const long long v = atoll(e); if (v >= lo && v <= hi) return v; }
    return dflt;
}
static int64_t CHOLQR_MIN_N = 32768, GRAM_CHUNK = 8192;

__device__ inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// counter-based standard normals: element e of the stream `seed` (identical on every device and rank)
__global__ __launch_bounds__(256) void randn_kernel(double *__restrict__ x, size_t count, uint32_t seed, uint64_t offset)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    const uint64_t g = offset + e;
    const uint32_t a = mix32((uint32_t)g ^ mix32(seed ^ (uint32_t)(g >> 32) ^ 0x9E3779B9U));
    const uint32_t b = mix32(a ^ 0x85EBCA6BU), c = mix32(b ^ 0xC2B2AE35U);
    const double u1 = ((double)a * 4294967296.0 + (double)b + 1.0) * (1.0 / 18446744073709551616.0);   // (0, 1]
    const double u2 = (double)c * (1.0 / 4294967296.0);
    x[e] = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}

// out[r][chunk] = sum over the chunk of (a[r][t] - s[r] * b[r][t])^2   (b == nullptr: plain squared row norms); grid (chunks, rows)
__global__ __launch_bounds__(256) void rowdiff_norm2_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                                           const double *__restrict__ s, int64_t n, double *__restrict__ out)
{
    const int r = blockIdx.y;
    const double f = b ? s[r] : 0.0;
    double acc = 0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const double v = a[(int64_t)r * n + t] - (b ? f * b[(int64_t)r * n + t] : 0.0);
        acc += v * v;
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_down(acc, o);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    // one partial per workgroup, summed by the host in a fixed order: every rank of a multi-process run must take the same
    // decisions from these norms, bit for bit (no atomics)
    if (threadIdx.x == 0) out[(size_t)r * gridDim.x + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// out[e] = sum over the batch of part[s][e]
__global__ __launch_bounds__(256) void batch_sum_kernel(const double *__restrict__ part, int batch, size_t elems,
                                                        double *__restrict__ out, int accumulate)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= elems) return;
    double s = accumulate ? out[e] : 0.0;
    for (int k = 0; k < batch; k++) s += part[(size_t)k * elems + e];
    out[e] = s;
}

// t[i][j] = K_i . W_j with W = C K up to rounding: the symmetric matrix it stands for.  W_j for j < exact comes from fp64
// products and the other columns from fp32 ones (a mixed cycle, krylov_topk): where only one of the two entries of a pair
// is of the exact kind it is taken alone -- averaging would put the fp32 error into the coupling of the cycle's first block
// with the rest, which is what the corrections of nearly converged vectors are made of
__global__ __launch_bounds__(256) void symmetrize_kernel(double *__restrict__ t, int m, int exact)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= m * m) return;
    const int i = e / m, j = e % m;
    if (j > i) {
        const double v = (i < exact && j >= exact) ? t[(size_t)j * m + i] : 0.5 * (t[(size_t)i * m + j] + t[(size_t)j * m + i]);
        t[(size_t)i * m + j] = v;
        t[(size_t)j * m + i] = v;
    }
}

struct Solver {
    EigOperator &op;
    int dev;
    int64_t n;
    hipStream_t st = nullptr;
    rocblas_handle bl = nullptr;
    hipsolverHandle_t sv = nullptr;
    DevBuf gpart, gsmall, tau, swork, info, norms;
    std::string err;

    explicit Solver(EigOperator &o) : op(o), dev(o.device()), n(o.n()) {}
    ~Solver()
    {
        (void)hipSetDevice(dev);
        if (st) (void)hipStreamSynchronize(st);
        DevBuf *all[] = {&gpart, &gsmall, &tau, &swork, &info, &norms};
        for (DevBuf *b : all) b->release();
        if (bl) (void)rocblas_destroy_handle(bl);
        if (sv) (void)hipsolverDestroy(sv);
        if (st) (void)hipStreamDestroy(st);
    }
    int fail(const std::string &m) { set_error("top-k eigen solver: " + m); return 1; }
    int init()
    {
        CHOLQR_MIN_N = env_i64("SNPGPU_EIG_CHOLQR_MIN_N", 32768, 0, (int64_t)1 << 40);
        GRAM_CHUNK = env_i64("SNPGPU_EIG_GRAM_CHUNK", 8192, 16, 1 << 20);
        SNPGPU_HIP_CHECK(hipSetDevice(dev));
        SNPGPU_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        if (rocblas_create_handle(&bl) != rocblas_status_success) return fail("rocblas_create_handle failed");
        rocblas_set_stream(bl, st);
        rocblas_set_pointer_mode(bl, rocblas_pointer_mode_host);
        rocblas_set_atomics_mode(bl, rocblas_atomics_not_allowed);      // reproducible products (ranks must agree bit for bit)
        if (hipsolverCreate(&sv) != HIPSOLVER_STATUS_SUCCESS) return fail("hipsolverCreate failed");
        hipsolverSetStream(sv, st);
        if (info.alloc(sizeof(int)) || norms.alloc(sizeof(double) * 4096)) return 1;
        return 0;
    }
    int sync() { SNPGPU_HIP_CHECK(hipStreamSynchronize(st)); return 0; }

    // G[p][q] (row-major, device) = A B^T for A [p][n], B [q][n]
    int gram(const double *A, int p, const double *B, int q, double *G)
    {
        const double one = 1.0, zero = 0.0;
        const size_t pq = (size_t)p * (size_t)q;
        const int64_t batch = (n >= 4 * GRAM_CHUNK) ? n / GRAM_CHUNK : 0;
        const int64_t done = batch * GRAM_CHUNK;
        if (batch > 0) {
            if (gpart.bytes < sizeof(double) * pq * (size_t)batch) {
                if (sync()) return 1;
                gpart.release();
                if (gpart.alloc(sizeof(double) * pq * (size_t)batch)) return 1;
            }
            // column-major: Gc (q x p) = Bc^T Ac with Bc = n x q, Ac = n x p (ld n), K split into chunks of rows
            if (rocblas_dgemm_strided_batched(bl, rocblas_operation_transpose, rocblas_operation_none, q, p, (rocblas_int)GRAM_CHUNK,
                                              &one, B, (rocblas_int)n, GRAM_CHUNK, A, (rocblas_int)n, GRAM_CHUNK, &zero,
                                              (double *)gpart.p, q, (rocblas_stride)pq, (rocblas_int)batch) != rocblas_status_success)
                return fail("rocblas_dgemm_strided_batched failed");
            hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)((pq + 255) / 256)), dim3(256), 0, st, (const double *)gpart.p,
                               (int)batch, pq, G, 0);
        }
        if (done < n) {
            const double beta = batch > 0 ? 1.0 : 0.0;
            if (rocblas_dgemm(bl, rocblas_operation_transpose, rocblas_operation_none, q, p, (rocblas_int)(n - done), &one,
                              B + done, (rocblas_int)n, A + done, (rocblas_int)n, &beta, G, q) != rocblas_status_success)
                return fail("rocblas_dgemm (Gram) failed");
        }
        return 0;
    }
    // R [p][n] -= G [p][q] * B [q][n]
    int sub_gb(double *R, int p, const double *G, const double *B, int q)
    {
        const double m1 = -1.0, one = 1.0;
        if (rocblas_dgemm(bl, rocblas_operation_none, rocblas_operation_none, (rocblas_int)n, p, q, &m1, B, (rocblas_int)n, G, q,
                          &one, R, (rocblas_int)n) != rocblas_status_success)
            return fail("rocblas_dgemm (projection) failed");
        return 0;
    }
    // X [p][n] = S^T-combination: X[i] = sum_j S[j + i * lds] * B[j], S column-major m x p
    int combine(double *X, int p, const double *S, int lds, const double *B, int m)
    {
        const double one = 1.0, zero = 0.0;
        if (rocblas_dgemm(bl, rocblas_operation_none, rocblas_operation_none, (rocblas_int)n, p, m, &one, B, (rocblas_int)n, S, lds,
                          &zero, X, (rocblas_int)n) != rocblas_status_success)
            return fail("rocblas_dgemm (Ritz vectors) failed");
        return 0;
    }
    int small(size_t doubles)
    {
        if (gsmall.bytes >= sizeof(double) * doubles) return 0;
        if (sync()) return 1;
        gsmall.release();
        return gsmall.alloc(sizeof(double) * doubles);
    }
    // r <- r - (r basis^T) basis for orthonormal rows of `basis`
    int project_out(double *R, int p, const double *basis, int q)
    {
        if (q <= 0) return 0;
        if (small((size_t)p * q)) return 1;
        if (gram(R, p, basis, q, (double *)gsmall.p)) return 1;
        return sub_gb(R, p, (const double *)gsmall.p, basis, q);
    }
    // the same for a block that should already be orthogonal to the basis: the coefficients R . basis are looked at on the
    // host first and the subtraction is skipped when none exceeds `tol` (largest one in max_coef)
    int project_out_if(double *R, int p, const double *basis, int q, double tol, double &max_coef)
    {
        max_coef = 0;
        if (q <= 0) return 0;
        if (small((size_t)p * q)) return 1;
        if (gram(R, p, basis, q, (double *)gsmall.p)) return 1;
        std::vector<double> g((size_t)p * q);
        SNPGPU_HIP_CHECK(hipMemcpyAsync(g.data(), gsmall.p, sizeof(double) * g.size(), hipMemcpyDeviceToHost, st));
        if (sync()) return 1;
        for (double v : g) max_coef = std::max(max_coef, std::isfinite(v) ? std::fabs(v) : std::numeric_limits<double>::infinity());
        if (max_coef <= tol) return 0;
        return sub_gb(R, p, (const double *)gsmall.p, basis, q);
    }
    // squared row norms of A (or of A - s .* B) to the host
    int row_norm2(const double *A, const double *B, const double *s_dev, int p, std::vector<double> &out)
    {
        const int gx = (int)std::min<int64_t>((n + 255) / 256, 256);
        if (norms.bytes < sizeof(double) * (size_t)p * gx) { if (sync()) return 1; norms.release(); if (norms.alloc(sizeof(double) * (size_t)p * gx)) return 1; }
        hipLaunchKernelGGL(rowdiff_norm2_kernel, dim3((unsigned)gx, (unsigned)p), dim3(256), 0, st, A, B, s_dev, n, (double *)norms.p);
        std::vector<double> part((size_t)p * gx);
        SNPGPU_HIP_CHECK(hipMemcpyAsync(part.data(), norms.p, sizeof(double) * part.size(), hipMemcpyDeviceToHost, st));
        if (sync()) return 1;
        out.assign((size_t)p, 0.0);
        for (int r = 0; r < p; r++)
            for (int g = 0; g < gx; g++) out[(size_t)r] += part[(size_t)r * gx + g];
        return 0;
    }
    int householder(double *X, int p)
    {
        if (p > n) return fail("more vectors than samples");
        if (tau.bytes < sizeof(double) * (size_t)p) { tau.release(); if (tau.alloc(sizeof(double) * (size_t)p)) return 1; }
        int lw1 = 0, lw2 = 0;
        if (hipsolverDnDgeqrf_bufferSize(sv, (int)n, p, X, (int)n, &lw1) != HIPSOLVER_STATUS_SUCCESS ||
            hipsolverDnDorgqr_bufferSize(sv, (int)n, p, p, X, (int)n, (double *)tau.p, &lw2) != HIPSOLVER_STATUS_SUCCESS)
            return fail("QR workspace query failed");
        const size_t lw = (size_t)std::max(std::max(lw1, lw2), 1);
        if (swork.bytes < sizeof(double) * lw) { if (sync()) return 1; swork.release(); if (swork.alloc(sizeof(double) * lw)) return 1; }
        if (hipsolverDnDgeqrf(sv, (int)n, p, X, (int)n, (double *)tau.p, (double *)swork.p, (int)lw, (int *)info.p) != HIPSOLVER_STATUS_SUCCESS ||
            hipsolverDnDorgqr(sv, (int)n, p, p, X, (int)n, (double *)tau.p, (double *)swork.p, (int)lw, (int *)info.p) != HIPSOLVER_STATUS_SUCCESS)
            return fail("Householder QR failed");
        return 0;
    }
    // orthonormalise the rows of X [p][n] in place.  CholeskyQR2 (two Gram matrices, two p x p Cholesky factors on the host,
    // two triangular solves: all large-output products) when the rows are well conditioned, Householder QR otherwise.
    int orth(double *X, int p)
    {
        if (n < CHOLQR_MIN_N) return householder(X, p);
        std::vector<double> g((size_t)p * p), l((size_t)p * p);
        for (int pass = 0; pass < 2; pass++) {
            if (small((size_t)p * p)) return 1;
            if (gram(X, p, X, p, (double *)gsmall.p)) return 1;
            SNPGPU_HIP_CHECK(hipMemcpyAsync(g.data(), gsmall.p, sizeof(double) * g.size(), hipMemcpyDeviceToHost, st));
            if (sync()) return 1;
            // Cholesky G = L L^T on the host, L stored column-major lower
            bool ok = true;
            std::fill(l.begin(), l.end(), 0.0);
            for (int j = 0; j < p && ok; j++) {
                double d = g[(size_t)j * p + j];
                for (int k = 0; k < j; k++) d -= l[(size_t)k * p + j] * l[(size_t)k * p + j];
                // reject near-singular Gram matrices: the factor must reproduce a well-scaled diagonal
                if (!(d > 1e-6 * g[(size_t)j * p + j]) || !std::isfinite(d)) { ok = false; break; }
                const double ljj = std::sqrt(d);
                l[(size_t)j * p + j] = ljj;
                for (int i = j + 1; i < p; i++) {
                    double v = g[(size_t)i * p + j];
                    for (int k = 0; k < j; k++) v -= l[(size_t)k * p + i] * l[(size_t)k * p + j];
                    l[(size_t)j * p + i] = v / ljj;
                }
            }
            if (!ok) return householder(X, p);
            SNPGPU_HIP_CHECK(hipMemcpyAsync(gsmall.p, l.data(), sizeof(double) * l.size(), hipMemcpyHostToDevice, st));
            // rows: X <- L^-1 X, i.e. column-major Xc (n x p) <- Xc L^-T
            const double one = 1.0;
            if (rocblas_dtrsm(bl, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit,
                              (rocblas_int)n, p, &one, (const double *)gsmall.p, p, X, (rocblas_int)n) != rocblas_status_success)
                return fail("rocblas_dtrsm failed");
            if (sync()) return 1;       // `l` is reused by the second pass
        }
        return 0;
    }
    int randn(double *X, size_t count, uint32_t seed, uint64_t offset)
    {
        hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, X, count, seed, offset);
        SNPGPU_HIP_CHECK(hipGetLastError());
        return 0;
    }
    // eigen-decomposition of the symmetric m x m matrix T (device, overwritten by the eigenvectors, ascending values in W)
    int syevd(double *T, int m, double *W)
    {
        int lw = 0;
        if (hipsolverDnDsyevd_bufferSize(sv, HIPSOLVER_EIG_MODE_VECTOR, HIPSOLVER_FILL_MODE_LOWER, m, T, m, W, &lw) != HIPSOLVER_STATUS_SUCCESS)
            return fail("syevd workspace query failed");
        const size_t need = (size_t)std::max(lw, 1);
        if (swork.bytes < sizeof(double) * need) { if (sync()) return 1; swork.release(); if (swork.alloc(sizeof(double) * need)) return 1; }
        if (hipsolverDnDsyevd(sv, HIPSOLVER_EIG_MODE_VECTOR, HIPSOLVER_FILL_MODE_LOWER, m, T, m, W, (double *)swork.p, (int)need,
                              (int *)info.p) != HIPSOLVER_STATUS_SUCCESS)
            return fail("syevd failed");
        int hinfo = 0;
        SNPGPU_HIP_CHECK(hipMemcpyAsync(&hinfo, info.p, sizeof(int), hipMemcpyDeviceToHost, st));
        if (sync()) return 1;
        // message of src/genPCA.cpp:1333
        if (hinfo != 0) return fail("LAPACK::DSPEVX error (" + std::to_string(hinfo) + "), infinite or missing values in the genetic covariance matrix!");
        return 0;
    }
};

}  // namespace

int krylov_topk(EigOperator &op, int k, const snpgpu_eig_opts *user, double *eigval_host, double *eigvec, int mem,
                snpgpu_eig_info *info_out)
{
    const int64_t n = op.n();
    if (k <= 0 || k > n) { set_error("Invalid 'eigen.cnt'."); return 1; }
    snpgpu_eig_opts o{};
    if (user) o = *user;
    const double tol = o.tol > 0 ? o.tol : 1e-9;
    const int max_restarts = o.max_restarts > 0 ? o.max_restarts : 60;
    const uint32_t seed = o.seed ? o.seed : 20240601u;
    // the panel product works on 16-vector MFMA tiles (kernels_eig.hip): a block of k + 8 vectors costs as much as the next
    // multiple of 16, so take that (k = 32: 48 vectors per product, fewer products to converge)
    int b = o.block > 0 ? o.block : (int)std::min<int64_t>(n, (k + 8 + 15) / 16 * 16);
    b = (int)std::min<int64_t>(std::max(b, k), n);
    // blocks per restart cycle: long recurrences beat restarts (N = 100 000, flat noise spectrum, k = 32: 396 products at 12
    // blocks, 240 at 24, 216 at 36); 24 by default = 18 KB of basis + products per sample, halved while the device lacks the memory
    int depth = o.depth > 0 ? o.depth : 24;
    depth = (int)std::max<int64_t>(2, std::min<int64_t>(depth, std::max<int64_t>(2, n / b)));
    if ((int64_t)depth * b > n) depth = (int)std::max<int64_t>(1, n / b);
    if (depth < 2) { b = (int)n; depth = 1; }      // fewer than two blocks fit: one block that IS the space, exact at once

    Solver S(op);
    if (S.init()) return 1;
    const size_t bn = (size_t)b * (size_t)n;
    DevBuf basis, cw, r, ritz, cr, tmat, wvals, evsel;
    struct Free { std::vector<DevBuf *> v; ~Free() { for (DevBuf *d : v) d->release(); } } fr;
    fr.v = {&basis, &cw, &r, &ritz, &cr, &tmat, &wvals, &evsel};
    for (;;) {                                    // the two big blocks first; a shorter cycle if they do not fit
        size_t free_b = 0, total_b = 0;
        SNPGPU_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        const size_t need = sizeof(double) * bn * ((size_t)2 * depth + 2 + 2 * (size_t)std::max(1, depth / 3));
        // (never with a caller-side reduction: every rank of a multi-process run must take the same cycle length)
        if (need + ((size_t)1 << 30) <= free_b || depth <= 4 || o.depth > 0 || o.reduce) break;
        depth = std::max(4, depth / 2);
    }
    const int mmax = depth * b;
    // Thick restart: a cycle does not restart from its best b Ritz vectors alone but keeps `keep` blocks of them TOGETHER WITH
    // their products (C Y = W S: the solver stores W = C K for its whole basis, so they cost no product).  What a block method
    // converges against is the gap between the wanted eigenvalues and the first one NOT represented in what it carries along:
    // at the edge of a flat noise spectrum (spacing ~ j^(2/3)) 8 blocks instead of 1 widen that gap several times, while a cycle
    // still adds depth - keep new blocks.
    int keep = (int)env_i64("SNPGPU_EIG_KEEP", std::max(1, depth / 3), 1, 1 << 20);
    keep = depth >= 3 ? std::min(keep, depth - 2) : 1;
    const size_t kn = (size_t)keep * bn;
    if (basis.alloc(sizeof(double) * bn * (size_t)depth) || cw.alloc(sizeof(double) * bn * (size_t)depth) ||
        r.alloc(sizeof(double) * bn) || ritz.alloc(sizeof(double) * kn) || cr.alloc(sizeof(double) * kn) ||
        tmat.alloc(sizeof(double) * (size_t)mmax * mmax) || wvals.alloc(sizeof(double) * (size_t)mmax) ||
        evsel.alloc(sizeof(double) * (size_t)keep * b))
        return 1;
    double *K = (double *)basis.p, *W = (double *)cw.p, *R = (double *)r.p;
    uint64_t rnd_off = 0;
    if (S.randn(K, bn, seed, rnd_off)) return 1;
    rnd_off += bn;
    if (S.orth(K, b)) return 1;

    std::vector<double> nr, ev((size_t)mmax), theta((size_t)k);
    int restarts = 0, n_mm = 0, n_mm32 = 0;
    double rel = std::numeric_limits<double>::infinity();
    // Mixed precision: the fp64 panel product is bound by the fp64 matrix rate (and its atomics), the fp32 form takes half the
    // time with a relative error of ~5e-7 per product.  Three kinds of restart cycle:
    //   FP32   every product fp32.  Converges like an fp64 cycle while the residual is far above the products' error; run while
    //          the residual is above fp32_until and still falls by half per cycle.
    //   MIXED  the product of the cycle's FIRST block -- the Ritz vectors the cycle starts from -- in fp64, the other depth - 1
    //          in fp32.  A Ritz vector of the cycle is y = K_0 s_0 + sum_{j>0} K_j s_j with |s_j| ~ residual / gap for j > 0
    //          (the correction to a nearly converged vector), so the fp32 errors E_j enter the Ritz pair and its residual as
    //          sum_j E_j s_j ~ 5e-7 * residual / gap: far below the residual itself.  And the fp64 first product gives, at no
    //          extra product, the TRUE fp64 residuals of the vectors the previous cycle produced (Rayleigh-Ritz on (K_0, C K_0)
    //          alone): that check -- not the cycle's own estimate -- is what accepts them.
    //   FP64   every product fp64 (the round-2 solver): taken when fp32 products are off, or a mixed cycle stops converging.
    // Either way the returned pairs and `max_rel_residual` are statements about fp64 products.
    const double fp32_until = o.fp32_until > 0 ? o.fp32_until : 1e-5;
    enum Phase { P_FP32, P_MIXED, P_FP64 };
    const bool fp32_ok = !(o.fp32_until < 0) && env_i64("SNPGPU_EIG_FP32", 1, 0, 1) != 0 && tol < fp32_until && op.has_fp32_products();
    const bool mixed_ok = fp32_ok && env_i64("SNPGPU_EIG_MIXED", 1, 0, 1) != 0 && depth > 1;
    Phase phase = fp32_ok ? P_FP32 : P_FP64;
    bool accepted = false;
    int refused = 0;                              // estimates below tol that the fp64 check did not confirm
    // Rayleigh-Ritz on the first block alone: K_0 = K[0..b), W_0 = C K_0 (fp64): the k largest Ritz pairs of span(K_0) into
    // theta / ritz, their largest relative residual into rel_out
    auto verify = [&](double &rel_out) -> int {
        double *T = (double *)tmat.p;
        if (S.gram(K, b, W, b, T)) return 1;
        hipLaunchKernelGGL(symmetrize_kernel, dim3((unsigned)(((size_t)b * b + 255) / 256)), dim3(256), 0, S.st, T, b, b);
        if (S.syevd(T, b, (double *)wvals.p)) return 1;
        SNPGPU_HIP_CHECK(hipMemcpyAsync(ev.data(), wvals.p, sizeof(double) * (size_t)b, hipMemcpyDeviceToHost, S.st));
        if (S.sync()) return 1;
        std::vector<double> sel((size_t)k);
        for (int i = 0; i < k; i++) sel[(size_t)i] = ev[(size_t)(b - 1 - i)];
        if (S.small((size_t)b * k)) return 1;
        for (int i = 0; i < k; i++)
            SNPGPU_HIP_CHECK(hipMemcpyAsync((double *)S.gsmall.p + (size_t)i * b, T + (size_t)(b - 1 - i) * b, sizeof(double) * (size_t)b,
                                            hipMemcpyDeviceToDevice, S.st));
        if (S.combine((double *)ritz.p, k, (const double *)S.gsmall.p, b, K, b) ||
            S.combine((double *)cr.p, k, (const double *)S.gsmall.p, b, W, b))
            return 1;
        SNPGPU_HIP_CHECK(hipMemcpyAsync(evsel.p, sel.data(), sizeof(double) * (size_t)k, hipMemcpyHostToDevice, S.st));
        if (S.row_norm2((const double *)cr.p, (const double *)ritz.p, (const double *)evsel.p, k, nr)) return 1;
        rel_out = 0;
        for (int i = 0; i < k; i++) {
            theta[(size_t)i] = sel[(size_t)i];
            rel_out = std::max(rel_out, std::sqrt(nr[(size_t)i]) / std::max(std::fabs(sel[(size_t)i]), 1e-300));
        }
        return 0;
    };
    const bool verbose = getenv("SNPGPU_EIG_VERBOSE") != nullptr;
    auto now = [&]() { if (verbose) S.sync(); return std::chrono::steady_clock::now(); };       // (phase times: verbose runs only)
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    int kept = 0;                                 // blocks carried over from the previous cycle, with their products
    bool refresh = false;                         // recompute those products in fp64 (entering the fp64 phase)
    int stalled = 0;
    bool converged = false;                       // the returned pairs met `tol` by fp64 products
    for (int restart = 0; restart < max_restarts; restart++) {
        restarts = restart + 1;
        const Phase cycle = phase;
        double t_prod = 0, t_orth = 0, rel_true = -1;
        const auto t_cycle = now();
        const double rel_prev = rel;
        int nk = std::max(kept, 1);               // blocks in the basis; all but (in the first cycle) block 0 have their product
        auto product = [&](int j, bool lowp) -> int {
            if (S.sync()) return 1;
            const auto t0 = now();
            if (op.apply(K + (size_t)j * bn, b, W + (size_t)j * bn, lowp)) return 1;
            n_mm++; n_mm32 += lowp;
            t_prod += secs(t0, now());
            return 0;
        };
        const bool refreshed = refresh;
        if (refresh) {
            for (int j = 1; j < kept; j++)
                if (product(j, false)) return 1;
            refresh = false;
        }
        // the block whose product yields the next block: block 0 first (for kept Ritz vectors C Y_0 minus its part in the basis
        // IS their residual block), then always the newest block
        int src = 0;
        // block 0's product is at hand (C Y = W S), except that a mixed cycle forms it in fp64 -- its check -- and so does the
        // first fp64 cycle after cycles with fp32 products
        bool have = kept > 0 && (cycle == P_FP32 || (cycle == P_FP64 && !refreshed));
        for (;;) {
            if (!have) {
                const bool lowp = cycle == P_FP32 || (cycle == P_MIXED && src > 0);
                if (product(src, lowp)) return 1;
                if (src == 0 && cycle == P_MIXED && restart > 0) {
                    if (verify(rel_true)) return 1;
                    if (rel_true < tol) { accepted = true; rel = rel_true; break; }
                }
            }
            if (nk == depth) break;
            const auto t1 = now();
            SNPGPU_HIP_CHECK(hipMemcpyAsync(R, W + (size_t)src * bn, sizeof(double) * bn, hipMemcpyDeviceToDevice, S.st));
            for (int t = 0; t < 2; t++)           // full re-orthogonalisation, twice
                if (S.project_out(R, b, K, nk * b)) return 1;
            if (S.row_norm2(R, nullptr, nullptr, b, nr)) return 1;
            double nmax = 0;
            for (double v : nr) nmax = std::max(nmax, v);
            std::vector<double> wn;
            if (S.row_norm2(W + (size_t)src * bn, nullptr, nullptr, b, wn)) return 1;
            double wf = 0;
            for (double v : wn) wf += v;
            if (std::sqrt(nmax) < 1e-12 * std::max(1.0, std::sqrt(wf))) break;     // invariant subspace found
            // a (numerically) rank-deficient remainder makes QR return directions that are not orthogonal to the basis:
            // orthonormalise, look at what is left along the basis, and only while that is above rounding project it out and
            // orthonormalise again (round 2 did the three passes unconditionally: 0.19 -> 0.11 s per cycle at N = 150 000)
            double *Kn = K + (size_t)nk * bn;
            SNPGPU_HIP_CHECK(hipMemcpyAsync(Kn, R, sizeof(double) * bn, hipMemcpyDeviceToDevice, S.st));
            if (S.orth(Kn, b)) return 1;
            for (int round = 0; round < 2; round++) {
                double left = 0;
                if (S.project_out_if(Kn, b, K, nk * b, 1e-12, left)) return 1;
                if (left <= 1e-12) break;
                if (S.orth(Kn, b)) return 1;
            }
            src = nk++;
            have = false;
            t_orth += secs(t1, now());
        }
        if (accepted) {
            if (verbose)
                fprintf(stderr, "[snpgpu eigen] restart %d: fp64 product of the restart vectors: max relative residual %.3e: accepted\n",
                        restart + 1, rel);
            converged = true;
            break;
        }
        const auto t_rr = now();
        const int m = nk * b;                     // Rayleigh-Ritz on span(K[0..nk)): every block has its product
        double *T = (double *)tmat.p;
        if (S.gram(K, m, W, m, T)) return 1;       // T[i][j] = K_i . (C K_j)
        hipLaunchKernelGGL(symmetrize_kernel, dim3((unsigned)(((size_t)m * m + 255) / 256)), dim3(256), 0, S.st, T, m,
                           cycle == P_MIXED ? b : m);
        if (S.syevd(T, m, (double *)wvals.p)) return 1;
        SNPGPU_HIP_CHECK(hipMemcpyAsync(ev.data(), wvals.p, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, S.st));
        if (S.sync()) return 1;
        // Ritz vectors to form: the k wanted, a whole block of them for the plain restart, `keep` blocks for the thick one
        // (while that leaves at least two blocks of room for new directions)
        const int kk = m >= (keep + 2) * b ? keep * b : b;
        // the kk largest: columns m-1, m-2, ... of the eigenvector matrix; gather them in descending order
        std::vector<double> sel((size_t)kk);
        for (int i = 0; i < kk; i++) sel[(size_t)i] = ev[(size_t)(m - 1 - i)];
        // ritz[i] = sum_j S[j][m-1-i] K_j: reverse the column order by addressing column m-1-i one at a time would cost kk
        // GEMV calls; instead combine with the trailing kk columns (ascending) and swap rows afterwards on the small side:
        // build the reordered coefficient matrix on the device by kk column copies (m doubles each).
        if (S.small((size_t)m * kk)) return 1;
        for (int i = 0; i < kk; i++)
            SNPGPU_HIP_CHECK(hipMemcpyAsync((double *)S.gsmall.p + (size_t)i * m, T + (size_t)(m - 1 - i) * m, sizeof(double) * (size_t)m,
                                            hipMemcpyDeviceToDevice, S.st));
        if (S.combine((double *)ritz.p, kk, (const double *)S.gsmall.p, m, K, m) ||
            S.combine((double *)cr.p, kk, (const double *)S.gsmall.p, m, W, m))
            return 1;
        SNPGPU_HIP_CHECK(hipMemcpyAsync(evsel.p, sel.data(), sizeof(double) * (size_t)kk, hipMemcpyHostToDevice, S.st));
        if (S.row_norm2((const double *)cr.p, (const double *)ritz.p, (const double *)evsel.p, k, nr)) return 1;
        rel = 0;
        for (int i = 0; i < k; i++) {
            theta[(size_t)i] = sel[(size_t)i];
            rel = std::max(rel, std::sqrt(nr[(size_t)i]) / std::max(std::fabs(sel[(size_t)i]), 1e-300));
        }
        if (verbose)
            fprintf(stderr, "[snpgpu eigen] restart %d (%s products, %d blocks kept): %d products, basis %d x %lld, max relative residual %.3e; "
                    "products %.2f s, orthogonalisation %.2f s, Rayleigh-Ritz %.2f s\n",
                    restart + 1, cycle == P_FP32 ? "fp32" : cycle == P_MIXED ? "fp64 first, then fp32" : "fp64", kept, n_mm, m, (long long)n,
                    rel, t_prod, t_orth, secs(t_rr, now()));
        if (verbose && rel_true >= 0)
            fprintf(stderr, "[snpgpu eigen]     (the vectors this cycle started from: max relative residual %.3e by their fp64 product)\n", rel_true);
        (void)t_cycle;
        stalled = rel > 0.7 * rel_prev ? stalled + 1 : 0;
        if (cycle == P_FP32) {
            if (rel < fp32_until || stalled) { phase = mixed_ok ? P_MIXED : P_FP64; stalled = 0; }
        } else if (cycle == P_MIXED) {
            // (a cycle whose estimate is below tol is followed by one more first product: the check above)
            if (rel_true >= 0 && rel_prev < tol) refused++;
            if ((rel >= tol && stalled >= 2) || refused >= 2) phase = P_FP64;
        } else if (rel < tol) {
            converged = true;
            break;
        }
        if (restart + 1 == max_restarts) break;
        // thick restart: the Ritz vectors (orthonormal: K S with orthonormal K and S) and their products take the first kk / b
        // blocks of the next cycle (kk is a whole number of blocks: b >= k and the basis holds whole blocks)
        SNPGPU_HIP_CHECK(hipMemcpyAsync(K, ritz.p, sizeof(double) * (size_t)kk * (size_t)n, hipMemcpyDeviceToDevice, S.st));
        SNPGPU_HIP_CHECK(hipMemcpyAsync(W, cr.p, sizeof(double) * (size_t)kk * (size_t)n, hipMemcpyDeviceToDevice, S.st));
        kept = kk / b;
        refresh = phase == P_FP64 && cycle != P_FP64;     // the kept products carry fp32 errors: form them again
    }
    if (!converged) {
        // The cycle budget ran out.  What `rel` holds may be an estimate from fp32 products, and it did not meet the tolerance
        // either way: nothing unverified is returned as if it were exact.  One fp64 product of the Ritz vectors at hand and a
        // Rayleigh-Ritz step on their span give their TRUE residuals; below the tolerance they are accepted, otherwise the call
        // fails -- as LAPACK's dspevx reports INFO > 0 for eigenvectors that failed to converge (src/genPCA.cpp:1328-1335).
        SNPGPU_HIP_CHECK(hipMemcpyAsync(K, ritz.p, sizeof(double) * bn, hipMemcpyDeviceToDevice, S.st));
        if (S.sync()) return 1;
        if (op.apply(K, b, W, false)) return 1;
        n_mm++;
        double rel_true = rel;
        if (verify(rel_true)) return 1;
        rel = rel_true;
        if (verbose)
            fprintf(stderr, "[snpgpu eigen] %d restart cycles used up: fp64 check of the returned vectors: max relative residual %.3e (tolerance %.1e)\n",
                    restarts, rel, tol);
        if (!(rel < tol)) {
            if (info_out) { info_out->restarts = restarts; info_out->matmuls = n_mm; info_out->max_rel_residual = rel; info_out->block = b; info_out->depth = depth; info_out->matmuls_fp32 = n_mm32; info_out->reserved = 0; }
            char msg[320];
            snprintf(msg, sizeof(msg), "top-k eigen solver: not converged after %d restart cycles (%d products): max relative residual %.3e of the %d "
                     "wanted pairs against a tolerance of %.1e (n = %lld, block %d, %d blocks per cycle); raise max_restarts or the block size, "
                     "or use the dense solver (SNPGPU_EIG_DENSE_MAX)", restarts, n_mm, rel, k, tol, (long long)n, b, depth);
            set_error(msg);
            return 1;
        }
    }
    if (eigval_host) memcpy(eigval_host, theta.data(), sizeof(double) * (size_t)k);
    if (eigvec) {
        SNPGPU_HIP_CHECK(hipMemcpyAsync(eigvec, ritz.p, sizeof(double) * (size_t)k * (size_t)n,
                                        mem == SNPGPU_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, S.st));
        if (S.sync()) return 1;
    }
    if (info_out) { info_out->restarts = restarts; info_out->matmuls = n_mm; info_out->max_rel_residual = rel; info_out->block = b; info_out->depth = depth; info_out->matmuls_fp32 = n_mm32; info_out->reserved = 0; }
    return 0;
}

// ---- operators over row panels resident on ONE device --------------------------------------------------------------
int PanelsOperator::apply(const double *Q, int b, double *Y, bool fp32_products)
{
    SNPGPU_HIP_CHECK(hipSetDevice(dev_));
    double *dst = y_buf_ ? y_buf_ : Y;
    // (the panels' streams are non-blocking: nothing orders them against the NULL stream, so the buffer is cleared on one
    // of them and that one is waited for before any panel adds to it)
    SNPGPU_HIP_CHECK(hipMemsetAsync(dst, 0, sizeof(double) * (size_t)b * (size_t)n_, panels_[0]->stream));
    SNPGPU_HIP_CHECK(hipStreamSynchronize(panels_[0]->stream));
    // the one-pass kernel adds with atomics: the panels' streams may run side by side; the dgemm form (SNPGPU_EIG_BLAS=1)
    // reads and writes Y rows that panels share, one panel at a time
    const bool serial = getenv("SNPGPU_EIG_BLAS") != nullptr;
    for (snpgpu_ctx *c : panels_) {
        if (ctx_panel_matmul_enqueue(c, scale_, Q, b, dst, fp32_products && !serial)) return 1;
        if (serial) SNPGPU_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    for (snpgpu_ctx *c : panels_) SNPGPU_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (reduce_ && reduce_(user_)) { set_error("top-k eigen solver: the caller's reduction callback failed"); return 1; }
    if (dst != Y) {
        SNPGPU_HIP_CHECK(hipMemcpyAsync(Y, dst, sizeof(double) * (size_t)b * (size_t)n_, hipMemcpyDeviceToDevice, panels_[0]->stream));
        SNPGPU_HIP_CHECK(hipStreamSynchronize(panels_[0]->stream));
    }
    return 0;
}

}  // namespace snpgpu

using namespace snpgpu;

extern "C" {

int snpgpu_panels_topk_eigen(snpgpu_ctx *const *panels, int n_panels, double scale, int k, const snpgpu_eig_opts *opts,
                             double *eigval, double *eigvec, int mem, snpgpu_eig_info *info)
{
    if (!panels || n_panels <= 0 || !panels[0]) { set_error("snpgpu_panels_topk_eigen: no panels"); return 1; }
    const int dev = panels[0]->device;
    const int64_t n = panels[0]->N;
    std::vector<snpgpu_ctx *> v;
    for (int i = 0; i < n_panels; i++) {
        snpgpu_ctx *c = panels[i];
        if (!c || c->device != dev || c->N != n) { set_error("snpgpu_panels_topk_eigen: the panels must share one device and one sample count"); return 1; }
        if (!(c->kind == SNPGPU_PCA_COV || ((c->kind == SNPGPU_GRM_GCTA || c->kind == SNPGPU_EIGMIX) && c->frozen))) {
            set_error("snpgpu_panels_topk_eigen: needs PCA_COV contexts, or GRM_GCTA / EIGMIX contexts after snpgpu_finalize_inplace");
            return 1;
        }
        v.push_back(c);
    }
    PanelsOperator op(v, n, dev, scale, opts ? opts->y_buf : nullptr, opts ? opts->reduce : nullptr, opts ? opts->user : nullptr);
    std::vector<double> w((size_t)std::max(k, 1));
    const int rc = krylov_topk(op, k, opts, w.data(), eigvec, mem, info);
    // the fp32 copies of the panels (tens of GB at N = 1e5) served the solver's fp32 phase only: give the memory back, so that
    // later allocations of the session do not fail next to a buffer nobody reads (a second solve converts the panel again)
    for (snpgpu_ctx *c : v) { c->acc_f32.release(); c->acc_f32_valid = false; }
    if (rc) return 1;
    if (eigval) memcpy(eigval, w.data(), sizeof(double) * (size_t)k);      // always host memory
    return 0;
}

}  // extern "C"
