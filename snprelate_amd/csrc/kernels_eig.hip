// Y += scale * C Q for the symmetric fp64 covariance held as a row panel (building block of the top-k eigen
// solver, snprelate_amd/eigen.py; replaces two rocBLAS dgemm calls that read the panel twice and run a
// 40-column product at 0.9 TB/s).  One pass: every tile T = P[I, J] strictly right of the diagonal tile is used
// for BOTH triangles,
//     Y[I] += T   Q[J]      and      Y[J] += T^T Q[I],
// the mirrored diagonal tile for the first product only.  fp64 MFMA (v_mfma_f64_16x16x4_f64), up to 48 vectors
// per launch.  Operand roles are chosen so that both products take T as the B operand:
//     (1) D1[v][i] = sum_j Q[v][j] T[i][j]     B[k=j][c=i]: lane (i = l&15, j = l>>4)   strided 32-byte reads
//     (2) D2[v][j] = sum_i Q[v][i] T[i][j]     B[k=i][c=j]: lane (j = l&15, i = l>>4)   coalesced 128-byte rows
// (the second read of T hits L1/L2).  Q and Y are vector-major [m][N] (what torch hands over).
#include "snpgpu_internal.h"

namespace snpgpu {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int EG_STRIP = 64;     // panel rows per workgroup
constexpr int EG_VT = 3;         // 16-vector tiles per launch (48 vectors)

__global__ __launch_bounds__(256, 2) void sym_panel_matmul_kernel(const double *__restrict__ P, int64_t ld, int64_t tiles_c,
                                                                  int64_t nI, int64_t nJ, int64_t col0, int64_t N, double scale,
                                                                  const double *__restrict__ Q, int m,
                                                                  double *__restrict__ Y, const double *__restrict__ Qt)
{
    __shared__ double sQ[EG_STRIP][EG_VT * 16 + 2];      // Q[v][I] of this strip, [i][v] (+2: bank spread)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * EG_STRIP;   // panel-relative first row (= relative column of the diagonal)
    if (i0 >= nI) return;
    const int lc = lane & 15, lk = lane >> 4;
    const int64_t rs = tiles_c ? ACC_TILE : ld;          // row stride inside a block

    for (int e = tid; e < EG_STRIP * EG_VT * 16; e += 256) {
        const int i = e % EG_STRIP, v = e / EG_STRIP;
        const int64_t gi = col0 + i0 + i;
        sQ[i][v] = (v < m && gi < N) ? Q[(int64_t)v * N + gi] : 0.0;
    }
    __syncthreads();

    f64x4 d1[4][EG_VT];
#pragma unroll
    for (int it = 0; it < 4; it++)
#pragma unroll
        for (int vt = 0; vt < EG_VT; vt++) d1[it][vt] = (f64x4){0, 0, 0, 0};

    const int64_t n_jb = (nJ + 15) / 16;
    for (int64_t jb = i0 / 16 + wave; jb < n_jb; jb += 4) {
        const int64_t j0 = jb * 16;
        const bool both = (j0 >= i0 + EG_STRIP);          // right of the (mirrored) diagonal tile
        // T in the two operand arrangements (a 64 x 16 block never crosses a 256 x 256 tile of a tile-major panel)
        double ts[4][4], tb[4][4];
        const double *__restrict__ pt = P + acc_off(ld, tiles_c, i0, j0);
        // the row-contiguous arrangement first: its 128-byte rows are what travels from HBM, the strided 32-byte reads of the
        // other arrangement then hit L1 / L2
#pragma unroll
        for (int it = 0; it < 4; it++)
#pragma unroll
            for (int s = 0; s < 4; s++) tb[it][s] = pt[(16 * it + 4 * s + lk) * rs + lc];
#pragma unroll
        for (int it = 0; it < 4; it++)
#pragma unroll
            for (int s = 0; s < 4; s++) ts[it][s] = pt[(16 * it + lc) * rs + 4 * s + lk];
        // A operands of product (1): Q[v][J] from the sample-major copy Qt[j][v] (16 consecutive doubles per lane group)
        double qj[EG_VT][4];
#pragma unroll
        for (int vt = 0; vt < EG_VT; vt++)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                int64_t gj = col0 + j0 + 4 * s + lk;
                gj = gj < N ? gj : N - 1;                 // columns >= N hold zeros in P: any finite value will do
                qj[vt][s] = Qt[gj * (EG_VT * 16) + 16 * vt + lc];      // vectors >= m are zero in Qt
            }
#pragma unroll
        for (int it = 0; it < 4; it++)
#pragma unroll
            for (int vt = 0; vt < EG_VT; vt++)
#pragma unroll
                for (int s = 0; s < 4; s++)
                    d1[it][vt] = __builtin_amdgcn_mfma_f64_16x16x4f64(qj[vt][s], ts[it][s], d1[it][vt], 0, 0, 0);
        if (both) {
            f64x4 d2[EG_VT];
#pragma unroll
            for (int vt = 0; vt < EG_VT; vt++) d2[vt] = (f64x4){0, 0, 0, 0};
#pragma unroll
            for (int it = 0; it < 4; it++)
#pragma unroll
                for (int s = 0; s < 4; s++) {
#pragma unroll
                    for (int vt = 0; vt < EG_VT; vt++)
                        d2[vt] = __builtin_amdgcn_mfma_f64_16x16x4f64(sQ[16 * it + 4 * s + lk][16 * vt + lc], tb[it][s],
                                                                      d2[vt], 0, 0, 0);
                }
            // D2[v][j]: lane holds column j = lc, rows v = 4 * r + lk (measured: tools/ubench/mfma_f64_layout.hip)
            const int64_t gj = col0 + j0 + lc;
            if (gj < N) {
#pragma unroll
                for (int vt = 0; vt < EG_VT; vt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int v = 16 * vt + 4 * r + lk;
                        if (v < m) unsafeAtomicAdd(Y + (int64_t)v * N + gj, scale * d2[vt][r]);
                    }
            }
        }
    }
    // D1[v][i]: lane holds row-sample i = lc of tile it, vectors v = 4 * r + lk
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int64_t ri = i0 + 16 * it + lc;
        const int64_t gi = col0 + ri;
        if (ri < nI && gi < N) {
#pragma unroll
            for (int vt = 0; vt < EG_VT; vt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int v = 16 * vt + 4 * r + lk;
                    if (v < m) unsafeAtomicAdd(Y + (int64_t)v * N + gi, scale * d1[it][vt][r]);
                }
        }
    }
}

// Qt[j][v] = Q[v][j] for the (up to) 48 vectors of one launch; vectors >= m are zero
__global__ __launch_bounds__(256) void eig_qt_kernel(const double *__restrict__ Q, int m, int64_t N, double *__restrict__ Qt)
{
    __shared__ double t[EG_VT * 16][65];
    const int64_t j0 = (int64_t)blockIdx.x * 64;
    for (int e = threadIdx.x; e < EG_VT * 16 * 64; e += 256) {
        const int v = e / 64, j = e % 64;
        t[v][j] = (v < m && j0 + j < N) ? Q[(int64_t)v * N + j0 + j] : 0.0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < EG_VT * 16 * 64; e += 256) {
        const int j = e / (EG_VT * 16), v = e % (EG_VT * 16);
        if (j0 + j < N) Qt[(j0 + j) * (EG_VT * 16) + v] = t[v][j];
    }
}

// P: panel accumulator [rows_pad][ld] with its diagonal square mirrored; nI = panel rows, nJ = N - col0 columns
int launch_sym_panel_matmul(hipStream_t st, const double *P, int64_t ld, int64_t tiles_c, int64_t nI, int64_t nJ, int64_t col0,
                            int64_t N, double scale, const double *Q, int m, double *Y, double *qt_scratch)
{
    if (nI <= 0 || m <= 0) return 0;
    for (int v0 = 0; v0 < m; v0 += EG_VT * 16) {
        const int mc = (m - v0 < EG_VT * 16) ? (m - v0) : EG_VT * 16;
        hipLaunchKernelGGL(eig_qt_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, st, Q + (int64_t)v0 * N, mc, N, qt_scratch);
        hipLaunchKernelGGL(sym_panel_matmul_kernel, dim3((unsigned)((nI + EG_STRIP - 1) / EG_STRIP)), dim3(256), 0, st, P, ld,
                           tiles_c, nI, nJ, col0, N, scale, Q + (int64_t)v0 * N, mc, Y + (int64_t)v0 * N, qt_scratch);
    }
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace snpgpu
