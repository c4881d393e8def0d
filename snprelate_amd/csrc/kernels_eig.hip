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
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int EG_STRIP = 64;     // panel rows per workgroup
constexpr int EG_VT = 3;         // 16-vector tiles per launch (48 vectors)

// The fp64 product: v_mfma_f64_16x16x4_f64 (47.5 TFLOP/s sustained, tools/ubench/mfma_f64_rate.hip).  The same lane maps as the
// fp32 instruction for A and B; D: lane holds column lc, rows 4 * r + lk (measured: tools/ubench/mfma_f64_layout.hip; the
// fp32 instruction's rows are 4 * lk + r).
__global__ __launch_bounds__(256, 2) void sym_panel_matmul_kernel(const double *__restrict__ P, int64_t ld, int64_t tiles_c,
                                                                  int64_t nI, int64_t nJ, int64_t col0, int64_t N, double scale,
                                                                  const double *__restrict__ Q, int m,
                                                                  double *__restrict__ Y, const double *__restrict__ Qt)
{
    typedef f64x4 acc4;
    __shared__ double sQ[EG_STRIP][EG_VT * 16 + 2];           // Q[v][I] of this strip, [i][v] (+2: bank spread)
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

    acc4 d1[4][EG_VT];
#pragma unroll
    for (int it = 0; it < 4; it++)
#pragma unroll
        for (int vt = 0; vt < EG_VT; vt++) d1[it][vt] = (acc4){0, 0, 0, 0};
    // D1[v][i]: lane holds row-sample i = lc of tile it, vectors v = drow(r, lk)
    auto flush_d1 = [&]() {
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
                        if (v < m) unsafeAtomicAdd(Y + (int64_t)v * N + gi, scale * (double)d1[it][vt][r]);
                    }
            }
        }
    };

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
        // A operands of product (1): Q[v][J] from the sample-major copy Qt[j][v] (16 consecutive values per lane group)
        double qj[EG_VT][4];
#pragma unroll
        for (int vt = 0; vt < EG_VT; vt++)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int64_t gj = col0 + j0 + 4 * s + lk;            // (columns N .. next multiple of 16: zero rows of Qt,
                qj[vt][s] = Qt[gj * (EG_VT * 16) + 16 * vt + lc];      //  whatever the panel's padding holds); vectors >= m are zero in Qt
            }
#pragma unroll
        for (int it = 0; it < 4; it++)
#pragma unroll
            for (int vt = 0; vt < EG_VT; vt++)
#pragma unroll
                for (int s = 0; s < 4; s++) d1[it][vt] = __builtin_amdgcn_mfma_f64_16x16x4f64(qj[vt][s], ts[it][s], d1[it][vt], 0, 0, 0);
        if (both) {
            acc4 d2[EG_VT];
#pragma unroll
            for (int vt = 0; vt < EG_VT; vt++) d2[vt] = (acc4){0, 0, 0, 0};
#pragma unroll
            for (int it = 0; it < 4; it++)
#pragma unroll
                for (int s = 0; s < 4; s++) {
#pragma unroll
                    for (int vt = 0; vt < EG_VT; vt++)
                        d2[vt] = __builtin_amdgcn_mfma_f64_16x16x4f64(sQ[16 * it + 4 * s + lk][16 * vt + lc], tb[it][s], d2[vt], 0, 0, 0);
                }
            // D2[v][j]: lane holds column j = lc, rows v = drow(r, lk)
            const int64_t gj = col0 + j0 + lc;
            if (gj < N) {
#pragma unroll
                for (int vt = 0; vt < EG_VT; vt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int v = 16 * vt + 4 * r + lk;
                        if (v < m) unsafeAtomicAdd(Y + (int64_t)v * N + gj, scale * (double)d2[vt][r]);
                    }
            }
        }
    }
    flush_d1();
}

// Qt[j][v] = Q[v][j] for the (up to) 48 vectors of one launch, j < N rounded up to 16; vectors >= m and rows >= N are zero
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
        if (j0 + j < (N + 15) / 16 * 16) Qt[(j0 + j) * (EG_VT * 16) + v] = t[v][j];      // rows N .. next multiple of 16: zeros
    }
}

// ---- the fp32 form, arranged for its own bounds --------------------------------------------------------------------------
// Measured at N = 100 000, 48 vectors (HISTORY.md 4.6, 8): the generic kernel above in fp32 takes 27.5 ms
// against 22.4 ms in fp64 -- with the matrix time halved its loads bind (48 of 8 bytes per lane and column tile, 16 of them
// gathering 32-byte pieces of 16 rows, all waited for in front of the products); with the loads restructured it takes the
// same 22.4 ms, and 10.5 ms with the fp64 atomics of product (2) left out: 768 atomics per 64 x 16 tile, 3.6e9 per product,
// are the bound of BOTH precisions once the matrix instructions are out of the way.  This kernel therefore
//   * gives a workgroup 256 panel rows (64 per wave) x a chunk of EG_CHUNK columns; the four waves walk the chunk's column
//     tiles together, their product-(2) partial sums are added through LDS and leave as ONE set of atomics per 256 x 16 tile
//     (a quarter of the atomics); product (1) stays in the wave's registers for the whole chunk;
//   * reads the tile once, in the row-contiguous arrangement (4 full 128-byte rows per instruction), one column tile ahead
//     of the products, and makes the other arrangement by a transposition through wave-private LDS;
//   * reads the vectors as 16-byte loads from a copy laid out in the products' own lane order (eig_qt4_kernel): those of
//     product (1) one tile ahead, those of product (2) -- the wave's own 64 rows -- once, into registers.
constexpr int EG_TP = 17;          // LDS pitch of the transposition tile (floats)
constexpr int EG_ROWS = 256;       // panel rows per workgroup
constexpr int EG_CHUNK = 1024;     // columns per workgroup = the longest fp32 sum of product (1) before it goes to fp64
                                   // (512 .. 8192: 12.2, 11.7, 11.6, 11.9, 12.5 ms; rms error 3.3, 4.6, 6.4, 9.0, 12.6 e-7)

// TP = double: the panel's fp64 sums are converted on the way in; TP = float (round 4): an fp32 COPY of the finalised panel
// (panel_to_f32_kernel, made once where the device has the memory) is streamed instead -- half the bytes, no conversions.
template <typename TP>
__global__ __launch_bounds__(256, 2) void sym_panel_matmul_f32_kernel(const TP *__restrict__ P, int64_t ld, int64_t tiles_c,
                                                                      int64_t nI, int64_t nJ, int64_t col0, int64_t N,
                                                                      double scale, int m, double *__restrict__ Y,
                                                                      const f32x4 *__restrict__ Qt4, int chunk_tiles)
{
    __shared__ float sT[4][EG_STRIP][EG_TP];
    __shared__ float red[2][4][EG_VT * 4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i0 = (int64_t)blockIdx.y * EG_ROWS;     // panel-relative first row of the workgroup (= its diagonal column)
    const int64_t n_jb = (nJ + 15) / 16;
    const int64_t jb_begin = max((int64_t)blockIdx.x * chunk_tiles, i0 / 16);
    const int64_t jb_end = min((int64_t)(blockIdx.x + 1) * chunk_tiles, n_jb);
    if (jb_begin >= jb_end) return;                        // the chunk lies left of the diagonal (uniform over the workgroup)
    const int lc = lane & 15, lk = lane >> 4;
    const int64_t rs = tiles_c ? ACC_TILE : ld;
    const int64_t iw = i0 + 64 * wave;                     // this wave's rows
    const bool rows_live = iw < nI;
    const int64_t gb0 = col0 / 16;                         // col0 is a multiple of 16 (checked by the launcher)

    // A operands of product (2): Q[v][i] for the wave's rows, aq[it][vt][s] = Q[16 vt + lc][iw + 16 it + 4 s + lk]
    f32x4 aq[4][EG_VT];
#pragma unroll
    for (int it = 0; it < 4; it++)
#pragma unroll
        for (int vt = 0; vt < EG_VT; vt++) {
            const int64_t gb = gb0 + iw / 16 + it;
            f32x4 q = (f32x4){0, 0, 0, 0};
            if (rows_live && gb * 16 < N) q = Qt4[(gb * EG_VT + vt) * 64 + lane];
#pragma unroll
            for (int s = 0; s < 4; s++)
                if (iw + 16 * it + 4 * s + lk >= nI) q[s] = 0.0f;      // rows of the next panel: not this one's to add
            aq[it][vt] = q;
        }
    f32x4 d1[4][EG_VT];
#pragma unroll
    for (int it = 0; it < 4; it++)
#pragma unroll
        for (int vt = 0; vt < EG_VT; vt++) d1[it][vt] = (f32x4){0, 0, 0, 0};

    TP nx[4][4];
    auto fetch = [&](int64_t jb) {
        const TP *__restrict__ pt = P + acc_off(ld, tiles_c, iw, jb * 16);
#pragma unroll
        for (int it = 0; it < 4; it++)
#pragma unroll
            for (int s = 0; s < 4; s++) nx[it][s] = pt[(16 * it + 4 * s + lk) * rs + lc];
    };
    const int64_t jb_mine = iw / 16;                       // the wave's first column tile: its mirrored diagonal tile
    if (rows_live && jb_begin >= jb_mine) fetch(jb_begin);
    int buf = 0;
    for (int64_t jb = jb_begin; jb < jb_end; jb++, buf ^= 1) {
        const int64_t j0 = jb * 16;
        f32x4 d2[EG_VT];
#pragma unroll
        for (int vt = 0; vt < EG_VT; vt++) d2[vt] = (f32x4){0, 0, 0, 0};
        if (rows_live && jb >= jb_mine) {                  // wave-uniform
            float tb[4][4];
            f32x4 qj[EG_VT];
#pragma unroll
            for (int it = 0; it < 4; it++)
#pragma unroll
                for (int s = 0; s < 4; s++) tb[it][s] = (float)nx[it][s];
            // in flight under product (2): the vectors of product (1) and the next tile
#pragma unroll
            for (int vt = 0; vt < EG_VT; vt++) qj[vt] = Qt4[((gb0 + jb) * EG_VT + vt) * 64 + lane];
            if (jb + 1 < jb_end) fetch(jb + 1);
            // transposition: tb holds T[16 it + 4 s + lk][lc], product (1) needs T[16 it + lc][4 s + lk]
#pragma unroll
            for (int it = 0; it < 4; it++)
#pragma unroll
                for (int s = 0; s < 4; s++) sT[wave][16 * it + 4 * s + lk][lc] = tb[it][s];
            if (j0 >= iw + EG_STRIP) {                     // right of the wave's (mirrored) diagonal tile
#pragma unroll
                for (int it = 0; it < 4; it++)
#pragma unroll
                    for (int s = 0; s < 4; s++)
#pragma unroll
                        for (int vt = 0; vt < EG_VT; vt++)
                            d2[vt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[it][vt][s], tb[it][s], d2[vt], 0, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < 4; it++) {
                float ts[4];
#pragma unroll
                for (int s = 0; s < 4; s++) ts[s] = sT[wave][16 * it + lc][4 * s + lk];
#pragma unroll
                for (int vt = 0; vt < EG_VT; vt++)
#pragma unroll
                    for (int s = 0; s < 4; s++)
                        d1[it][vt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qj[vt][s], ts[s], d1[it][vt], 0, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else if (rows_live && jb + 1 == jb_mine && jb + 1 < jb_end)
            fetch(jb + 1);
        // D2[v][j]: lane holds column j = lc, rows v = 16 vt + 4 lk + r; the four waves' partial sums meet in LDS
#pragma unroll
        for (int vt = 0; vt < EG_VT; vt++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[buf][wave][4 * vt + r][lane] = d2[vt][r];
        __syncthreads();
        if (j0 >= i0 + EG_STRIP) {                         // otherwise no wave had anything to add
            const int64_t gj = col0 + j0 + lc;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int e = 3 * wave + c;                // (vt, r) = (e / 4, e % 4)
                const float sum = (red[buf][0][e][lane] + red[buf][1][e][lane]) + (red[buf][2][e][lane] + red[buf][3][e][lane]);
                const int v = 16 * (e >> 2) + 4 * lk + (e & 3);
                if (gj < N && v < m) unsafeAtomicAdd(Y + (int64_t)v * N + gj, scale * (double)sum);
            }
        }
    }
    // D1[v][i]: lane holds row-sample i = lc of tile it, vectors v = 16 vt + 4 lk + r
    if (rows_live) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int64_t ri = iw + 16 * it + lc;
            const int64_t gi = col0 + ri;
            if (ri < nI && gi < N) {
#pragma unroll
                for (int vt = 0; vt < EG_VT; vt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int v = 16 * vt + 4 * lk + r;
                        if (v < m) unsafeAtomicAdd(Y + (int64_t)v * N + gi, scale * (double)d1[it][vt][r]);
                    }
            }
        }
    }
}

// the vectors in the lane order of product (1): Qt4[(gb * 3 + vt) * 64 + lane] = { Q[16 vt + lc][16 gb + 4 s + lk], s = 0..3 },
// zero for vectors >= m and columns >= N; gb < ceil(N / 16)
__global__ __launch_bounds__(256) void eig_qt4_kernel(const double *__restrict__ Q, int m, int64_t N, f32x4 *__restrict__ Qt4)
{
    const int64_t gb = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gb * 16 >= N) return;
    const int lane = threadIdx.x & 63, lc = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int vt = 0; vt < EG_VT; vt++) {
        const int v = 16 * vt + lc;
        f32x4 o;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int64_t g = gb * 16 + 4 * s + lk;
            o[s] = (v < m && g < N) ? (float)Q[(int64_t)v * N + g] : 0.0f;
        }
        Qt4[(gb * EG_VT + vt) * 64 + lane] = o;
    }
}

// element-wise fp32 copy of a panel plane (same offsets: tile-major or row-major alike)
__global__ __launch_bounds__(256) void panel_to_f32_kernel(const double *__restrict__ src, float *__restrict__ dst, size_t n4)
{
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256) {
        const double4 v = reinterpret_cast<const double4 *>(src)[e];
        reinterpret_cast<float4 *>(dst)[e] = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
    }
}

int launch_panel_to_f32(hipStream_t st, const double *src, float *dst, size_t n_elems)
{
    if (n_elems == 0) return 0;
    hipLaunchKernelGGL(panel_to_f32_kernel, dim3(8192), dim3(256), 0, st, src, dst, n_elems / 4);     // planes are multiples of 256 x 256
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// P: panel accumulator [rows_pad][ld] with its diagonal square mirrored; nI = panel rows, nJ = N - col0 columns;
// qt_scratch: 48 * (N + 16) doubles; fp32_products: the fp32 form of the product (sym_panel_matmul_f32_kernel; col0 is a
// multiple of 256 for every panel, snpgpu_create); P32 != nullptr: the fp32 copy of the panel for that form
int launch_sym_panel_matmul(hipStream_t st, const double *P, int64_t ld, int64_t tiles_c, int64_t nI, int64_t nJ, int64_t col0,
                            int64_t N, double scale, const double *Q, int m, double *Y, double *qt_scratch, bool fp32_products,
                            const float *P32)
{
    if (nI <= 0 || m <= 0) return 0;
    for (int v0 = 0; v0 < m; v0 += EG_VT * 16) {
        const int mc = (m - v0 < EG_VT * 16) ? (m - v0) : EG_VT * 16;
        if (fp32_products) {
            const int64_t n_gb = (N + 15) / 16;
            hipLaunchKernelGGL(eig_qt4_kernel, dim3((unsigned)((n_gb + 3) / 4)), dim3(256), 0, st, Q + (int64_t)v0 * N, mc, N,
                               (f32x4 *)qt_scratch);
            const int chunk = EG_CHUNK;
            const dim3 grid((unsigned)((nJ + chunk - 1) / chunk), (unsigned)((nI + EG_ROWS - 1) / EG_ROWS));
            if (P32)
                hipLaunchKernelGGL(sym_panel_matmul_f32_kernel<float>, grid, dim3(256), 0, st, P32, ld, tiles_c, nI, nJ, col0, N, scale, mc,
                                   Y + (int64_t)v0 * N, (const f32x4 *)qt_scratch, chunk / 16);
            else
                hipLaunchKernelGGL(sym_panel_matmul_f32_kernel<double>, grid, dim3(256), 0, st, P, ld, tiles_c, nI, nJ, col0, N, scale, mc,
                                   Y + (int64_t)v0 * N, (const f32x4 *)qt_scratch, chunk / 16);
        } else {
            hipLaunchKernelGGL(eig_qt_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, st, Q + (int64_t)v0 * N, mc, N, qt_scratch);
            hipLaunchKernelGGL(sym_panel_matmul_kernel, dim3((unsigned)((nI + EG_STRIP - 1) / EG_STRIP)), dim3(256), 0, st, P, ld, tiles_c,
                               nI, nJ, col0, N, scale, Q + (int64_t)v0 * N, mc, Y + (int64_t)v0 * N, (const double *)qt_scratch);
        }
    }
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace snpgpu
