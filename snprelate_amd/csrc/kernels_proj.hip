// PCA projections: SNP correlations, SNP loadings, sample loadings (O(N L k), fp64 like the reference).
//   proj_snp_kernel<CORR>   CPCA_SNPCorr::thread_corr        src/genPCA.cpp:822-858
//   proj_snp_kernel<LOAD>   CPCA_SNPLoad::thread_loading     src/genPCA.cpp:950-998
//   proj_samp_kernel        CPCA_SampleLoad::thread_loading  src/genPCA.cpp:1046-1068
// One scheme for all three: a wave owns 64 OUTPUT rows (64 SNPs, or 64 samples), lane = row, and walks
// the reduction index (samples, or SNPs).  The operand row of a step (the eigenvector entries of that
// sample / the scaled loadings of that SNP) is wave-uniform -> scalar loads, SGPR operands of the fp64
// FMAs; the 64 two-bit codes of a step are 16 wave-uniform bytes (sample-major words w2 for the SNP
// kernels, the SNP-major packed rows for the sample kernel), each lane shifts out its own code.
#include "snpgpu_internal.h"

namespace snpgpu {

constexpr int PJ_KC_CORR = 8;     // output columns per wave pass of the SNP-side kernels (16 measured slower)
constexpr int PJ_KC = 16;         //   ... of the sample-side kernel

// code of lane l in a 16-byte group: dword l >> 4, bits 2 * (l & 15)
__device__ __forceinline__ uint32_t lane_code(uint32_t w, int sh) { return (w >> sh) & 3u; }

// ---- SNP-side kernels: lane = SNP, reduction over samples -------------------------------------------
// w2: [n_d][ncols_pad] (16 SNPs per dword, sample fastest); et: eigenvectors transposed [n_pad][kp]
// (rows >= N are zero); sum/num: per-SNP genotype sum / number of calls.  The sample range is split over
// blockIdx.z (a 64-SNP group alone is one wave: too few waves to hide the load latency); partial sums are
// added with fp64 atomics into zeroed buffers: LOAD -> out [n_snp][k]; CORR -> part [3][n_snp][k]
// (XY, X, XX) and cnt [n_snp][3] (#het, #hom2, #calls), turned into correlations by proj_corr_final_kernel.
template <int CORR, int KC>
__global__ __launch_bounds__(256) void proj_snp_kernel(const uint32_t *__restrict__ w2, int64_t ncols_pad, int64_t N,
                                                       int64_t n_snp, int64_t samp_per_block,
                                                       const double *__restrict__ et, int kp, int k,
                                                       const int32_t *__restrict__ sum, const int32_t *__restrict__ num,
                                                       int bayesian, double *__restrict__ out, int *__restrict__ cnt,
                                                       double *__restrict__ out_avg, double *__restrict__ out_scale,
                                                       const double *__restrict__ ext_avg,
                                                       const double *__restrict__ ext_scale)
{
    const int lane = threadIdx.x & 63;
    const int64_t grp = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // 64-SNP group
    if (grp * 64 >= n_snp) return;
    const int64_t snp = grp * 64 + lane;
    const int c0 = (int)blockIdx.y * KC;                                 // first output column of this pass
    const uint32_t *__restrict__ pw = w2 + (grp * 4 + (lane >> 4)) * ncols_pad;
    const int sh = 2 * (lane & 15);
    const int64_t n4 = (N + 3) & ~(int64_t)3;           // w2 holds code 3 for samples >= N
    const int64_t i_beg = (int64_t)blockIdx.z * samp_per_block;             // multiple of 8
    const int64_t i_end = (i_beg + samp_per_block < n4) ? (i_beg + samp_per_block) : n4;

    // per-SNP centring / scaling (PCA loadings) from the block statistics
    double y0 = 0, y1 = 0, y2 = 0;
    if (!CORR) {
        double avg = 0, scale = 0;
        if (snp < n_snp && ext_avg) {                   // caller's centring / scaling (EIGMIX: 2 p, 1/sqrt(sum 4p(1-p)))
            avg = ext_avg[snp]; scale = ext_scale[snp];
        } else if (snp < n_snp) {
            const int s = sum[snp], c = num[snp];
            if (c > 0) {
                avg = (double)s / c;
                if (!bayesian) {
                    const double p = avg * 0.5;
                    scale = (0.0 < p && p < 1.0) ? (1.0 / sqrt(p * (1.0 - p))) : 0.0;
                } else {
                    const double p = (double)(s + 1) / (2 * c + 2);
                    scale = 1.0 / sqrt(p * (1.0 - p));
                }
            }
            if (blockIdx.y == 0 && blockIdx.z == 0 && out_avg) { out_avg[snp] = avg; out_scale[snp] = scale; }
        }
        y0 = (0.0 - avg) * scale; y1 = (1.0 - avg) * scale; y2 = (2.0 - avg) * scale;
    }

    double a1[KC], a2[CORR ? KC : 1], a3[CORR ? KC : 1];
#pragma unroll
    for (int j = 0; j < KC; j++) a1[j] = 0;
    if (CORR) {
#pragma unroll
        for (int j = 0; j < KC; j++) { a2[j] = 0; a3[j] = 0; }
    }
    int n1 = 0, n2 = 0, nv = 0;

    for (int64_t i = i_beg; i < i_end; i += 4) {
        const uint4 cw = *reinterpret_cast<const uint4 *>(pw + i);          // 4 samples, 4 distinct dwords per wave
        const uint32_t w[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t code = lane_code(w[u], sh);
            const double *__restrict__ e = et + (i + u) * kp + c0;          // wave-uniform row
            if (CORR) {
                const bool valid = (code != 3u);
                const double b = valid ? 1.0 : 0.0;
                const double a = valid ? (double)code : 0.0;
                n1 += (code == 1u); n2 += (code == 2u); nv += valid;
#pragma unroll
                for (int j = 0; j < KC; j++) {
                    const double ev = e[j];
                    a1[j] = fma(a, ev, a1[j]);          // XY
                    a2[j] = fma(b, ev, a2[j]);          // X
                    a3[j] = fma(b * ev, ev, a3[j]);     // XX
                }
            } else {
                const double y = (code == 0u) ? y0 : (code == 1u) ? y1 : (code == 2u) ? y2 : 0.0;
#pragma unroll
                for (int j = 0; j < KC; j++) a1[j] = fma(y, e[j], a1[j]);
            }
        }
    }
    if (snp >= n_snp) return;
    const int64_t plane = n_snp * k;
#pragma unroll
    for (int j = 0; j < KC; j++) {
        if (c0 + j < k) {
            unsafeAtomicAdd(out + snp * k + c0 + j, a1[j]);
            if (CORR) {
                unsafeAtomicAdd(out + plane + snp * k + c0 + j, a2[j]);
                unsafeAtomicAdd(out + 2 * plane + snp * k + c0 + j, a3[j]);
            }
        }
    }
    if (CORR && blockIdx.y == 0) {
        atomicAdd(cnt + snp * 3, n1); atomicAdd(cnt + snp * 3 + 1, n2); atomicAdd(cnt + snp * 3 + 2, nv);
    }
}

// SNP_PC_Corr, src/genPCA.cpp:822-845, from the summed moments
__global__ __launch_bounds__(256) void proj_corr_final_kernel(const double *__restrict__ part, const int *__restrict__ cnt,
                                                              int64_t n_snp, int k, double *__restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_snp * k) return;
    const int64_t snp = idx / k, plane = n_snp * k;
    const int n1 = cnt[snp * 3], n2 = cnt[snp * 3 + 1], nv = cnt[snp * 3 + 2];
    double r = __builtin_nan("");
    if (nv > 1) {
        const double m = nv, Y = n1 + 2.0 * n2, YY = n1 + 4.0 * n2;
        const double xy = part[idx], x = part[plane + idx], xx = part[2 * plane + idx];
        const double v1 = xx - x * x / m, v2 = YY - Y * Y / m, val = v1 * v2;
        if (val > 0) r = (xy - x * Y / m) / sqrt(val);
    }
    out[idx] = r;
}

int launch_proj_snp(hipStream_t st, int corr, const uint32_t *w2, int64_t ncols_pad, int64_t N, int64_t n_snp,
                    const double *et, int kp, int k, const int32_t *sum, const int32_t *num, int bayesian, double *out,
                    double *part, int *cnt, double *out_avg, double *out_scale, const double *ext_avg,
                    const double *ext_scale)
{
    if (n_snp <= 0 || k <= 0) return 0;
    const int kc = PJ_KC_CORR;
    const int64_t gx = (n_snp + 255) / 256, gy = (k + kc - 1) / kc;
    // about 8 waves per SIMD: split the samples when SNP groups x column passes are few
    int64_t split = (8192 + gx * 4 * gy - 1) / (gx * 4 * gy);
    if (split < 1) split = 1;
    if (split > 64) split = 64;
    int64_t per = ((N + split - 1) / split + 7) & ~(int64_t)7;
    if (per < 256) per = 256;
    split = (N + per - 1) / per;
    dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)split);
    if (corr) {
        SNPGPU_HIP_CHECK(hipMemsetAsync(part, 0, sizeof(double) * 3 * (size_t)n_snp * (size_t)k, st));
        SNPGPU_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(int) * 3 * (size_t)n_snp, st));
        hipLaunchKernelGGL((proj_snp_kernel<1, PJ_KC_CORR>), grid, dim3(256), 0, st, w2, ncols_pad, N, n_snp, per, et, kp, k, sum, num,
                           bayesian, part, cnt, out_avg, out_scale, (const double *)nullptr, (const double *)nullptr);
        hipLaunchKernelGGL(proj_corr_final_kernel, dim3((unsigned)((n_snp * k + 255) / 256)), dim3(256), 0, st, part, cnt,
                           n_snp, k, out);
    } else {
        SNPGPU_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(double) * (size_t)n_snp * (size_t)k, st));
        hipLaunchKernelGGL((proj_snp_kernel<0, PJ_KC_CORR>), grid, dim3(256), 0, st, w2, ncols_pad, N, n_snp, per, et, kp, k, sum, num,
                           bayesian, out, cnt, out_avg, out_scale, ext_avg, ext_scale);
    }
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- sample-side kernel: lane = sample, reduction over the SNPs of one block ---------------------------
// packed: [n_snp][RB] SNP-major 2-bit rows; sl: scaled SNP loadings [n_snp][kp]; af / sc: per-SNP average
// and scale; out: [k][N] accumulated with fp64 atomics (blocks split the SNP range and feeds accumulate).
__global__ __launch_bounds__(256) void proj_samp_kernel(const uint8_t *__restrict__ packed, int64_t RB, int64_t N,
                                                        int64_t n_snp, int64_t snps_per_block,
                                                        const double *__restrict__ sl, int kp, int k,
                                                        const double *__restrict__ af, const double *__restrict__ sc,
                                                        double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t grp = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // 64-sample group
    if (grp * 64 >= N) return;
    const int64_t samp = grp * 64 + lane;
    const int c0 = (int)blockIdx.y * PJ_KC;
    const int64_t s_beg = (int64_t)blockIdx.z * snps_per_block;
    const int64_t s_end = (s_beg + snps_per_block < n_snp) ? (s_beg + snps_per_block) : n_snp;
    const uint32_t *__restrict__ pw = reinterpret_cast<const uint32_t *>(packed) + grp * 4 + (lane >> 4);
    const int64_t rs = RB >> 2;                          // dwords per SNP row
    const int sh = 2 * (lane & 15);
    double a1[PJ_KC];
#pragma unroll
    for (int j = 0; j < PJ_KC; j++) a1[j] = 0;
    int64_t s = s_beg;
    for (; s + 4 <= s_end; s += 4) {                     // four SNPs per iteration: their loads go out together
        uint32_t cw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) cw[u] = pw[(s + u) * rs];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t code = lane_code(cw[u], sh);
            const double f = af[s + u], c = sc[s + u];   // wave-uniform
            const double y = (code == 3u) ? 0.0 : ((double)code - f) * c;
            const double *__restrict__ h = sl + (s + u) * kp + c0;
#pragma unroll
            for (int j = 0; j < PJ_KC; j++) a1[j] = fma(y, h[j], a1[j]);
        }
    }
    for (; s < s_end; s++) {
        const uint32_t code = lane_code(pw[s * rs], sh);
        const double f = af[s], c = sc[s];
        const double y = (code == 3u) ? 0.0 : ((double)code - f) * c;
        const double *__restrict__ h = sl + s * kp + c0;
#pragma unroll
        for (int j = 0; j < PJ_KC; j++) a1[j] = fma(y, h[j], a1[j]);
    }
    if (samp >= N) return;
#pragma unroll
    for (int j = 0; j < PJ_KC; j++)
        if (c0 + j < k) unsafeAtomicAdd(out + (int64_t)(c0 + j) * N + samp, a1[j]);
}

int launch_proj_samp(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t N, int64_t n_snp, const double *sl,
                     int kp, int k, const double *af, const double *sc, double *out)
{
    if (n_snp <= 0 || k <= 0) return 0;
    const int64_t groups = (N + 63) / 64;
    // enough waves to fill the chip: split the SNP range when there are few sample groups
    int64_t split = (4096 + groups - 1) / groups;
    if (split < 1) split = 1;
    if (split > 64) split = 64;
    int64_t per = (n_snp + split - 1) / split;
    if (per < 64) per = 64;
    split = (n_snp + per - 1) / per;
    dim3 grid((unsigned)((groups + 3) / 4), (unsigned)((k + PJ_KC - 1) / PJ_KC), (unsigned)split);
    hipLaunchKernelGGL(proj_samp_kernel, grid, dim3(256), 0, st, packed, RB, N, n_snp, per, sl, kp, k, af, sc, out);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// eigenvectors [k][N] (R: N x k column-major) -> transposed, zero-padded [n_pad][kp]
__global__ __launch_bounds__(256) void proj_transpose_kernel(const double *__restrict__ src, int64_t N, int k,
                                                             double *__restrict__ dst, int64_t n_pad, int kp)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pad) return;
    for (int j = 0; j < kp; j++) dst[i * kp + j] = (i < N && j < k) ? src[(int64_t)j * N + i] : 0.0;
}

int launch_proj_transpose(hipStream_t st, const double *src, int64_t N, int k, double *dst, int64_t n_pad, int kp)
{
    hipLaunchKernelGGL(proj_transpose_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, st, src, N, k, dst,
                       n_pad, kp);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace snpgpu
