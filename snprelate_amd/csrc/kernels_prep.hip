// Pre-pass kernels over one feed block (all HBM-bound, O(N*B) bytes):
//   repack      caller block (uint8 or 2-bit rows)  -> aligned 2-bit rows, pad samples = missing
//   snp_stats   per-SNP genotype sum / non-missing count   (vec_u8_geno_count, src/dVect.cpp:30-117)
//   build_lut   per-SNP decode table for the SYRK kernel   (DivideGeno/rsqrt_prod, src/genPCA.cpp:98-181)
//   bitplanes   SNP-major 2-bit codes -> sample-major bit planes via wave ballots
//               (the role of PackSNPGeno1b, src/dGenGWAS.cpp:1429-1475, with a 4-plane encoding)
#include "snpgpu_internal.h"

namespace snpgpu {

// ---------------------------------------------------------------------------
// repack: one thread produces one output byte (4 samples).
// format U8: src[snp*N + samp];  PACKED2: src[snp*ceil(N/4) + samp/4]
__global__ __launch_bounds__(256) void repack_kernel(const uint8_t *__restrict__ src, int format,
                                                     int64_t n_snp, int64_t N, uint8_t *__restrict__ dst,
                                                     int64_t RB)
{
    const int64_t snp = blockIdx.y;
    const int64_t rb_in = (N + 3) >> 2;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < RB; b += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s0 = b * 4;
        unsigned out;
        if (s0 >= N) {
            out = 0xFFu;
        } else if (format == SNPGPU_GENO_U8) {
            out = 0;
            const uint8_t *p = src + snp * N + s0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                unsigned g = (s0 + k < N) ? p[k] : 3u;
                g = g > 3u ? 3u : g;  // vec_u8_geno_valid, src/dGenGWAS.cpp:1388
                out |= g << (2 * k);
            }
        } else {
            out = src[snp * rb_in + b];
            const int rem = (int)(N - s0);
            if (rem < 4) out |= (0xFFu << (2 * rem)) & 0xFFu;
        }
        dst[snp * RB + b] = (uint8_t)out;
    }
}

int launch_repack(hipStream_t st, const void *src, int format, int64_t n_snp, int64_t n_samp,
                  uint8_t *packed, int64_t RB)
{
    if (n_snp <= 0) return 0;
    int gx = (int)((RB + 255) / 256);
    if (gx > 64) gx = 64;
    dim3 grid(gx, (unsigned)n_snp);
    hipLaunchKernelGGL(repack_kernel, grid, dim3(256), 0, st, (const uint8_t *)src, format, n_snp, n_samp,
                       packed, RB);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// snp_stats: one workgroup per SNP; 16-byte loads (64 samples), popcount on the code bits.
//   code bits (hi,lo): 0=(0,0) 1=(0,1) 2=(1,0) 3=(1,1)
__device__ __forceinline__ void count_word(uint32_t w, int &n1, int &n2, int &nm)
{
    const uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
    n1 += __popc(lo & ~hi);
    n2 += __popc(hi & ~lo);
    nm += __popc(lo & hi);
}

__global__ __launch_bounds__(256) void snp_stats_kernel(const uint8_t *__restrict__ packed, int64_t RB,
                                                        int64_t N, int32_t *__restrict__ sum,
                                                        int32_t *__restrict__ num,
                                                        unsigned long long *__restrict__ d_missing,
                                                        int32_t *__restrict__ nhet)
{
    const int64_t snp = blockIdx.x;
    const uint4 *row = reinterpret_cast<const uint4 *>(packed + snp * RB);
    const int nvec = (int)(RB >> 4);
    int n1 = 0, n2 = 0, nm = 0;
    for (int v = threadIdx.x; v < nvec; v += 256) {
        const uint4 q = row[v];
        count_word(q.x, n1, n2, nm);
        count_word(q.y, n1, n2, nm);
        count_word(q.z, n1, n2, nm);
        count_word(q.w, n1, n2, nm);
    }
    // wave reduce then LDS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n1 += __shfl_down(n1, off);
        n2 += __shfl_down(n2, off);
        nm += __shfl_down(nm, off);
    }
    __shared__ int red[3][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = n1; red[1][wave] = n2; red[2][wave] = nm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        n1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        n2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        nm = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        const int npad = (int)(RB * 4 - N);  // padding samples are stored as missing
        const int miss = nm - npad;
        sum[snp] = n1 + 2 * n2;
        num[snp] = (int)N - miss;
        if (nhet) nhet[snp] = n1;   // #(g == 1): AB count of GetABNumPerSNP (src/dGenGWAS.cpp:314-360)
        // only ever tested against zero ("does this block hold missing calls"): a plain store, not 16 384
        // same-address atomics (measured 199 us per block at 5 % missing)
        if (miss > 0) *d_missing = 1ull;
    }
}

int launch_snp_stats(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t n_samp,
                     int32_t *sum, int32_t *num, unsigned long long *d_missing_cells, int32_t *nhet)
{
    if (n_snp <= 0) return 0;
    hipLaunchKernelGGL(snp_stats_kernel, dim3((unsigned)n_snp), dim3(256), 0, st, packed, RB, n_samp, sum, num,
                       d_missing_cells, nhet);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// repack + snp_stats in one pass over the caller's block (feed path): one workgroup per SNP, every
// thread turns 16 samples into one aligned output dword (16-byte / 4-byte loads when the row is
// aligned, byte loads otherwise) and counts on the produced code bits.
__global__ __launch_bounds__(256) void repack_stats_kernel(const uint8_t *__restrict__ src, int format, int64_t N,
                                                           uint8_t *__restrict__ dst, int64_t RB,
                                                           int32_t *__restrict__ sum, int32_t *__restrict__ num,
                                                           unsigned long long *__restrict__ d_missing)
{
    const int64_t snp = blockIdx.x;
    const int64_t rb_in = (N + 3) >> 2;
    const int n_dw = (int)(RB >> 2);                 // output dwords of this SNP (RB is a multiple of 64)
    uint32_t *__restrict__ out_row = reinterpret_cast<uint32_t *>(dst + snp * RB);
    int n1 = 0, n2 = 0, nm = 0;
    if (format == SNPGPU_GENO_U8) {
        const uint8_t *__restrict__ row = src + snp * N;
        const bool aligned = ((reinterpret_cast<uintptr_t>(row) & 15u) == 0);
        for (int d = threadIdx.x; d < n_dw; d += 256) {
            const int64_t s0 = (int64_t)d * 16;
            uint32_t out;
            if (s0 + 16 <= N) {
                uint32_t q[4];
                if (aligned) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(row + s0);
                    q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        q[k] = (uint32_t)row[s0 + 4 * k] | ((uint32_t)row[s0 + 4 * k + 1] << 8) |
                               ((uint32_t)row[s0 + 4 * k + 2] << 16) | ((uint32_t)row[s0 + 4 * k + 3] << 24);
                }
                out = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // bytes > 3 are missing (vec_u8_geno_valid, src/dGenGWAS.cpp:1388): any of bits 2..7 set -> 3
                    uint32_t x = q[k];
                    const uint32_t t = x | (x >> 1);          // bits 2, 4, 6 of t = (2|3), (4|5), (6|7) of x
                    const uint32_t big = ((t >> 2) | (t >> 4) | (t >> 6)) & 0x01010101u;
                    const uint32_t bigx = big * 3u;
                    x = (x & 0x03030303u) | bigx;
                    // gather the four 2-bit codes of this dword into one byte
                    const uint32_t b = (x | (x >> 6) | (x >> 12) | (x >> 18)) & 0xFFu;
                    out |= b << (8 * k);
                }
            } else {
                out = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    unsigned g = (s0 + k < N) ? row[s0 + k] : 3u;
                    g = g > 3u ? 3u : g;
                    out |= g << (2 * k);
                }
            }
            out_row[d] = out;
            count_word(out, n1, n2, nm);
        }
    } else {
        const uint8_t *__restrict__ row = src + snp * rb_in;
        const bool aligned = ((reinterpret_cast<uintptr_t>(row) & 3u) == 0);      // dword loads where the row allows them
        for (int d = threadIdx.x; d < n_dw; d += 256) {
            const int64_t b0 = (int64_t)d * 4;
            if (aligned && b0 * 4 + 16 <= N) {
                const uint32_t out = *reinterpret_cast<const uint32_t *>(row + b0);
                out_row[d] = out;
                count_word(out, n1, n2, nm);
                continue;
            }
            uint32_t out = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int64_t b = b0 + k, s0 = b * 4;
                uint32_t v = 0xFFu;
                if (s0 < N) {
                    v = row[b];
                    const int rem = (int)(N - s0);
                    if (rem < 4) v |= (0xFFu << (2 * rem)) & 0xFFu;
                }
                out |= v << (8 * k);
            }
            out_row[d] = out;
            count_word(out, n1, n2, nm);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n1 += __shfl_down(n1, off);
        n2 += __shfl_down(n2, off);
        nm += __shfl_down(nm, off);
    }
    __shared__ int red[3][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = n1; red[1][wave] = n2; red[2][wave] = nm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        n1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        n2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        nm = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        const int miss = nm - (int)(RB * 4 - N);      // padding samples are stored as missing
        sum[snp] = n1 + 2 * n2;
        num[snp] = (int)N - miss;
        if (miss > 0) *d_missing = 1ull;              // only ever tested against zero
    }
}

int launch_repack_stats(hipStream_t st, const void *src, int format, int64_t n_snp, int64_t n_samp, uint8_t *packed,
                        int64_t RB, int32_t *sum, int32_t *num, unsigned long long *d_missing)
{
    if (n_snp <= 0) return 0;
    hipLaunchKernelGGL(repack_stats_kernel, dim3((unsigned)n_snp), dim3(256), 0, st, (const uint8_t *)src, format,
                       n_samp, packed, RB, sum, num, d_missing);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// build_lut: per-SNP values {z(0), z(1), z(2), z(missing)} (missing is 0 except for the EIGMIX weight
// table), stored as a per-SNP-PAIR table for the SYRK kernel.
// Arithmetic in fp64 like the reference, each entry rounded once to fp32.
__global__ __launch_bounds__(256) void build_lut_kernel(const int32_t *__restrict__ sum,
                                                        const int32_t *__restrict__ num, int64_t n_snp,
                                                        int64_t n_snp_pad, int mode, int split16,
                                                        float2 *__restrict__ lut,
                                                        unsigned long long *__restrict__ d_nlocus,
                                                        double *__restrict__ d_sumden, double *__restrict__ dvals,
                                                        const unsigned long long *__restrict__ d_missing,
                                                        double2 *__restrict__ ccoef, int exact_rows_always, int w_shift,
                                                        int exact_with_missing, int entry12, double *__restrict__ homo_const,
                                                        double4 *__restrict__ uvsp_miss, int x1_sparse_mac,
                                                        unsigned long long *__restrict__ d_short_runs)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;   // n_snp_pad is a multiple of 64: whole waves
    if (k >= n_snp_pad) return;
    double x = 0, y = 0, wmiss = 0, dden = 0, avg = 0, wtrue = 0;
    bool poly = false;
    int mac = 1 << 30, minor_is_counted = 1;
    if (k < n_snp) {
        const int s = sum[k], c = num[k];
        mac = (s < 2 * c - s) ? s : (2 * c - s);
        minor_is_counted = (s <= c);
        avg = (c > 0) ? ((double)s / c) : 0.0;               // DivideGeno, genPCA.cpp:98-142
        poly = (0 < s) && (s < 2 * c);                        // genPCA.cpp:1206
        if (mode == LUT_GCTA) {
            const double p = avg * 0.5;                       // rsqrt_prod, genPCA.cpp:145-181
            const double sc = (0 < p && p < 1) ? (1.0 / sqrt(p * (1 - p))) : 0.0;
            y = sc; x = -avg * sc;
        } else if (mode == LUT_BAYES) {
            const double p = (s + 1.0) / (2.0 * c + 2.0);     // genPCA.cpp:441-453
            const double sc = 1.0 / sqrt(p * (1 - p));
            y = sc; x = -avg * sc;
        } else if (mode == LUT_EIGMIX_NUM || mode == LUT_EIGMIX_MISSW) {
            const double af = 0.5 * avg;                      // genEIGMIX.cpp:116-121
            dden = 4 * af * (1 - af);
            if (mode == LUT_EIGMIX_NUM) { x = -avg; y = 1.0; }
            else wmiss = exact_rows_always ? dden : sqrt(dden);   // m_i * [d m_j]  |  [sqrt(d) m_i] * [sqrt(d) m_j]
        } else {
            const double p = (c > 0) ? (0.5 * s / c) : 0.0;   // genKING.cpp:236-248
            const double w = p * (1 - p);
            wtrue = (mode == LUT_HOMO_W1) ? w : w * w;
            if (exact_rows_always) {                          // v_i * [c v_j]: the whole weight (and scale) on the column side
                x = ldexp((mode == LUT_HOMO_W1) ? w : w * w, 2 * H3_HOMO_SHIFT);
            } else {
                x = (mode == LUT_HOMO_W1) ? sqrt(w) : w;
                if (split16) x = ldexp(x, H3_HOMO_SHIFT);    // keep p(1-p) ~ 1e-6 in fp16's normal range
            }
            y = 0;
        }
    }
    // pair table: SNPs (2p, 2p+1) share 16 float2 entries indexed by c0 + 4*c1 -> (z_2p(c0), z_2p+1(c1)),
    // so that the SYRK kernel decodes TWO operand values with one table read (ds_read_b64).
    // The even lane writes entries 0..7, the odd lane 8..15 (n_snp_pad is even, lanes pair up).
    const bool odd = (k & 1);
    if (split16) {
        // fp16 pair hi = fp16(z), lo = fp16(z - hi) (22 significant bits); entry = {hi0 | hi1 << 16, lo0 | lo1 << 16}
        double zd[4] = {x, x + y, x + 2.0 * y, wmiss};
        // Exact-row-side SYRK (syrk_h3_kernel<2, true>): 16-byte entries {hi pair, lo pair, row pair, row pair}.
        // Column operand w = y z 2^-w_shift (0 for a missing call); row operand (g - cs) 2^w_shift with the centre
        // cs = avg rounded to the fewest binary digits that keep (avg - cs)^2 <= Var(g)/64, so that the products have the
        // variance of the centred form at any allele frequency; the column term (avg - cs) w(g) = u + v g is summed per
        // chunk by colcorr_kernel and subtracted from every row at the flush.
        // A MISSING row call must contribute 0 = a w - (avg - cs) w, i.e. its row value is a = avg - cs: a real number,
        // kept as fp16(avg - cs).  In a block with missing calls cs therefore takes all 9 fractional digits an exact
        // fp16 (g - cs) allows, |avg - cs| <= 2^-10, and the rounding of a is <= 2^-21 (2^-25 absolute in the fp16
        // subnormal range) per missing cell: below the lo parts' own 2^-22 |w|.
        const bool has_missing = (*d_missing != 0ull);
        const bool exact_rows = ccoef && (exact_with_missing || !has_missing);
        // Rare variants in a block WITH missing calls (uvsp_miss; GCTA / Bayesian weights y^2 = 1 / (p (1 - p)) up to ~N): a
        // pair of carriers would put y^2 ~ 1e4 .. 1e5 into an fp32 accumulator whose other terms are O(1), and every later
        // addition of the run is then rounded at that magnitude (measured: 1.3e-5 off-diagonal figure on a rare-variant
        // spectrum with 2 % missing calls; 7.8e-6 on a flat one).  Such an SNP stays in the dense product with every CALLED
        // genotype replaced by the non-carrier's (the tables below: all three codes get the non-carrier's value), i.e. it
        // contributes y^2 avg'^2 m_i m_j exactly as before for pairs of non-carriers, and uv_sparse_kernel adds what the
        // carriers' pairs lack in fp64.
        // (weights below X1_SPARSE_MIN_W stay where they are: nothing large enters the accumulators, and the fp64 atomics of
        // the sparse path -- whose order is not fixed -- stay out of small data sets, where two runs are expected to agree bit
        // for bit)
        const bool rare = uvsp_miss && has_missing && exact_rows && y * y >= X1_SPARSE_MIN_W && (mode == LUT_GCTA || mode == LUT_BAYES) &&
                          mac <= x1_sparse_mac;
        const double g_nc = minor_is_counted ? 0.0 : 2.0;         // the non-carrier's genotype
        if (uvsp_miss && has_missing)
            uvsp_miss[k] = rare ? make_double4(y * y, minor_is_counted ? avg : 2.0 - avg, minor_is_counted ? 0.0 : 1.0, 1.0)
                                : make_double4(0, 0, 0, 0);
        if (rare) zd[0] = zd[1] = zd[2] = x + g_nc * y;
        // such a block runs as 4096-SNP fp32 runs (syrk_x1_kernel reads the flag): what the carriers leave in the dense product is
        // small, but the spectrum that holds them is the thinnest accuracy case (9.3e-6 with 8192-SNP runs, 5.9e-6 with 4096)
        if (rare && d_short_runs) *d_short_runs = 1ull;
        double cs = 1.0;
        if (ccoef) {
            if (exact_rows) {
                const double var = 0.5 * avg * (2.0 - avg);
                for (int kb = has_missing ? 9 : 0; kb <= 9; kb++) {
                    cs = ldexp(rint(ldexp(avg, kb)), -kb);
                    if ((avg - cs) * (avg - cs) * 64.0 <= var) break;
                }
                for (int c = 0; c < 3; c++) zd[c] = ldexp(zd[c] * y, -w_shift);
            }
            ccoef[k] = !exact_rows ? make_double2(0.0, 0.0)
                       : rare ? make_double2((avg - cs) * y * (x + g_nc * y), 0.0)      // w(g) = u + v g is the same for every call
                              : make_double2((avg - cs) * y * x, (avg - cs) * y * y);
        }
        uint32_t hl[4], ho[4], ar[4], ao[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const _Float16 hi = (_Float16)zd[c];    // each value on its own fp16 grid: 22 bits of THAT value in hi + lo
            const _Float16 lo = (_Float16)(zd[c] - (double)hi);
            hl[c] = (uint32_t)__builtin_bit_cast(uint16_t, hi) | ((uint32_t)__builtin_bit_cast(uint16_t, lo) << 16);
            // exact for c < 3; c == 3 (missing call, SNP / sample padding): the centre residual, see above
            const _Float16 a = (y != 0.0) ? (_Float16)ldexp((c < 3 ? (rare ? g_nc : (double)c) : avg) - cs, w_shift) : (_Float16)0.0;
            ar[c] = (uint32_t)__builtin_bit_cast(uint16_t, a);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) { ho[c] = (uint32_t)__shfl_xor((int)hl[c], 1); ao[c] = (uint32_t)__shfl_xor((int)ar[c], 1); }
        if (exact_rows && entry12) {
            // syrk_x1_kernel: 12-byte entries {hi pair, lo pair, row pair}; dword banks 3 c + {0, 1, 2} (mod 32) are distinct
            // for the 16 entries of a pair, so plain ds_read_b32 lookups are conflict-free and land in place
            uint32_t *dst = reinterpret_cast<uint32_t *>(lut) + ((k >> 1) * 16 + (odd ? 8 : 0)) * 3;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int idx = e + (odd ? 8 : 0), c0 = idx & 3, c1 = idx >> 2;
                const uint32_t a = odd ? ho[c0] : hl[c0], b = odd ? hl[c1] : ho[c1];   // SNP 2p, SNP 2p+1
                const uint32_t ra = odd ? ao[c0] : ar[c0], rb = odd ? ar[c1] : ao[c1];
                dst[3 * e] = (a & 0xFFFFu) | (b << 16);
                dst[3 * e + 1] = (a >> 16) | (b & 0xFFFF0000u);
                dst[3 * e + 2] = ra | (rb << 16);
            }
        } else if (exact_rows) {
            uint4 *dst = reinterpret_cast<uint4 *>(lut) + (k >> 1) * 16 + (odd ? 8 : 0);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int idx = e + (odd ? 8 : 0), c0 = idx & 3, c1 = idx >> 2;
                const uint32_t a = odd ? ho[c0] : hl[c0], b = odd ? hl[c1] : ho[c1];   // SNP 2p, SNP 2p+1
                const uint32_t ra = odd ? ao[c0] : ar[c0], rb = odd ? ar[c1] : ao[c1];
                // the row pair twice: the two lane halves of the kernel read different copies (LDS banks)
                dst[e] = make_uint4((a & 0xFFFFu) | (b << 16), (a >> 16) | (b & 0xFFFF0000u), ra | (rb << 16), ra | (rb << 16));
            }
        } else {
            uint2 *dst = reinterpret_cast<uint2 *>(lut) + (k >> 1) * 16 + (odd ? 8 : 0);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int idx = e + (odd ? 8 : 0), c0 = idx & 3, c1 = idx >> 2;
                const uint32_t a = odd ? ho[c0] : hl[c0], b = odd ? hl[c1] : ho[c1];   // SNP 2p, SNP 2p+1
                dst[e] = make_uint2((a & 0xFFFFu) | (b << 16), (a >> 16) | (b & 0xFFFF0000u));
            }
        }
    } else {
        const float z[4] = {(float)x, (float)(x + y), (float)(x + 2.0 * y), (float)wmiss};
        float zo[4];
#pragma unroll
        for (int c = 0; c < 4; c++) zo[c] = __shfl_xor(z[c], 1);
        float2 *dst = lut + (k >> 1) * 16 + (odd ? 8 : 0);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int idx = e + (odd ? 8 : 0), c0 = idx & 3, c1 = idx >> 2;
            dst[e] = odd ? make_float2(zo[c0], z[c1]) : make_float2(z[c0], zo[c1]);
        }
    }
    if (homo_const && *d_missing == 0ull) {   // KING-homo: in a block without missing calls every pair gets the whole sum
        double v = wtrue;                     // (the masked SYRK of this table is skipped for such a block)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0 && v != 0.0) unsafeAtomicAdd(homo_const, v);
    }
    if (dvals) { dvals[2 * k] = dden; dvals[2 * k + 1] = -x; }   // {4p(1-p), avg} in fp64 for the per-sample sums
    if (d_sumden) {            // SumDenominator of CEigMix_AlgArith::Run, one fp64 atomic per wave
        double v = dden;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0 && v != 0.0) unsafeAtomicAdd(d_sumden, v);
    }
    if (d_nlocus) {
        const unsigned long long b = __ballot(poly);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(d_nlocus, (unsigned long long)__popcll(b));
    }
}

int launch_build_lut(hipStream_t st, const int32_t *sum, const int32_t *num, int64_t n_snp, int64_t n_snp_pad,
                     int lut_mode, int split16, float2 *lut, unsigned long long *d_nlocus, double *d_sumden,
                     double *dvals, const unsigned long long *d_missing, double2 *ccoef, int exact_rows_always, int w_shift,
                     int exact_with_missing, int entry12, double *homo_const, double4 *uvsp_miss, int x1_sparse_mac,
                     unsigned long long *d_short_runs)
{
    if (n_snp_pad <= 0) return 0;
    hipLaunchKernelGGL(build_lut_kernel, dim3((unsigned)((n_snp_pad + 255) / 256)), dim3(256), 0, st, sum, num,
                       n_snp, n_snp_pad, lut_mode, split16, lut, d_nlocus, d_sumden, dvals, d_missing, ccoef,
                       exact_rows_always, w_shift, exact_with_missing, entry12, homo_const, uvsp_miss, x1_sparse_mac, d_short_runs);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// Column term of the exact-row-side SYRK: T[j] += sum over the block's SNPs of (avg_s - c_s) w_s(g_js) = u_s + v_s g_js
// (fp64; g_js from the pair-coded words W8, byte = 16 * (c0 + 4 * c1)).  Cells with code 3 (missing calls, SNP / sample
// padding) have w = 0 and contribute nothing.  One thread per column walks the block in SNP order, so the sum does not
// depend on the launch geometry.  always == 0: only for blocks without missing calls (the others take the three-product
// kernel).  The term is the same for every row of the panel: it is subtracted once, by colterm_settle_kernel.
__global__ __launch_bounds__(256) void colcorr_kernel(const uint32_t *__restrict__ w8, int64_t ncols_pad, int n_d,
                                                      const double2 *__restrict__ ccoef, double *__restrict__ tc,
                                                      const unsigned long long *__restrict__ d_missing, int always,
                                                      int entry12)
{
    // always 0: blocks without missing calls only; 1: every block; 2: blocks WITH missing calls only (the others take
    // the single-product kernel and uvcorr_kernel)
    if (always == 2 ? (*d_missing == 0ull) : (!always && *d_missing != 0ull)) return;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= ncols_pad) return;
    const int d0 = blockIdx.y * (H3_LUTCH / 16);          // chunks of H3_LUTCH / 2 SNPs
    const int d1 = (d0 + H3_LUTCH / 16 < n_d) ? (d0 + H3_LUTCH / 16) : n_d;
    double s = 0.0;
    for (int d = d0; d < d1; d++) {
        const uint32_t w = w8[(int64_t)d * ncols_pad + col];
        const double2 *__restrict__ cf = ccoef + (int64_t)d * 8;     // wave-uniform: scalar loads
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t by = (w >> (8 * p)) & 0xFFu;
            const uint32_t b = entry12 ? (by * 171u) >> 11 : by >> 4, c0 = b & 3u, c1 = b >> 2;   // bytes carry 12 / 16 * code here
            const double2 f0 = cf[2 * p], f1 = cf[2 * p + 1];
            s += (c0 == 3u) ? 0.0 : (f0.x + f0.y * (double)c0);
            s += (c1 == 3u) ? 0.0 : (f1.x + f1.y * (double)c1);
        }
    }
    tc[(int64_t)blockIdx.y * ncols_pad + col] = s;
}

// colterm[j] += the chunk sums of this block, in chunk order
__global__ __launch_bounds__(256) void colterm_add_kernel(const double *__restrict__ tc, int n_chunk, int64_t ncols_pad,
                                                          double *__restrict__ colterm,
                                                          const unsigned long long *__restrict__ d_missing, int always)
{
    if (always == 2 ? (*d_missing == 0ull) : (!always && *d_missing != 0ull)) return;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= ncols_pad) return;
    double s = colterm[col];
    for (int k = 0; k < n_chunk; k++) s += tc[(int64_t)k * ncols_pad + col];
    colterm[col] = s;
}

int launch_colcorr(hipStream_t st, const uint32_t *w8, int64_t ncols_pad, int n_d, const double2 *ccoef, double *tc,
                   double *colterm, const unsigned long long *d_missing, int always, int entry12)
{
    if (n_d <= 0) return 0;
    const int n_chunk = (n_d + H3_LUTCH / 16 - 1) / (H3_LUTCH / 16);
    dim3 grid((unsigned)((ncols_pad + 255) / 256), (unsigned)n_chunk);
    hipLaunchKernelGGL(colcorr_kernel, grid, dim3(256), 0, st, w8, ncols_pad, n_d, ccoef, tc, d_missing, always, entry12);
    hipLaunchKernelGGL(colterm_add_kernel, dim3(grid.x), dim3(256), 0, st, tc, n_chunk, ncols_pad, colterm, d_missing, always);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// Tables of the single-product SYRK (syrk_uv_kernel; blocks without missing calls).  Per SNP:
//   * the weight t = y^2 = 1 / (p (1 - p)) as a product of two fp16 numbers: u runs over the 1024 mantissas of its octave
//     (u ~ sqrt(t)), v = fp16(t / u); the pair with the smallest |u v - t| is kept.  One product of two 11-bit mantissas
//     reaches a given weight only to ~1e-6 rms (6.6e-6 at worst: the candidates' errors are a Poisson process of density
//     ~1 / 1.4e-6, the best one Laplace-distributed) -- round 3 bought that down with a second slot for a quarter of the SNPs
//     (1.25 x the MFMA work).  Round 4: WEIGHT TARGETS PER fp32 RUN.  The kernel runs a block as R launches ("runs", one fp64
//     flush each) and a run's flush may multiply its fp32 sums by a constant for free.  Run q therefore carries the factor
//     f_q = 1 - q / 4096 (exact in 13 bits: f_q x an fp32 partial is exact in fp64) and a SNP placed in run q needs
//     u v ~ t / f_q: R different targets per SNP, R x 1024 candidates, and SNP order inside a feed block is free (the sum is
//     order-independent).  uv_factor_kernel finds the best pair for every target, uv_assign_kernel deals the SNPs to the runs
//     (each SNP to its best target while the run has room -- deterministic, in SNP order; the ~1 % that overflow take their
//     next best), uv_tables_kernel builds the tables in slot order.  The factorisation error falls as 1 / R: 1.05e-6 rms for one
//     target, 0.37e-6 for three, 0.285e-6 for four (numpy emulation and tools/panel_error_distribution.py), with NO extra slots;
//     u v f_q IS the SNP's weight from then on (row / column / constant terms);
//   * integer centres c_a (rows), c_b (columns): one lane per 64-slot chunk walks its slots in order and keeps the running
//     mean of the products, cum = sum d_a d_b u v, near zero: (near, near) adds d^2 u v >= 0, (near, other neighbour) adds
//     d_near d_far u v <= 0 and is taken when it brings cum closer to zero -- but only for SNPs where it costs at most a
//     factor 6 in the variance of the products, (Var g + d_a^2)(Var g + d_b^2) <= 6 (Var g)^2: avg within ~0.3 of x.5.  A
//     far centre on a RARE variant would put +-u v ~ 1/p into every column of a carrier's row (cancelled later by the row
//     term, but carried through the fp32 sums); rare variants keep (near, near), whose products are sparse and whose
//     mean d^2 u v ~ 2 avg is small, and lean on the common SNPs of the chunk to cancel it;
//   * pair table entry c0 + 4 c1 = {(c0 - c_a) u | (c1 - c_a') u' << 16, (c0 - c_b) v | (c1 - c_b') v' << 16}: exact fp16
//     values, 0 for code 3 (SNP / sample padding);
//   * uvcoef = {d_b u v f, c_a, d_a u v f, c_b} for the row / column terms, kpart[chunk] = sum d_a d_b u v f.
// The K dimension of such a block is a list of SLOTS: slot_src maps slots to the block's SNPs for the transposition (-1: an
// empty slot; SNPs without weight -- monomorphic, rare variants on the fp64 path, padding -- own none).  One run (or a kind
// whose weight is exact: EIGMIX) = one target, slot k = SNP k, no map.
struct UvSnp { double t, avg; };
__device__ __forceinline__ UvSnp uv_snp_weight(const int32_t *__restrict__ sum, const int32_t *__restrict__ num, int64_t k,
                                               int64_t n_snp, int mode, bool *sparse)
{
    UvSnp r{0.0, 0.0};
    *sparse = false;
    if (k >= n_snp) return r;
    const int s = sum[k], c = num[k];
    r.avg = (c > 0) ? ((double)s / c) : 0.0;
    if (mode == LUT_GCTA) {
        const double p = r.avg * 0.5;
        r.t = (0 < p && p < 1) ? (1.0 / (p * (1 - p))) : 0.0;
    } else if (mode == LUT_EIGMIX_NUM) {
        r.t = 1.0;                                            // (g_i - 2p)(g_j - 2p): u = v = 1, no factorisation error
    } else {                                                  // LUT_BAYES
        const double p = (s + 1.0) / (2.0 * c + 2.0);
        r.t = 1.0 / (p * (1 - p));
    }
    // rare variants (<= UV_SPARSE_MAC copies of the minor allele) leave the dense product: uv_sparse_kernel adds their
    // few carrier pairs and their row / column terms in fp64 with the exact weight
    if (r.t > 0) {
        const int mac = (s < 2 * c - s) ? s : (2 * c - s);
        *sparse = (mac <= UV_SPARSE_MAC);
    }
    return r;
}

// one wave per SNP: the lanes share out the 1024 mantissas of u for each of the n_target targets t / f_q
__global__ __launch_bounds__(256) void uv_factor_kernel(const int32_t *__restrict__ sum, const int32_t *__restrict__ num,
                                                        int64_t n_snp, int64_t n_snp_pad, int mode, int n_target,
                                                        float *__restrict__ cand_err, uint32_t *__restrict__ cand_uv,
                                                        double2 *__restrict__ snp_tavg, double4 *__restrict__ uvsp,
                                                        const unsigned long long *__restrict__ d_missing)
{
    if (*d_missing != 0ull) return;
    const int lane = threadIdx.x & 63;
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n_snp_pad) return;
    bool sparse;
    UvSnp w = uv_snp_weight(sum, num, k, n_snp, mode, &sparse);
    if (lane == 0) {
        uvsp[k] = sparse ? make_double4(w.t, (sum[k] <= num[k]) ? w.avg : 2.0 - w.avg, (sum[k] <= num[k]) ? 0.0 : 1.0, 1.0)
                         : make_double4(0, 0, 0, 0);
        snp_tavg[k] = make_double2(sparse ? 0.0 : w.t, w.avg);
    }
    if (sparse) w.t = 0;
    for (int q = 0; q < n_target; q++) {
        float rel = 0.f;
        uint32_t uv = 0;
        if (w.t > 0) {                                        // wave-uniform
            const double tt = w.t / uv_run_factor(q);
            const int e = ilogb(sqrt(tt));
            const float tf = (float)tt;
            double best = 1e300;
            int bm = 0;
            _Float16 bv = (_Float16)0.0;
#pragma unroll 4
            for (int i = 0; i < 16; i++) {
                const int m = lane * 16 + i;
                const double uc = ldexp(1.0 + (double)m * (1.0 / 1024.0), e);
                const _Float16 vh = (_Float16)(tf / (float)uc);             // any fp16 near the quotient: judged by the product
                const double err = fabs(uc * (double)vh - tt);
                if (err < best) { best = err; bm = m; bv = vh; }
            }
            for (int o = 32; o; o >>= 1) {                    // arg-min over the wave; ties to the smaller mantissa
                const double oe = __shfl_xor(best, o);
                const int om = __shfl_xor(bm, o);
                const int ov = __shfl_xor((int)__builtin_bit_cast(uint16_t, bv), o);
                if (oe < best || (oe == best && om < bm)) { best = oe; bm = om; bv = __builtin_bit_cast(_Float16, (uint16_t)ov); }
            }
            const _Float16 uh = (_Float16)ldexp(1.0 + (double)bm * (1.0 / 1024.0), e);
            rel = (float)(best / tt);
            uv = (uint32_t)__builtin_bit_cast(uint16_t, uh) | ((uint32_t)__builtin_bit_cast(uint16_t, bv) << 16);
        }
        if (lane == 0) { cand_err[k * UV_QMAX + q] = rel; cand_uv[k * UV_QMAX + q] = uv; }
    }
}

// ONE workgroup deals the block's weighted SNPs to the runs: in rounds, every SNP not yet placed asks for the run with its
// smallest factorisation error among those that still have room; a run takes the askers in SNP order up to its capacity.
// Run r owns the slots [r * cpr * 1024, min((r + 1) * cpr, n_chunk) * 1024) and carries target r % n_target; target q's slots
// are those of its runs q, q + n_target, ... in order.  Deterministic (no atomics).
__global__ __launch_bounds__(1024) void uv_assign_kernel(const float *__restrict__ cand_err, const double2 *__restrict__ snp_tavg,
                                                         int64_t n_snp_pad, int n_target, int cpr, int n_chunk,
                                                         int32_t *__restrict__ slot_of, int32_t *__restrict__ slot_src,
                                                         const unsigned long long *__restrict__ d_missing)
{
    if (*d_missing != 0ull) return;
    __shared__ int s_rem[UV_QMAX], s_cap[UV_QMAX], s_tot[UV_QMAX];
    __shared__ int s_wsum[UV_QMAX][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // thread t takes the SNPs t, t + 1024, ...: neighbouring lanes read neighbouring 32-byte candidate records (round 5; with 64
    // consecutive SNPs per thread every load touched 64 cache lines and the kernel took 1.05 ms per 65 536-SNP block, all latency).
    // "SNP order" below is therefore the order (thread, then SNP): any fixed order makes the deal deterministic
    for (int64_t k = tid; k < n_snp_pad; k += 1024) { slot_of[k] = -1; slot_src[k] = -1; }
    const int run_len = cpr * UV_CHS;
    if (tid < UV_QMAX) {
        int cap = 0;
        if (tid < n_target)
            for (int c0 = tid * cpr; c0 < n_chunk; c0 += n_target * cpr) cap += (((c0 + cpr < n_chunk) ? (c0 + cpr) : n_chunk) - c0) * UV_CHS;
        s_cap[tid] = s_rem[tid] = cap;
    }
    __syncthreads();
    for (int rnd = 0; rnd < n_target; rnd++) {
        int rem[UV_QMAX], cnt[UV_QMAX];
#pragma unroll
        for (int q = 0; q < UV_QMAX; q++) { rem[q] = s_rem[q]; cnt[q] = 0; }
        auto choose = [&](int64_t k) -> int {
            int bq = -1;
            float be = 0.f;
#pragma unroll
            for (int q = 0; q < UV_QMAX; q++)
                if (q < n_target && rem[q] > 0) {
                    const float e = cand_err[k * UV_QMAX + q];
                    if (bq < 0 || e < be) { bq = q; be = e; }
                }
            return bq;
        };
        for (int64_t k = tid; k < n_snp_pad; k += 1024)
            if (slot_of[k] < 0 && snp_tavg[k].x > 0) {
                const int q = choose(k);
#pragma unroll
                for (int j = 0; j < UV_QMAX; j++) cnt[j] += (j == q) ? 1 : 0;
            }
        // exclusive prefix of cnt[q] over the threads (SNP order): wave scan + wave totals in LDS
        int pre[UV_QMAX];
#pragma unroll
        for (int q = 0; q < UV_QMAX; q++) {
            int x = cnt[q];
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
            pre[q] = x - cnt[q];
            if (lane == 63) s_wsum[q][wave] = x;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < UV_QMAX; q++) {
            int before = 0, tot = 0;
            for (int w = 0; w < 16; w++) { const int v = s_wsum[q][w]; if (w < wave) before += v; tot += v; }
            pre[q] += before;
            if (tid == 0) s_tot[q] = tot;
        }
        for (int64_t k = tid; k < n_snp_pad; k += 1024)
            if (slot_of[k] < 0 && snp_tavg[k].x > 0) {
                const int q = choose(k);
                int rank = 0;
#pragma unroll
                for (int j = 0; j < UV_QMAX; j++) if (j == q) { rank = pre[j]; pre[j]++; }
                if (q >= 0 && rank < rem[q]) {
                    const int pos = (s_cap[q] - rem[q]) + rank;                    // position in target q's slot list
                    const int slot = (q + n_target * (pos / run_len)) * run_len + pos % run_len;
                    slot_of[k] = slot;
                    slot_src[slot] = (int32_t)k;
                }
            }
        __syncthreads();
        int left = 0;
        if (tid == 0) {
#pragma unroll
            for (int q = 0; q < UV_QMAX; q++) {
                const int take = s_tot[q] < s_rem[q] ? s_tot[q] : s_rem[q];
                left += s_tot[q] - take;
                s_rem[q] -= take;
            }
            s_tot[0] = left;
        }
        __syncthreads();
        left = s_tot[0];
        __syncthreads();
        if (left == 0) break;
    }
}

// tables, row / column coefficients and constants of the block's slots (256 per workgroup, four chunks of 64)
__global__ __launch_bounds__(256) void uv_tables_kernel(const uint32_t *__restrict__ cand_uv, const double2 *__restrict__ snp_tavg,
                                                        const int32_t *__restrict__ slot_src, int64_t n_snp_pad, int n_target,
                                                        int cpr, uint2 *__restrict__ lut, double4 *__restrict__ uvcoef,
                                                        double *__restrict__ kpart,
                                                        const unsigned long long *__restrict__ d_missing, int swap_odd)
{
    // swap_odd == 2 (syrk_uv16c_kernel): no tables -- `lut` receives the FACTORS of the slots instead, 256 bytes per 32-slot group:
    // dword ((side * 2 + kind) * 4 + quarter) * 4 + d = the fp16 pair of slots 8 quarter + 2 d, + 1; side 0 = row (u, c_a), 1 = column
    // (v, c_b); kind 0 = 2 u, kind 1 = -c u: the operand (g - c) u = (g / 2) (2 u) - c u is ONE packed fma on the converted nibbles
    if (*d_missing != 0ull) return;
    __shared__ double s_avg[256], s_w[256], s_f[256];
    __shared__ int s_ca[256], s_cb[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t slot = (int64_t)blockIdx.x * 256 + tid;     // n_snp_pad is a multiple of 256
    const int64_t k = slot_src ? (int64_t)slot_src[slot] : slot;
    const int q = (n_target > 1) ? (int)(((slot / UV_CHS) / cpr) % n_target) : 0;
    double u = 0, v = 0, avg = 0;
    const double f = (n_target > 1) ? uv_run_factor(q) : 1.0;
    if (k >= 0) {
        const double2 ta = snp_tavg[k];
        if (ta.x > 0) {
            const uint32_t uv = cand_uv[k * UV_QMAX + q];
            u = (double)__builtin_bit_cast(_Float16, (uint16_t)(uv & 0xFFFFu));
            v = (double)__builtin_bit_cast(_Float16, (uint16_t)(uv >> 16));
            avg = ta.y;
        }
    }
    s_avg[tid] = avg; s_w[tid] = u * v; s_f[tid] = f;          // u v exact: 22 significant bits
    __syncthreads();
    if (lane == 0) {
        double cum = 0.0, ks = 0.0;
        for (int i = tid; i < tid + 64; i++) {
            const double a = s_avg[i], w = s_w[i], wf = w * s_f[i];
            int ca = 0, cb = 0;
            if (w > 0) {
                const double near = rint(a);
                double far = near + (a > near ? 1.0 : -1.0);
                if (far < 0.0 || far > 2.0) far = near;
                const double dn = a - near, df = a - far, var = 0.5 * a * (2.0 - a);
                const double mnn = dn * dn * w, mnf = dn * df * w;
                ca = cb = (int)near;
                if (far != near && (var + dn * dn) * (var + df * df) <= 6.0 * var * var && fabs(cum + mnf) < fabs(cum + mnn)) {
                    cb = (int)far; cum += mnf; ks += dn * df * wf;
                } else { cum += mnn; ks += dn * dn * wf; }
                // uvcorr_kernel sums d uv g, not d uv (g - c): the centre parts are constants and travel with K
                ks += ((a - (double)cb) * (double)ca + (a - (double)ca) * (double)cb) * wf;
            }
            s_ca[i] = ca; s_cb[i] = cb;
        }
        kpart[slot >> 6] = ks;
    }
    __syncthreads();
    const int ca = s_ca[tid], cb = s_cb[tid];
    const double yt = u * v * f;
    uvcoef[slot] = (yt > 0) ? make_double4((avg - cb) * yt, (double)ca, (avg - ca) * yt, (double)cb) : make_double4(0, 0, 0, 0);
    if (swap_odd == 2) {
        const _Float16 h[4] = {(_Float16)(2.0 * u), (_Float16)(-(double)ca * u), (_Float16)(2.0 * v), (_Float16)(-(double)cb * v)};
        uint32_t mine[4], other[4];
#pragma unroll
        for (int e = 0; e < 4; e++) mine[e] = (uint32_t)__builtin_bit_cast(uint16_t, h[e]);
#pragma unroll
        for (int e = 0; e < 4; e++) other[e] = (uint32_t)__shfl_xor((int)mine[e], 1);
        if (!(slot & 1)) {
            uint32_t *fac = reinterpret_cast<uint32_t *>(lut) + (slot >> 5) * 64;
            const int pp = (int)(slot & 31) >> 1, kq = pp >> 2, d = pp & 3;
#pragma unroll
            for (int e = 0; e < 4; e++) fac[(e * 4 + kq) * 4 + d] = mine[e] | (other[e] << 16);      // e = side * 2 + kind
        }
        return;
    }
    uint32_t ab[4], ao[4];                                    // per code: row value | column value << 16
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const _Float16 a = (c < 3) ? (_Float16)((double)(c - ca) * u) : (_Float16)0.0;
        const _Float16 b = (c < 3) ? (_Float16)((double)(c - cb) * v) : (_Float16)0.0;
        ab[c] = (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
    }
#pragma unroll
    for (int c = 0; c < 4; c++) ao[c] = (uint32_t)__shfl_xor((int)ab[c], 1);
    const bool odd = (slot & 1);
    uint2 *dst = lut + (slot >> 1) * 16 + (odd ? 8 : 0);       // the even lane writes entries 0..7, the odd lane 8..15
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int idx = e + (odd ? 8 : 0), c0 = idx & 3, c1 = idx >> 2;
        const uint32_t x0 = odd ? ao[c0] : ab[c0], x1 = odd ? ab[c1] : ao[c1];   // slot 2p, slot 2p+1
        const uint32_t rowp = (x0 & 0xFFFFu) | (x1 << 16), colp = (x0 >> 16) | (x1 & 0xFFFF0000u);
        // swap_odd (syrk_uv16_kernel): pairs of an odd 8-SNP quarter -- bit 2 of the pair index -- carry {column pair, row pair}, so
        // that the two quarters a 32-lane LDS pass spans read different banks
        dst[e] = (swap_odd && ((slot >> 3) & 1)) ? make_uint2(colp, rowp) : make_uint2(rowp, colp);
    }
}

// n_target > 1: the block's slots are dealt to n_target runs of cpr table chunks (slot_of / slot_src are written);
// n_target == 1: slot k = SNP k (slot_src may be null)
int launch_build_uv(hipStream_t st, const int32_t *sum, const int32_t *num, int64_t n_snp, int64_t n_snp_pad, int lut_mode,
                    uint2 *lut, double4 *uvcoef, double *kpart, double4 *uvsp, float *cand_err, uint32_t *cand_uv,
                    double2 *snp_tavg, int32_t *slot_of, int32_t *slot_src, int n_target, int cpr,
                    const unsigned long long *d_missing, int swap_odd)
{
    if (n_snp_pad <= 0) return 0;
    if (n_target < 1 || n_target > UV_QMAX || (n_target > 1 && ((n_snp_pad % UV_CHS) != 0 || !slot_src || !slot_of || cpr < 1))) {
        set_error("build_uv: invalid run plan");
        return 1;
    }
    const int n_chunk = (int)((n_snp_pad + UV_CHS - 1) / UV_CHS);
    hipLaunchKernelGGL(uv_factor_kernel, dim3((unsigned)((n_snp_pad + 3) / 4)), dim3(256), 0, st, sum, num, n_snp, n_snp_pad, lut_mode,
                       n_target, cand_err, cand_uv, snp_tavg, uvsp, d_missing);
    if (n_target > 1)
        hipLaunchKernelGGL(uv_assign_kernel, dim3(1), dim3(1024), 0, st, cand_err, snp_tavg, n_snp_pad, n_target, cpr, n_chunk, slot_of,
                           slot_src, d_missing);
    hipLaunchKernelGGL(uv_tables_kernel, dim3((unsigned)(n_snp_pad / 256)), dim3(256), 0, st, cand_uv, snp_tavg,
                       n_target > 1 ? slot_src : nullptr, n_snp_pad, n_target, cpr, lut, uvcoef, kpart, d_missing, swap_odd);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// Rare variants of a block without missing calls, in fp64 and with the exact weight y^2.  With g' the count of the MINOR
// allele (g or 2 - g; (g - avg) = -(g' - avg') so the products are the same) and C the carriers (g' > 0; at most
// UV_SPARSE_MAC of them):   y^2 (g'_i - avg')(g'_j - avg') = y^2 g'_i g'_j - y^2 avg' g'_i - y^2 avg' g'_j + y^2 avg'^2,
// i.e. |C|(|C| + 1) / 2 entries of the accumulator plus sparse additions to the row / column / constant terms that
// colterm_settle_kernel applies (acc[i][j] -= R[i] + Q[j] - K).  One wave per SNP: the lanes scan the SNP's packed row
// (16 bytes = 64 samples a time), collect the carriers in LDS and share out the pairs.
__global__ __launch_bounds__(256) void uv_sparse_kernel(const uint8_t *__restrict__ packed, int64_t RB, int64_t n_snp,
                                                        int64_t N, int64_t row0, int64_t row1, int64_t col0,
                                                        const double4 *__restrict__ uvsp, double *__restrict__ acc,
                                                        int64_t ld, int64_t tiles_c, int64_t ncols_pad, double *__restrict__ uvterm,
                                                        const unsigned long long *__restrict__ d_missing, int missing_blocks)
{
    // missing_blocks = 0: blocks without missing calls, the SNP has left the dense product altogether (weight 0 there).
    // missing_blocks = 1: blocks WITH missing calls (build_lut_kernel's `rare`): the dense product (exact-row kernel) still
    // holds the SNP with every called genotype replaced by the non-carrier's, i.e. y^2 avg'^2 m_i m_j; what is added here is the
    // rest of y^2 (g'_i - avg')(g'_j - avg') m_i m_j: the carrier pairs' y^2 g'_i g'_j, the carriers' row / column terms
    // y^2 avg' g'_i -- which colterm_settle_kernel subtracts from EVERY entry of the carrier's row and column, so they are
    // given back at the cells (carrier, sample with a missing call), whose pair does not count -- and no constant.
    if (missing_blocks ? (*d_missing == 0ull) : (*d_missing != 0ull)) return;
    constexpr int MAXC = X1_SPARSE_MAC > UV_SPARSE_MAC ? X1_SPARSE_MAC : UV_SPARSE_MAC;
    __shared__ int s_idx[4][MAXC];
    __shared__ int s_g[4][MAXC];
    __shared__ int s_cnt[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t k = (int64_t)blockIdx.x * 4 + wave;
    if (k >= n_snp) return;
    const double4 sp = uvsp[k];
    if (sp.w == 0.0) return;                       // wave-uniform
    if (lane == 0) s_cnt[wave] = 0;
    __builtin_amdgcn_wave_barrier();
    const bool flip = (sp.z != 0.0);
    const uint8_t *__restrict__ row = packed + k * RB;
    for (int64_t b0 = (int64_t)lane * 16; b0 < RB; b0 += 64 * 16) {
        const uint4 q = *reinterpret_cast<const uint4 *>(row + b0);     // RB is a multiple of 64 bytes (samples padded with code 3)
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int ws = 0; ws < 4; ws++) {
            if ((flip ? (w[ws] != 0xAAAAAAAAu) : (w[ws] != 0u))) {   // sixteen samples without a copy of the minor allele
                for (int j = 0; j < 16; j++) {
                    const uint32_t code = (w[ws] >> (2 * j)) & 3u;
                    const int64_t smp = b0 * 4 + ws * 16 + j;
                    if (code == 3u || smp >= N) continue;
                    const int gp = flip ? 2 - (int)code : (int)code;
                    if (gp > 0) {
                        const int slot = atomicAdd(&s_cnt[wave], 1);
                        if (slot < MAXC) { s_idx[wave][slot] = (int)smp; s_g[wave][slot] = gp; }
                    }
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const int cnt = s_cnt[wave] < MAXC ? s_cnt[wave] : MAXC;   // <= the mode's copy limit by construction
    const double y2 = sp.x, ya = sp.x * sp.y;
    for (int a = lane; a < cnt; a += 64) {
        const int64_t c = (int64_t)s_idx[wave][a] - col0;
        if (c >= 0) {
            const double t = ya * (double)s_g[wave][a];
            unsafeAtomicAdd(uvterm + c, t);
            unsafeAtomicAdd(uvterm + ncols_pad + c, t);
        }
    }
    if (lane == 0 && !missing_blocks) unsafeAtomicAdd(uvterm + 2 * ncols_pad, ya * sp.y);
    if (missing_blocks && cnt > 0) {
        for (int64_t b0 = (int64_t)lane * 16; b0 < RB; b0 += 64 * 16) {
            const uint4 q = *reinterpret_cast<const uint4 *>(row + b0);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int ws = 0; ws < 4; ws++) {
                uint32_t miss = w[ws] & (w[ws] >> 1) & 0x55555555u;      // code 3 = both bits
                while (miss) {
                    const int bit = __ffs((int)miss) - 1;
                    miss &= miss - 1;
                    const int64_t smp = b0 * 4 + ws * 16 + (bit >> 1);
                    if (smp >= N) break;                                  // sample padding
                    for (int a = 0; a < cnt; a++) {
                        const int64_t ca = s_idx[wave][a];
                        const int64_t i = ca < smp ? ca : smp, j = ca < smp ? smp : ca;
                        if (i >= row0 && i < row1)
                            unsafeAtomicAdd(acc + acc_off(ld, tiles_c, i - col0, j - col0), ya * (double)s_g[wave][a]);
                    }
                }
            }
        }
    }
    const int n_pair = cnt * (cnt + 1) / 2;
    for (int pi = lane; pi < n_pair; pi += 64) {
        // pair number pi -> (a <= b): row b of the lower triangle
        int b = (int)((sqrt(8.0 * pi + 1.0) - 1.0) * 0.5);
        while (b * (b + 1) / 2 > pi) b--;
        while ((b + 1) * (b + 2) / 2 <= pi) b++;
        const int a = pi - b * (b + 1) / 2;
        const int sa = s_idx[wave][a], sb = s_idx[wave][b];
        const int64_t i = sa < sb ? sa : sb, j = sa < sb ? sb : sa;
        if (i >= row0 && i < row1)
            unsafeAtomicAdd(acc + acc_off(ld, tiles_c, i - col0, j - col0), y2 * (double)(s_g[wave][a] * s_g[wave][b]));
    }
}

int launch_uv_sparse(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t N, int64_t row0, int64_t row1,
                     int64_t col0, const double4 *uvsp, double *acc, int64_t ld, int64_t tiles_c, int64_t ncols_pad, double *uvterm,
                     const unsigned long long *d_missing, int missing_blocks)
{
    if (n_snp <= 0) return 0;
    hipLaunchKernelGGL(uv_sparse_kernel, dim3((unsigned)((n_snp + 3) / 4)), dim3(256), 0, st, packed, RB, n_snp, N, row0, row1, col0,
                       uvsp, acc, ld, tiles_c, ncols_pad, uvterm, d_missing, missing_blocks);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// Row / column terms of the single-product SYRK: per sample j, over the block's SNPs,
//   R[j] += sum d_b u v g_js      Q[j] += sum d_a u v g_js       (fp64; bytes of W8 = 8 * (c0 + 4 * c1))
// in per-chunk partial sums added in chunk order (independent of the launch geometry); the centre parts
// sum d_b u v c_a + sum d_a u v c_b are the same for every sample and sit in K with sum d_a d_b u v (uv_tables_kernel).
// Code 3 occurs only as SNP padding (coefficients 0) and sample padding (terms never read): no special case.
__global__ __launch_bounds__(256) void uvcorr_kernel(const uint32_t *__restrict__ w8, int64_t ncols_pad, int n_d,
                                                     const double4 *__restrict__ uvcoef, double2 *__restrict__ tc,
                                                     const unsigned long long *__restrict__ d_missing, int nibble)
{
    if (*d_missing != 0ull) return;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= ncols_pad) return;
    const int d0 = blockIdx.y * (H3_LUTCH / 16);
    const int d1 = (d0 + H3_LUTCH / 16 < n_d) ? (d0 + H3_LUTCH / 16) : n_d;
    double sr = 0.0, sq = 0.0;
    for (int d = d0; d < d1; d++) {
        const uint32_t w = w8[(int64_t)d * ncols_pad + col];
        const double4 *__restrict__ cf = uvcoef + (int64_t)d * 8;    // wave-uniform: scalar loads
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t by = (w >> (8 * p)) & 0xFFu, b = by >> 3;    // 8 * (c0 + 4 c1), or the nibble form c0 | c1 << 4
            const double g0 = nibble ? (double)(by & 3u) : (double)(b & 3u), g1 = nibble ? (double)(by >> 4) : (double)(b >> 2);
            const double4 f0 = cf[2 * p], f1 = cf[2 * p + 1];
            sr = fma(f0.x, g0, sr); sq = fma(f0.z, g0, sq);
            sr = fma(f1.x, g1, sr); sq = fma(f1.z, g1, sq);
        }
    }
    tc[(int64_t)blockIdx.y * ncols_pad + col] = make_double2(sr, sq);
}

__global__ __launch_bounds__(256) void uvterm_add_kernel(const double2 *__restrict__ tc, int n_chunk, int64_t ncols_pad,
                                                         const double *__restrict__ kpart, int n_kpart,
                                                         double *__restrict__ uvterm,
                                                         const unsigned long long *__restrict__ d_missing)
{
    if (*d_missing != 0ull) return;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col == 0) {
        double ks = uvterm[2 * ncols_pad];
        for (int i = 0; i < n_kpart; i++) ks += kpart[i];
        uvterm[2 * ncols_pad] = ks;
    }
    if (col >= ncols_pad) return;
    double sr = uvterm[col], sq = uvterm[ncols_pad + col];
    for (int k = 0; k < n_chunk; k++) { const double2 t = tc[(int64_t)k * ncols_pad + col]; sr += t.x; sq += t.y; }
    uvterm[col] = sr; uvterm[ncols_pad + col] = sq;
}

int launch_uvcorr(hipStream_t st, const uint32_t *w8, int64_t ncols_pad, int n_d, const double4 *uvcoef, const double *kpart,
                  int n_kpart, double2 *tc, double *uvterm, const unsigned long long *d_missing, int nibble)
{
    if (n_d <= 0) return 0;
    const int n_chunk = (n_d + H3_LUTCH / 16 - 1) / (H3_LUTCH / 16);
    dim3 grid((unsigned)((ncols_pad + 255) / 256), (unsigned)n_chunk);
    hipLaunchKernelGGL(uvcorr_kernel, grid, dim3(256), 0, st, w8, ncols_pad, n_d, uvcoef, tc, d_missing, nibble);
    hipLaunchKernelGGL(uvterm_add_kernel, dim3(grid.x), dim3(256), 0, st, tc, n_chunk, ncols_pad, kpart, n_kpart, uvterm, d_missing);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// KING-homo, blocks WITH missing calls (round 5).  The masked weight sums SumAFreq(i, j) = sum over the SNPs where BOTH samples are
// called of c_s, c = p (1 - p) resp. (p (1 - p))^2 (src/genKING.cpp:236-248), were two-product fp16 SYRKs of an indicator against a
// hi / lo column operand.  With the missing indicator mu:  sum_s c_s (1 - mu_is)(1 - mu_js) = C - M_i - M_j + B_ij,
//     C = sum_s c_s,    M_i = sum_s c_s mu_is  (per sample, fp64, O(N B)),    B_ij = sum_s c_s mu_is mu_js,
// and only B is a pair contraction -- of BINARY operands, so c_s = u v with two fp16 numbers makes it ONE exact product per SNP
// (syrk_uv_kernel's arithmetic: row value u, column value v for code 3, zero otherwise); a factorisation error of 1e-6 (best of the
// 1024 mantissas of u, as uv_factor_kernel) meets a term that is f^2 of the sum.  u v IS the SNP's weight in C and M as well.
// homo_uv_tables_kernel: one wave per SNP, both weights: tables (8-byte entries {row pair, column pair}, syrk_uv_kernel's format),
// the effective weights {w1, w2} (x 2^-16: the tables carry 2^16 c so that (p(1-p))^2 ~ 1e-10 stays in fp16's normal range) and
// the block totals into the context's two KING-homo scalars.
__global__ __launch_bounds__(256) void homo_uv_tables_kernel(const int32_t *__restrict__ sum, const int32_t *__restrict__ num,
                                                             int64_t n_snp, int64_t n_snp_pad, uint2 *__restrict__ lut1,
                                                             uint2 *__restrict__ lut2, double2 *__restrict__ wts,
                                                             double *__restrict__ totals,
                                                             const unsigned long long *__restrict__ d_missing, int swap_odd)
{
    if (*d_missing == 0ull) return;               // blocks without missing calls: every pair gets the whole sum (build_lut_kernel)
    const int lane = threadIdx.x & 63;
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n_snp_pad) return;
    double c1 = 0.0;
    if (k < n_snp) {
        const int s = sum[k], c = num[k];
        const double p = (c > 0) ? (0.5 * s / c) : 0.0;       // genKING.cpp:236-248
        c1 = p * (1 - p);
    }
    uint32_t uv[2] = {0u, 0u};
    double weff[2] = {0.0, 0.0};
    for (int t = 0; t < 2; t++) {
        const double tt = ldexp(t == 0 ? c1 : c1 * c1, 2 * H3_HOMO_SHIFT);
        if (!(tt > 0) || tt < 1e-7) continue;                  // wave-uniform (weights below 2^-16 x 1e-7 ~ 1e-12 count as zero)
        const int e = ilogb(sqrt(tt));
        const float tf = (float)tt;
        double best = 1e300;
        int bm = 0;
        _Float16 bv = (_Float16)0.0;
#pragma unroll 4
        for (int i = 0; i < 16; i++) {
            const int m = lane * 16 + i;
            const double uc = ldexp(1.0 + (double)m * (1.0 / 1024.0), e);
            const _Float16 vh = (_Float16)(tf / (float)uc);
            const double err = fabs(uc * (double)vh - tt);
            if (err < best) { best = err; bm = m; bv = vh; }
        }
        for (int o = 32; o; o >>= 1) {                        // arg-min over the wave; ties to the smaller mantissa
            const double oe = __shfl_xor(best, o);
            const int om = __shfl_xor(bm, o);
            const int ov = __shfl_xor((int)__builtin_bit_cast(uint16_t, bv), o);
            if (oe < best || (oe == best && om < bm)) { best = oe; bm = om; bv = __builtin_bit_cast(_Float16, (uint16_t)ov); }
        }
        const _Float16 uh = (_Float16)ldexp(1.0 + (double)bm * (1.0 / 1024.0), e);
        uv[t] = (uint32_t)__builtin_bit_cast(uint16_t, uh) | ((uint32_t)__builtin_bit_cast(uint16_t, bv) << 16);
        weff[t] = ldexp((double)uh * (double)bv, -2 * H3_HOMO_SHIFT);
    }
    if (lane == 0) wts[k] = make_double2(weff[0], weff[1]);        // (the block totals: homo_totals_kernel, in a fixed order)
    // pair table of slots (2p, 2p+1): entry c0 + 4 c1 = {row value of slot 2p | of slot 2p+1 << 16, column values likewise}; lanes
    // 0..15 write the 16 entries of this SNP's pair, this SNP's half of each (the partner wave of the pair writes the other half)
    if (lane < 16) {
        const int c0 = lane & 3, c1i = lane >> 2;
        const bool odd = (k & 1);
        const bool mine3 = odd ? (c1i == 3) : (c0 == 3);
        for (int t = 0; t < 2; t++) {
            uint16_t *e16 = reinterpret_cast<uint16_t *>((t == 0 ? lut1 : lut2) + (k >> 1) * 16 + lane);
            const int sw = (swap_odd && ((k >> 3) & 1)) ? 2 : 0;      // odd quarters: {column pair, row pair} (syrk_uv16_kernel)
            e16[(odd ? 1 : 0) + sw] = mine3 ? (uint16_t)(uv[t] & 0xFFFFu) : (uint16_t)0;       // row value (u)
            e16[(odd ? 3 : 2) - sw] = mine3 ? (uint16_t)(uv[t] >> 16) : (uint16_t)0;           // column value (v)
        }
    }
}

// totals[0..1] += the block's sums of the two effective weights: ONE workgroup, strided partial sums, wave and LDS reduction in a
// fixed order (65 536 waves adding to one address with atomics took 1.5 ms per block and depended on their arrival order)
__global__ __launch_bounds__(1024) void homo_totals_kernel(const double2 *__restrict__ wts, int64_t n, double *__restrict__ totals,
                                                           const unsigned long long *__restrict__ d_missing)
{
    if (*d_missing == 0ull) return;
    __shared__ double s1[16], s2[16];
    double a = 0.0, b = 0.0;
    for (int64_t k = threadIdx.x; k < n; k += 1024) { const double2 w = wts[k]; a += w.x; b += w.y; }
    for (int o = 32; o; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); }
    if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = a; s2[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int w = 0; w < 16; w++) { ta += s1[w]; tb += s2[w]; }
        totals[0] += ta; totals[1] += tb;
    }
}

// per-sample sums M1[j] += sum_s w1_s mu_js, M2 likewise, from the pair-coded words (byte = 8 (c0 + 4 c1)): per-chunk partials
// added in chunk order (independent of the launch geometry), as uvcorr_kernel / uvterm_add_kernel
__global__ __launch_bounds__(256) void homo_miss_sums_kernel(const uint32_t *__restrict__ w8, int64_t ncols_pad, int n_d,
                                                             const double2 *__restrict__ wts, double2 *__restrict__ tc,
                                                             const unsigned long long *__restrict__ d_missing)
{
    if (*d_missing == 0ull) return;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= ncols_pad) return;
    const int d0 = blockIdx.y * (H3_LUTCH / 16);
    const int d1 = (d0 + H3_LUTCH / 16 < n_d) ? (d0 + H3_LUTCH / 16) : n_d;
    double s1 = 0.0, s2 = 0.0;
    for (int d = d0; d < d1; d++) {
        const uint32_t w = w8[(int64_t)d * ncols_pad + col];
        const double2 *__restrict__ cf = wts + (int64_t)d * 8;       // wave-uniform: scalar loads
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t b = ((w >> (8 * p)) & 0xFFu) >> 3;
            if ((b & 3u) == 3u) { s1 += cf[2 * p].x; s2 += cf[2 * p].y; }
            if ((b >> 2) == 3u) { s1 += cf[2 * p + 1].x; s2 += cf[2 * p + 1].y; }
        }
    }
    tc[(int64_t)blockIdx.y * ncols_pad + col] = make_double2(s1, s2);
}

__global__ __launch_bounds__(256) void homo_miss_add_kernel(const double2 *__restrict__ tc, int n_chunk, int64_t ncols_pad,
                                                            double *__restrict__ msum,
                                                            const unsigned long long *__restrict__ d_missing)
{
    if (*d_missing == 0ull) return;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= ncols_pad) return;
    double s1 = msum[col], s2 = msum[ncols_pad + col];
    for (int k = 0; k < n_chunk; k++) { const double2 t = tc[(int64_t)k * ncols_pad + col]; s1 += t.x; s2 += t.y; }
    msum[col] = s1; msum[ncols_pad + col] = s2;
}

int launch_homo_uv(hipStream_t st, const int32_t *sum, const int32_t *num, int64_t n_snp, int64_t n_snp_pad, uint2 *lut1, uint2 *lut2,
                   double2 *wts, double *totals, const uint32_t *w8, int64_t ncols_pad, double2 *tc, double *msum,
                   const unsigned long long *d_missing, int swap_odd)
{
    if (n_snp_pad <= 0) return 0;
    // tables of whole 1024-slot chunks (syrk_uv_kernel copies whole chunks): zero weights beyond the block
    const int64_t n_tab = (n_snp_pad + UV_CHS - 1) / UV_CHS * UV_CHS;
    hipLaunchKernelGGL(homo_uv_tables_kernel, dim3((unsigned)((n_tab + 3) / 4)), dim3(256), 0, st, sum, num, n_snp, n_tab, lut1, lut2, wts,
                       totals, d_missing, swap_odd);
    hipLaunchKernelGGL(homo_totals_kernel, dim3(1), dim3(1024), 0, st, wts, n_tab, totals, d_missing);
    const int n_d = (int)(n_snp_pad / 8);
    const int n_chunk = (n_d + H3_LUTCH / 16 - 1) / (H3_LUTCH / 16);
    dim3 grid((unsigned)((ncols_pad + 255) / 256), (unsigned)n_chunk);
    hipLaunchKernelGGL(homo_miss_sums_kernel, grid, dim3(256), 0, st, w8, ncols_pad, n_d, wts, tc, d_missing);
    hipLaunchKernelGGL(homo_miss_add_kernel, dim3(grid.x), dim3(256), 0, st, tc, n_chunk, ncols_pad, msum, d_missing);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// bitplanes: each wave owns 64 SNPs (one per lane on the read side) x 64 samples.
// Lane l reads the 16 bytes holding samples s0..s0+63 of SNP k0+l; for every sample s a wave
// ballot of "code(s) has property P" is the 64-SNP plane word of that sample, which lane s keeps
// (lane = sample on the write side).  Planes per sample:
//   V = call present, H = heterozygous (g==1), O = g==0, T = g==2      (all zero when missing)
// so that the pair kernel needs 8 (IBS) / 11 (KING) bit-ops per 32 SNP pairs.
// Output word index kw = snp/32; planes of one (sample, kw) are one uint4 {V,H,O,T}.
template <int MISS_ONLY>
__global__ __launch_bounds__(256) void bitplanes_kernel(const uint8_t *__restrict__ packed, int64_t RB,
                                                        int64_t n_snp, int64_t N, const int32_t *__restrict__ sum,
                                                        const int32_t *__restrict__ num, int64_t col0,
                                                        int64_t ncols_pad, int64_t rows_pad, int KW,
                                                        void *__restrict__ rowp_, void *__restrict__ colp_,
                                                        const unsigned long long *__restrict__ d_skip_if_zero)
{
    if (MISS_ONLY && d_skip_if_zero && *d_skip_if_zero == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t k0 = ((int64_t)blockIdx.y * 4 + wave) * 64;  // first SNP of this wave
    if (k0 >= (int64_t)KW * 32) return;
    const int64_t sc0 = (int64_t)blockIdx.x * 64;              // first column sample (panel relative)
    const int64_t s0 = col0 + sc0;                             // absolute sample
    const int64_t k = k0 + lane;
    uint4 q = make_uint4(~0u, ~0u, ~0u, ~0u);
    bool poly = false;
    if (k < n_snp) {
        if (s0 < RB * 4) q = *reinterpret_cast<const uint4 *>(packed + k * RB + (s0 >> 2));
        if (MISS_ONLY) {
            const int s = sum[k], c = num[k];
            poly = (0 < s) && (s < 2 * c);
        }
    }
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    uint32_t r0[4] = {0, 0, 0, 0}, r1[4] = {0, 0, 0, 0};  // lo (SNP k0..k0+31) / hi (k0+32..) words, planes V,H,O,T
#pragma unroll
    for (int ws = 0; ws < 4; ws++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int s = ws * 16 + j;
            const uint32_t code = (w[ws] >> (2 * j)) & 3u;
            const bool mine = (lane == s);
            if (MISS_ONLY) {
                // plane 0: missing call at a polymorphic SNP, real samples only (genPCA.cpp:1201-1224)
                const bool in_range = (s0 + s) < N;
                const unsigned long long m = __ballot(code == 3u && poly && in_range);
                if (mine) { r0[0] = (uint32_t)m; r1[0] = (uint32_t)(m >> 32); }
            } else {
                const unsigned long long mv = __ballot(code != 3u);
                const unsigned long long mh = __ballot(code == 1u);
                const unsigned long long mo = __ballot(code == 0u);
                const unsigned long long mt = __ballot(code == 2u);
                if (mine) {
                    r0[0] = (uint32_t)mv; r1[0] = (uint32_t)(mv >> 32);
                    r0[1] = (uint32_t)mh; r1[1] = (uint32_t)(mh >> 32);
                    r0[2] = (uint32_t)mo; r1[2] = (uint32_t)(mo >> 32);
                    r0[3] = (uint32_t)mt; r1[3] = (uint32_t)(mt >> 32);
                }
            }
        }
    }
    const int kw0 = (int)(k0 >> 5);
    const int64_t sc = sc0 + lane;  // panel-relative sample of this lane
    if (MISS_ONLY) {
        uint2 *rowp = (uint2 *)rowp_;
        uint2 *colp = (uint2 *)colp_;
        const int kp = kw0 >> 1;  // uint2 = two consecutive 32-SNP words
        colp[(int64_t)kp * ncols_pad + sc] = make_uint2(r0[0], r1[0]);
        if (sc < rows_pad) rowp[((sc >> 3) * (KW >> 1) + kp) * 8 + (sc & 7)] = make_uint2(r0[0], r1[0]);
    } else {
        uint4 *rowp = (uint4 *)rowp_;
        uint4 *colp = (uint4 *)colp_;
        const uint4 a = make_uint4(r0[0], r0[1], r0[2], r0[3]);
        const uint4 b = make_uint4(r1[0], r1[1], r1[2], r1[3]);
        colp[(int64_t)kw0 * ncols_pad + sc] = a;
        colp[(int64_t)(kw0 + 1) * ncols_pad + sc] = b;
        if (sc < rows_pad) {
            // [row group of 8][word][8 rows]: the pair kernel's wave reads 8 rows of one word at once
            rowp[((sc >> 3) * KW + kw0) * 8 + (sc & 7)] = a;
            rowp[((sc >> 3) * KW + kw0 + 1) * 8 + (sc & 7)] = b;
        }
    }
}

// ---------------------------------------------------------------------------
// Transposition of a 64 x 64 matrix of 2-bit elements spread over a wave: lane r holds row r as 128 bits (element e at
// bits 2e of x[0..3]); on return lane r holds column r in the same form.  Recursive block swap, blocks of 32, 16, 8, 4, 2, 1
// elements: a lane of the upper half of a block pair (bit j of the lane clear) keeps its elements with bit j clear and takes
// its partner's elements with bit j clear into the positions with bit j set; the lower half the other way round.  The two
// widest blocks move whole dwords, the others cost one rotate and one bit-field insert per dword: ~70 vector instructions and
// 20 cross-lane moves for 4096 genotypes, where the ballot form (one ballot per sample and bit plane, kept by the one lane
// it belongs to) took ~500 -- the pre-pass kernels were bound by exactly those.
__device__ __forceinline__ void transpose_2bit_64x64(uint32_t (&x)[4], int lane)
{
    {
        const bool hi = (lane & 32) != 0;
        const uint32_t r0 = (uint32_t)__shfl_xor((int)(hi ? x[0] : x[2]), 32), r1 = (uint32_t)__shfl_xor((int)(hi ? x[1] : x[3]), 32);
        if (hi) { x[0] = r0; x[1] = r1; } else { x[2] = r0; x[3] = r1; }
    }
    {
        const bool hi = (lane & 16) != 0;
        const uint32_t r0 = (uint32_t)__shfl_xor((int)(hi ? x[0] : x[1]), 16), r1 = (uint32_t)__shfl_xor((int)(hi ? x[2] : x[3]), 16);
        if (hi) { x[0] = r0; x[2] = r1; } else { x[1] = r0; x[3] = r1; }
    }
#pragma unroll
    for (int st = 0; st < 4; st++) {
        const int j = 8 >> st;                                        // elements per block
        const uint32_t m = st == 0 ? 0x0000FFFFu : st == 1 ? 0x00FF00FFu : st == 2 ? 0x0F0F0F0Fu : 0x33333333u;
        const int sh = 2 * j;                                         // bits per block
        const bool hi = (lane & j) != 0;
        const uint32_t keep = hi ? ~m : m;
        const uint32_t rot = hi ? (uint32_t)sh : (uint32_t)(32 - sh); // rotate right: upper half takes y << sh, lower y >> sh
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t y = (uint32_t)__shfl_xor((int)x[d], j);
            const uint32_t r = __builtin_amdgcn_alignbit(y, y, rot);
            x[d] = (x[d] & keep) | (r & ~keep);
        }
    }
}

// Read side of the transposition kernels: a workgroup takes 64 SNPs (or slots) x TR_SAMPLES samples of the repacked block
// (rows of RB bytes, a multiple of 64, samples >= N already code 3).  Every wave instruction reads 256 contiguous bytes of ONE
// row (lane = 16 samples) into the LDS tile; the waves then pick their 64 x 64 sub-tiles from it with one 16-byte read per lane.
// (Before, a lane read 16 bytes of its own row: 64 cache lines per load instruction, which -- not the bit work -- set the time.)
// row_of(r) = row of `packed` for tile row r, or -1 for a row of `fill`.
constexpr int TR_SAMPLES = 1024;
constexpr int TR_PITCH = TR_SAMPLES / 16 + 4;         // dwords per tile row (16-byte aligned)
template <typename RowOf>
__device__ __forceinline__ void load_tile_64(uint32_t (*tile)[TR_PITCH], const uint8_t *__restrict__ packed, int64_t RB,
                                             int64_t s_first, uint32_t fill, RowOf row_of)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = (s_first >> 2) + 4 * lane;        // byte offset of this lane's dword in a row
    for (int r = wave; r < 64; r += 4) {
        const int64_t k = row_of(r);
        tile[r][lane] = (k >= 0 && b + 4 <= RB) ? *reinterpret_cast<const uint32_t *>(packed + k * RB + b) : fill;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// transpose8: SNP-major 2-bit rows -> sample-major PAIR-coded words for the SYRK kernel.
//   W8[d][sample] (uint32) covers SNPs 8d .. 8d+7 of that sample: byte p = 8 * (c0 + 4*c1) with
//   c0/c1 the codes of SNPs 8d+2p / 8d+2p+1, i.e. the byte offset of the pair's float2 table entry:
//   ONE v_add_u32_sdwa (table address = base + byte) per two genotypes, no shift/mask.
// Same wave-ballot scheme as bitplanes: lane = SNP on the read side, lane = sample on the write side.
// d_wide16 != nullptr and *d_wide16 == 0 (a block without missing calls in a context with the exact-row SYRK): the
// byte is 16 * (c0 + 4*c1), the offset of a 16-byte table entry.
__global__ __launch_bounds__(256) void transpose8_kernel(const uint8_t *__restrict__ packed, int64_t RB,
                                                         int64_t n_snp, int64_t col0, int64_t ncols_pad,
                                                         int n_d, uint32_t *__restrict__ w8,
                                                         const unsigned long long *__restrict__ d_wide16, int always_wide,
                                                         const int32_t *__restrict__ slot_src, int nibble_nomiss)
{
    // bytes carry the table offset of the pair's entry: 8 / 16 * code (always_wide == 1), or 12 * code (always_wide == 2)
    // always_wide == 3: 12 * code, or 8 * code in a block without missing calls (syrk_uv_kernel: 8-byte entries)
    // always_wide == 4: 12 * code, and only for a block WITH missing calls (EIGMIX: a second word array for the exact-row
    // kernel next to the 8 * code words its other tables read)
    // nibble_nomiss (syrk_uv16c_kernel, always_wide == 3): in a block without missing calls byte p = c0 | c1 << 4 -- two e2m1 nibbles
    // of value c / 2 that v_cvt_scalef32_pk_f16_fp4 turns into an fp16 pair, no table
    if (always_wide == 4 && *d_wide16 == 0ull) return;
    const bool nib = nibble_nomiss && always_wide == 3 && *d_wide16 == 0ull;
    const uint32_t mul = (always_wide == 3) ? ((*d_wide16 == 0ull) ? 8u : 12u)
                         : (always_wide == 2 || always_wide == 4) ? 12u : (always_wide || (d_wide16 && *d_wide16 == 0ull)) ? 16u : 8u;
    __shared__ uint32_t tile[64][TR_PITCH];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t k0 = (int64_t)blockIdx.y * 64;
    if (k0 >= (int64_t)n_d * 8) return;
    const int64_t sc_wg = (int64_t)blockIdx.x * TR_SAMPLES;
    // the K dimension of a block without missing calls that runs as several fp32 runs is a list of SLOTS (uv_assign_kernel deals
    // the SNPs to the runs): slot_src maps them to the block's SNPs (-1: empty)
    const bool slots = slot_src && always_wide == 3 && *d_wide16 == 0ull;
    load_tile_64(tile, packed, RB, col0 + sc_wg, ~0u, [&](int r) -> int64_t {
        const int64_t k = k0 + r;
        return slots ? (int64_t)slot_src[k] : (k < n_snp ? k : (int64_t)-1);
    });
    const int d0 = (int)(k0 >> 3);
    for (int cc = wave; cc < TR_SAMPLES / 64; cc += 4) {
        const int64_t sc = sc_wg + 64 * cc + lane;
        if (sc - lane >= ncols_pad) break;
        const uint4 q = *reinterpret_cast<const uint4 *>(&tile[lane][4 * cc]);
        uint32_t x[4] = {q.x, q.y, q.z, q.w};
        transpose_2bit_64x64(x, lane);               // lane = sample now: x = the codes of the 64 SNPs (slots)
#pragma unroll
        for (int g = 0; g < 8; g++) {   // 8 SNPs = 4 pairs per output word: nibble p = c0 + 4 c1 of pair p -> byte p = nibble * mul
            uint32_t v = (x[g >> 1] >> (16 * (g & 1))) & 0xFFFFu;
            v = (v | (v << 8)) & 0x00FF00FFu;
            v = (v | (v << 4)) & 0x0F0F0F0Fu;
            w8[(int64_t)(d0 + g) * ncols_pad + sc] = nib ? ((v & 0x03030303u) | ((v & 0x0C0C0C0Cu) << 2))
                                                         : v * mul;  // 15 * 16 < 256: no carry between the bytes
        }
    }
}

int launch_transpose8(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t col0,
                      int64_t ncols_pad, int n_d, uint32_t *w8, const unsigned long long *d_wide16, int always_wide,
                      const int32_t *slot_src, int nibble_nomiss)
{
    dim3 grid((unsigned)((ncols_pad + TR_SAMPLES - 1) / TR_SAMPLES), (unsigned)((n_d + 7) / 8));   // groups of 64 SNPs (slots)
    hipLaunchKernelGGL(transpose8_kernel, grid, dim3(256), 0, st, packed, RB, n_snp, col0, ncols_pad, n_d, w8, d_wide16,
                       always_wide, slot_src, nibble_nomiss);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// sample-major 2-bit words for the MFMA pair kernels: W2[d / 2][sample][d & 1] = codes of SNPs 16d .. 16d+15
// (code m at bits 2m), same ballot transposition as above; SNPs >= n_snp and samples >= N are 3
// (missing -> every operand value 0).
__device__ __forceinline__ uint32_t spread16(uint32_t x)
{
    x &= 0xFFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

// MASK = 1 (GCTA denominators): code 3 only for "missing call at a polymorphic SNP of a real sample"
// (genPCA.cpp:1201-1224), every other cell 0; exits when the block holds no missing call.
template <int MASK>
__global__ __launch_bounds__(256) void transpose2_kernel(const uint8_t *__restrict__ packed, int64_t RB,
                                                         int64_t n_snp, int64_t col0, int64_t ncols_pad,
                                                         int n_d, uint32_t *__restrict__ w2, int64_t N,
                                                         const int32_t *__restrict__ sum, const int32_t *__restrict__ num,
                                                         const unsigned long long *__restrict__ d_skip_if_zero,
                                                         uint32_t *__restrict__ het, int classic)
{
    if (MASK && d_skip_if_zero && *d_skip_if_zero == 0ull) return;
    __shared__ uint32_t tile[64][TR_PITCH];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t k0 = (int64_t)blockIdx.y * 64;
    if (k0 >= (int64_t)n_d * 16) return;
    const int64_t sc_wg = (int64_t)blockIdx.x * TR_SAMPLES;
    load_tile_64(tile, packed, RB, col0 + sc_wg, MASK ? 0u : ~0u, [&](int r) -> int64_t {
        const int64_t k = k0 + r;
        if (k >= n_snp) return -1;
        if (MASK) {                                  // only polymorphic SNPs count (genPCA.cpp:1206)
            const int s = sum[k], c = num[k];
            if (!((0 < s) && (s < 2 * c))) return -1;
        }
        return k;
    });
    const int d0 = (int)(k0 >> 4);
    for (int cc = wave; cc < TR_SAMPLES / 64; cc += 4) {
        const int64_t sc0 = sc_wg + 64 * cc;
        if (sc0 >= ncols_pad) break;
        const uint4 q = *reinterpret_cast<const uint4 *>(&tile[lane][4 * cc]);
        uint32_t x[4] = {q.x, q.y, q.z, q.w};
        if (MASK) {
            // code 3 only where a real sample has a missing call (at a polymorphic SNP: the others were loaded as 0), 0 elsewhere
            const int64_t rem = N - (col0 + sc0);    // samples of this 64-chunk that exist
#pragma unroll
            for (int t = 0; t < 4; t++) {
                uint32_t m3 = x[t] & (x[t] >> 1) & 0x55555555u;
                const int64_t r = rem - 16 * t;
                if (r <= 0) m3 = 0u;
                else if (r < 16) m3 &= (1u << (2 * r)) - 1u;
                x[t] = m3 | (m3 << 1);
            }
        }
        transpose_2bit_64x64(x, lane);               // lane = sample now
        const int64_t sc = sc0 + lane;
        if (classic) {   // W2[d][sample]: the projection kernels (lane = SNP) read it
#pragma unroll
            for (int t = 0; t < 4; t++) w2[(int64_t)(d0 + t) * ncols_pad + sc] = x[t];
        } else {   // word rows 2 r and 2 r + 1 of a sample lie side by side (W2 = uint2[row pair][sample]): one 8-byte load per 32 SNPs
            uint2 *__restrict__ w2p = reinterpret_cast<uint2 *>(w2);
            w2p[(int64_t)(d0 >> 1) * ncols_pad + sc] = make_uint2(x[0], x[1]);
            w2p[(int64_t)((d0 >> 1) + 1) * ncols_pad + sc] = make_uint2(x[2], x[3]);
        }
        // per-sample het counts of a block WITHOUT missing calls: the rank-one terms of the binary pair kernel
        // (I8Scheme<PM_IBS_NOMISS>); d_skip_if_zero is the block's missing-call flag here
        // (het[0 .. ncols_pad) = #het, het[ncols_pad .. 2 ncols_pad) = #(g == 2))
        if (!MASK && het && *d_skip_if_zero == 0ull) {
            uint32_t c = 0, t2 = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                c += (uint32_t)__popc(x[t] & ~(x[t] >> 1) & 0x55555555u);
                t2 += (uint32_t)__popc(~x[t] & (x[t] >> 1) & 0x55555555u);
            }
            if (c) atomicAdd(het + sc, c);
            if (t2) atomicAdd(het + ncols_pad + sc, t2);
        }
    }
}

// per-sample number of code-3 cells of the masked words, added to diag[col0 + sample] (M(s,s) of the GCTA denominators)
__global__ __launch_bounds__(256) void miss_diag2_kernel(const uint32_t *__restrict__ w2, int n_d, int64_t ncols_pad,
                                                         int64_t col0, uint32_t *__restrict__ diag,
                                                         const unsigned long long *__restrict__ d_skip_if_zero)
{
    if (d_skip_if_zero && *d_skip_if_zero == 0ull) return;
    const int64_t sc = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (sc >= ncols_pad) return;
    uint32_t c = 0;
    const uint2 *__restrict__ w2p = reinterpret_cast<const uint2 *>(w2);      // n_d is even (blocks padded to >= 128 SNPs)
    for (int d = 0; d < n_d / 2; d++) {
        const uint2 w = w2p[(int64_t)d * ncols_pad + sc];
        c += __popc(w.x & (w.x >> 1) & 0x55555555u) + __popc(w.y & (w.y >> 1) & 0x55555555u);
    }
    diag[col0 + sc] += c;
}

// The pre-pass of the IBS / KING counters in ONE pass over a caller block of 2-bit rows (SNPGPU_GENO_PACKED2): the same
// ballot transposition as transpose2_kernel<0>, read straight from the caller's rows (row stride ceil(N/4) bytes, dword
// loads; samples >= N and SNPs >= n_snp become code 3), plus the two things the statistics pass delivered to these kinds:
// the block's "holds missing calls" flag and -- into a per-block buffer, committed by het_commit_kernel once the flag is
// final -- the per-sample het counts of a block without missing calls.  Saves one write and one read of the block
// (repack_stats_kernel + transpose2_kernel: 0.40 ms per 65 536-SNP block at N = 10 000, 8 % of an IBS step).
// Read side (round 3): a workgroup takes 64 SNPs x T2D_SAMPLES samples; every wave instruction reads 256 contiguous bytes of ONE
// row (lane = 16 samples) into an LDS tile, and the waves then pick their 64 x 64 sub-tiles from it.  (Before, a lane read 16
// bytes of its own row -- 64 cache lines per load instruction: 241 us per 65 536-SNP block at N = 10 000 whatever the
// transposition cost.)
constexpr int T2D_SAMPLES = 1024;
__global__ __launch_bounds__(256) void transpose2_direct_kernel(const uint8_t *__restrict__ src, int64_t rb_in, int64_t N,
                                                                int64_t n_snp, int64_t col0, int64_t ncols_pad, int n_d,
                                                                uint32_t *__restrict__ w2, uint32_t *__restrict__ het_blk,
                                                                unsigned long long *__restrict__ d_missing)
{
    __shared__ uint32_t tile[64][T2D_SAMPLES / 16 + 4];    // [SNP][dword of 16 samples], pitch 68 dwords (16-byte aligned rows)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t k0 = (int64_t)blockIdx.y * 64;
    if (k0 >= (int64_t)n_d * 16) return;
    const int64_t sc_wg = (int64_t)blockIdx.x * T2D_SAMPLES;     // panel-relative first sample of the workgroup
    const int64_t sd = col0 + sc_wg + 16 * lane;                 // first sample of this lane's dword (byte offset sd / 4)
    for (int r = wave; r < 64; r += 4) {
        const int64_t k = k0 + r;
        uint32_t v = ~0u;                                        // samples >= N and SNPs >= n_snp: code 3
        if (k < n_snp && sd < N) {
            const uint8_t *row = src + k * rb_in;
            const int64_t b = sd >> 2;
            if (b + 4 <= rb_in) v = *reinterpret_cast<const uint32_t *>(row + b);          // rb_in % 4 == 0 (launcher)
            else
                for (int e = 0; e < 4; e++)
                    if (b + e < rb_in) v = (v & ~(0xFFu << (8 * e))) | ((uint32_t)row[b + e] << (8 * e));
            const int64_t rem = N - sd;                          // samples of this dword that exist
            if (rem < 16) v |= ~0u << (2 * rem);
        }
        tile[r][lane] = v;
    }
    __syncthreads();
    const int64_t n_real = n_snp - k0;               // real SNPs among the workgroup's 64
    for (int cc = wave; cc < T2D_SAMPLES / 64; cc += 4) {
        const int64_t sc0 = sc_wg + 64 * cc;
        if (sc0 >= ncols_pad) break;
        const uint4 q = *reinterpret_cast<const uint4 *>(&tile[lane][4 * cc]);
        uint32_t w[4] = {q.x, q.y, q.z, q.w};
        transpose_2bit_64x64(w, lane);               // lane = sample now: w = the codes of the 64 SNPs
        const int64_t sc = sc0 + lane;
        const int d0 = (int)(k0 >> 4);
        {
            uint2 *__restrict__ w2p = reinterpret_cast<uint2 *>(w2);          // row pairs side by side, as transpose2_kernel
            w2p[(int64_t)(d0 >> 1) * ncols_pad + sc] = make_uint2(w[0], w[1]);
            w2p[(int64_t)((d0 >> 1) + 1) * ncols_pad + sc] = make_uint2(w[2], w[3]);
        }
        // a missing call = code 3 of a real sample at a real SNP (codes of SNPs >= n_snp are padding)
        uint32_t any3 = 0, c = 0, t2 = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            uint32_t m3 = w[t] & (w[t] >> 1) & 0x55555555u;
            const int64_t r = n_real - 16 * t;
            if (r <= 0) m3 = 0u;
            else if (r < 16) m3 &= (1u << (2 * r)) - 1u;
            any3 |= m3;
            c += (uint32_t)__popc(w[t] & ~(w[t] >> 1) & 0x55555555u);
            t2 += (uint32_t)__popc(~w[t] & (w[t] >> 1) & 0x55555555u);
        }
        if (col0 + sc >= N) any3 = 0u;
        if (__ballot(any3 != 0u) && lane == 0) *d_missing = 1ull;       // only ever tested against zero
        if (het_blk) {
            if (c) atomicAdd(het_blk + sc, c);
            if (t2) atomicAdd(het_blk + ncols_pad + sc, t2);
        }
    }
}

// het[j] += het_blk[j] if the block held no missing call (the binary pair kernel took it); het_blk is cleared either way
__global__ __launch_bounds__(256) void het_commit_kernel(uint32_t *__restrict__ het, uint32_t *__restrict__ het_blk,
                                                         int64_t ncols_pad, const unsigned long long *__restrict__ d_missing)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= 2 * ncols_pad) return;                   // #het, then #(g == 2)
    const uint32_t v = het_blk[j];
    if (v) {
        if (*d_missing == 0ull) het[j] += v;
        het_blk[j] = 0u;
    }
}

int launch_transpose2_direct(hipStream_t st, const uint8_t *src, int64_t n_samp, int64_t n_snp, int64_t col0,
                             int64_t ncols_pad, int n_d, uint32_t *w2, uint32_t *het, uint32_t *het_blk,
                             unsigned long long *d_missing)
{
    const int64_t rb_in = (n_samp + 3) / 4;
    dim3 grid((unsigned)((ncols_pad + T2D_SAMPLES - 1) / T2D_SAMPLES), (unsigned)((n_d + 3) / 4));      // n_d * 16 SNPs in groups of 64
    hipLaunchKernelGGL(transpose2_direct_kernel, grid, dim3(256), 0, st, src, rb_in, n_samp, n_snp, col0, ncols_pad, n_d, w2,
                       het ? het_blk : nullptr, d_missing);
    if (het)
        hipLaunchKernelGGL(het_commit_kernel, dim3((unsigned)((2 * ncols_pad + 255) / 256)), dim3(256), 0, st, het, het_blk, ncols_pad,
                           d_missing);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_transpose2(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t col0,
                      int64_t ncols_pad, int n_d, uint32_t *w2, uint32_t *het, const unsigned long long *d_missing, bool classic)
{
    dim3 grid((unsigned)((ncols_pad + TR_SAMPLES - 1) / TR_SAMPLES), (unsigned)((n_d + 3) / 4));
    hipLaunchKernelGGL(transpose2_kernel<0>, grid, dim3(256), 0, st, packed, RB, n_snp, col0, ncols_pad, n_d, w2,
                       (int64_t)0, (const int32_t *)nullptr, (const int32_t *)nullptr, d_missing, het, classic ? 1 : 0);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_transpose2_missmask(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t n_samp,
                               const int32_t *sum, const int32_t *num, int64_t col0, int64_t ncols_pad, int n_d,
                               uint32_t *w2, uint32_t *diag, const unsigned long long *d_skip_if_zero)
{
    dim3 grid((unsigned)((ncols_pad + TR_SAMPLES - 1) / TR_SAMPLES), (unsigned)((n_d + 3) / 4));
    hipLaunchKernelGGL(transpose2_kernel<1>, grid, dim3(256), 0, st, packed, RB, n_snp, col0, ncols_pad, n_d, w2, n_samp,
                       sum, num, d_skip_if_zero, (uint32_t *)nullptr, 0);
    hipLaunchKernelGGL(miss_diag2_kernel, dim3((unsigned)((ncols_pad + 255) / 256)), dim3(256), 0, st, w2, n_d, ncols_pad,
                       col0, diag, d_skip_if_zero);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// GCTA denominators, sparse form (round 4).  The both-missing counts M(i, j) = #{polymorphic SNPs where i AND j are missing}
// are a dense N^2 B contraction for the int8 kernel (81 ms per 32 768-SNP block at N = 100 000, 15 % of the step) whatever the
// missing rate f -- but only f^2 of its products are non-zero.  missmask256_kernel writes, per SNP and group of 256 samples, the
// 256-bit set of samples with a missing call (SNP-major 2-bit rows in, MM[group][snp][8 dwords] out; monomorphic / all-missing
// SNPs, which GCTA does not count -- src/genPCA.cpp:1206 --, and the sample padding give empty sets); pair_sparse_miss_kernel
// (kernels_pair.hip) walks a 256 x 256 tile's two lists of sets and counts the pairs in LDS.
__global__ __launch_bounds__(256) void missmask256_kernel(const uint8_t *__restrict__ packed, int64_t RB, int64_t n_snp,
                                                          int64_t N, const int32_t *__restrict__ sum, const int32_t *__restrict__ num,
                                                          int64_t col0, int n_groups, int64_t snp_stride, uint4 *__restrict__ mm,
                                                          const unsigned long long *__restrict__ d_run)
{
    if (*d_run == 0ull) return;
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int G = blockIdx.y;
    if (k >= snp_stride || G >= n_groups) return;
    uint32_t out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (k < n_snp) {
        const int s = sum[k], c = num[k];
        if (0 < s && s < 2 * c) {                                         // genPCA.cpp:1206
            const int64_t s0 = col0 + (int64_t)G * 256;                   // first sample of the group (col0 is a multiple of 256)
            const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(packed + k * RB + (s0 >> 2));   // RB is a multiple of 64
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 v = src[q];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    uint32_t x = w[t] & (w[t] >> 1) & 0x55555555u;        // code 3 -> bit 2 j
                    x = (x | (x >> 1)) & 0x33333333u;
                    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
                    x = (x | (x >> 4)) & 0x00FF00FFu;
                    x = (x | (x >> 8)) & 0x0000FFFFu;                     // 16 samples -> 16 bits
                    out[2 * q + (t >> 1)] |= x << (16 * (t & 1));
                }
            }
            const int64_t left = N - s0;                                  // samples of this group that exist (padding is code 3)
            if (left < 256)
#pragma unroll
                for (int d = 0; d < 8; d++) {
                    const int64_t r = left - 32 * d;
                    if (r <= 0) out[d] = 0u;
                    else if (r < 32) out[d] &= (1u << r) - 1u;
                }
        }
    }
    uint4 *dst = mm + ((int64_t)G * snp_stride + k) * 2;
    dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
    dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
}

// which form of the both-missing contraction takes this block: flags[0] = sparse (0 < missing calls <= max_cells), flags[1] =
// dense int8 product (more missing calls than that); both 0 for a block without missing calls.  The block's number of missing
// calls = sum over its SNPs of N - num[k] (d_missing is only a flag); one workgroup, summed in a fixed order.
__global__ __launch_bounds__(256) void miss_route_kernel(const int32_t *__restrict__ num, int64_t n_snp, int64_t N,
                                                         unsigned long long max_cells, unsigned long long *__restrict__ flags)
{
    __shared__ unsigned long long part[256];
    unsigned long long m = 0;
    for (int64_t k = threadIdx.x; k < n_snp; k += 256) m += (unsigned long long)(N - num[k]);
    part[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0];
        flags[0] = (t != 0ull && t <= max_cells) ? 1ull : 0ull;
        flags[1] = (t > max_cells) ? 1ull : 0ull;
    }
}

int launch_missmask256(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t N, const int32_t *sum,
                       const int32_t *num, int64_t col0, int n_groups, int64_t snp_stride, uint4 *mm,
                       const unsigned long long *d_missing, unsigned long long max_cells, unsigned long long *flags)
{
    (void)d_missing;
    hipLaunchKernelGGL(miss_route_kernel, dim3(1), dim3(256), 0, st, num, n_snp, N, max_cells, flags);
    if (n_snp > 0 && n_groups > 0)
        hipLaunchKernelGGL(missmask256_kernel, dim3((unsigned)((snp_stride + 255) / 256), (unsigned)n_groups), dim3(256), 0, st, packed, RB,
                           n_snp, N, sum, num, col0, n_groups, snp_stride, mm, flags);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

constexpr int EIGMIX_SAMPLES_CHUNK = 128;    // words (of 8 SNPs) per thread of eigmix_samples_kernel
// per-sample sums of EIGMIX over one block (pair-coded words, see transpose8): number of
// heterozygous calls (DiagAdjVal, genEIGMIX.cpp:125-128) and sum of 4p(1-p) over the SNPs where the
// sample is missing (row/column totals of the missing-union denominator, :129-136)
__global__ __launch_bounds__(256) void eigmix_samples_kernel(const uint32_t *__restrict__ w8, int n_d,
                                                             int64_t ncols_pad, int64_t col0,
                                                             const double *__restrict__ dvals,
                                                             uint32_t *__restrict__ het, double *__restrict__ dmiss,
                                                             double *__restrict__ dsq,
                                                             const unsigned long long *__restrict__ d_wide16)
{
    const int64_t sc = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (sc >= ncols_pad) return;
    const int sh = (d_wide16 && *d_wide16 == 0ull) ? 4 : 3;     // same rule as transpose8_kernel
    uint32_t h = 0;
    double dm = 0, sq = 0;
    // blockIdx.y: a chunk of EIGMIX_SAMPLES_CHUNK words (one thread per sample over the whole block left a 10 000-sample
    // panel with 157 waves for 65 536 SNPs: 11 ms of a 18 ms step); the chunk sums are added atomically
    const int d_lo = blockIdx.y * EIGMIX_SAMPLES_CHUNK;
    const int d_hi = (d_lo + EIGMIX_SAMPLES_CHUNK < n_d) ? (d_lo + EIGMIX_SAMPLES_CHUNK) : n_d;
    for (int d = d_lo; d < d_hi; d++) {
        const uint32_t w = w8[(int64_t)d * ncols_pad + sc];
#pragma unroll
        for (int t = 0; t < 8; t++) {       // byte p = (8 or 16) * (c0 + 4*c1)
            const uint32_t idx = ((w >> (8 * (t >> 1))) & 0xFFu) >> sh;
            const uint32_t code = (t & 1) ? (idx >> 2) : (idx & 3u);
            const int k = 8 * d + t;
            h += (code == 1u);
            if (code == 3u) dm += dvals[2 * k];
            else { const double z = (double)code - dvals[2 * k + 1]; sq += z * z; }
        }
    }
    if (h) atomicAdd(het + col0 + sc, h);
    if (dm != 0.0) unsafeAtomicAdd(dmiss + col0 + sc, dm);
    unsafeAtomicAdd(dsq + col0 + sc, sq);     // fp64 diagonal numerator: (diag - #het) cancels to ~2 % of its terms
}

int launch_eigmix_samples(hipStream_t st, const uint32_t *w8, int n_d, int64_t ncols_pad, int64_t col0,
                          const double *dvals, uint32_t *het, double *dmiss, double *dsq,
                          const unsigned long long *d_wide16)
{
    if (n_d <= 0) return 0;
    hipLaunchKernelGGL(eigmix_samples_kernel,
                       dim3((unsigned)((ncols_pad + 255) / 256), (unsigned)((n_d + EIGMIX_SAMPLES_CHUNK - 1) / EIGMIX_SAMPLES_CHUNK)),
                       dim3(256), 0, st, w8, n_d,
                       ncols_pad, col0, dvals, het, dmiss, dsq, d_wide16);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_bitplanes4(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t n_samp,
                      int64_t col0, int64_t ncols_pad, int64_t rows_pad, int KW, uint4 *rowp, uint4 *colp)
{
    dim3 grid((unsigned)(ncols_pad / 64), (unsigned)((KW / 2 + 3) / 4));
    hipLaunchKernelGGL(bitplanes_kernel<0>, grid, dim3(256), 0, st, packed, RB, n_snp, n_samp,
                       (const int32_t *)nullptr, (const int32_t *)nullptr, col0, ncols_pad, rows_pad, KW,
                       (void *)rowp, (void *)colp, (const unsigned long long *)nullptr);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_bitplanes_miss(hipStream_t st, const uint8_t *packed, int64_t RB, int64_t n_snp, int64_t n_samp,
                          const int32_t *sum, const int32_t *num, int64_t col0, int64_t ncols_pad,
                          int64_t rows_pad, int KW, uint2 *rowp, uint2 *colp,
                          const unsigned long long *d_missing_cells)
{
    dim3 grid((unsigned)(ncols_pad / 64), (unsigned)((KW / 2 + 3) / 4));
    hipLaunchKernelGGL(bitplanes_kernel<1>, grid, dim3(256), 0, st, packed, RB, n_snp, n_samp, sum, num, col0,
                       ncols_pad, rows_pad, KW, (void *)rowp, (void *)colp, d_missing_cells);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// Counter-based synthetic genotypes (SURVEY.md 8(d) generator; bench / test utility, no reference counterpart):
// every cell (snp, sample) is a pure integer function of (seed, snp, sample), so the same block can be produced
// on any GPU and re-computed for a handful of samples on the CPU (oracle/synth.py: synth_hash_*) without I/O.
//   mix32        = the "lowbias32" integer finaliser
//   ks           = mix32(seed ^ mix32(snp + 0x9E3779B9))                  per-SNP key
//   t (16 bit)   = allele-frequency threshold from mix32(ks ^ 0xA5A5A5A5) (spectrum 0: p ~ U(0.05, 0.95);
//                  1: p = u^3 / 2 "rare variants"; 2: p ~ U(0.01, 0.5))
//   h            = mix32(ks ^ (sample * 0x9E3779B1));  g = [h & 0xFFFF < t] + [h >> 16 < t]
//   missing      : mix32(h ^ 0x68E31DA4) < floor(missing * 2^32)
//   special != 0 : SNPs with snp % 997 == 3 / 5 / 7 are all 0 / all 2 / all missing (edge cases)
// Round 4: two spectra with STRUCTURE (accuracy evidence beyond independent SNPs and unrelated samples):
//   spectrum 3   : three sub-populations (sample % 3), Fst ~ 0.1: ancestral p ~ U(0.05, 0.95), population threshold
//                  t_k = t + z_k isqrt(t (65536 - t) / 10) / 148 with z_k = (sum of the four bytes of a per-(SNP, k) hash) - 510
//                  (~N(0, 148^2)), clamped to [655, 64880] -- large off-diagonal entries within and between populations;
//   spectrum 4   : linkage disequilibrium: LD blocks of 48 consecutive SNPs; in each block every sample copies its two
//                  haplotypes from 6 founder haplotypes (founder pair = the 16-bit halves of a per-(block, sample) hash, mod 6), founder f carries the
//                  allele of a SNP iff (mix32(ks ^ (f * 0x85EBCA6B + 0x1B873593)) & 0xFFFF) < t; each haplotype's allele is drawn
//                  independently instead (as in spectrum 0) with probability 2 % (16-bit halves of mix32(h ^ 0x3C6EF372) < 1311).
//                  Consecutive SNPs are strongly correlated, so the products of a pair do not form a random walk within a block.
__device__ __forceinline__ uint32_t synth_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void synth_block_kernel(uint8_t *__restrict__ dst, int64_t N, int64_t rb,
                                                          int64_t snp_begin, uint32_t seed, uint32_t miss32,
                                                          int spectrum, int special)
{
    const int64_t snp = snp_begin + blockIdx.y;
    const uint32_t ks = synth_mix32(seed ^ synth_mix32((uint32_t)snp + 0x9E3779B9u));
    const uint32_t u = synth_mix32(ks ^ 0xA5A5A5A5u) >> 16;                 // 16-bit uniform
    uint32_t t;
    if (spectrum == 1) t = (uint32_t)(((uint64_t)u * u * u) >> 33);          // p = (u / 2^16)^3 / 2
    else if (spectrum == 2) t = 655u + ((u * 32113u) >> 16);                 // p ~ U(0.01, 0.5)
    else t = 3277u + ((u * 58982u) >> 16);                                   // p ~ U(0.05, 0.95)
    uint32_t tk[3] = {t, t, t};
    if (spectrum == 3) {
        uint32_t x = (t * (65536u - t)) / 10u, r = 0;                        // isqrt, bit by bit
        for (uint32_t bit = 1u << 15; bit; bit >>= 1) { const uint32_t c = r | bit; if (c * c <= x) r = c; }
        for (int k = 0; k < 3; k++) {
            const uint32_t hk = synth_mix32(ks ^ (0x0051ED27u + (uint32_t)k * 0x01234567u));
            const int zi = (int)((hk & 0xFFu) + ((hk >> 8) & 0xFFu) + ((hk >> 16) & 0xFFu) + (hk >> 24)) - 510;
            const long long q = ((long long)zi * (long long)r + 148ll * 16777216ll) / 148ll - 16777216ll;   // floor division
            long long v = (long long)t + q;
            tk[k] = (uint32_t)(v < 655 ? 655 : v > 64880 ? 64880 : v);
        }
    }
    uint32_t founders = 0, kb = 0;
    if (spectrum == 4) {
        for (uint32_t f = 0; f < 6; f++)
            founders |= (uint32_t)((synth_mix32(ks ^ (f * 0x85EBCA6Bu + 0x1B873593u)) & 0xFFFFu) < t) << f;
        kb = synth_mix32(seed ^ synth_mix32((uint32_t)(snp / 48) + 0x7F4A7C15u));
    }
    int force = -1;
    if (special) { const int m = (int)(snp % 997); force = (m == 3) ? 0 : (m == 5) ? 2 : (m == 7) ? 3 : -1; }
    uint8_t *__restrict__ row = dst + (int64_t)blockIdx.y * rb;
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < rb; b += (int64_t)gridDim.x * 256) {
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t s = 4 * b + k;
            uint32_t g = 3u;
            if (s < N) {
                const uint32_t h = synth_mix32(ks ^ ((uint32_t)s * 0x9E3779B1u));
                if (spectrum == 3) {
                    const uint32_t tt = tk[s % 3];
                    g = ((h & 0xFFFFu) < tt) + ((h >> 16) < tt);
                } else if (spectrum == 4) {
                    const uint32_t fb = synth_mix32(kb ^ ((uint32_t)s * 0x9E3779B1u)), nz = synth_mix32(h ^ 0x3C6EF372u);
                    const uint32_t a1 = ((nz & 0xFFFFu) < 1311u) ? ((h & 0xFFFFu) < t) : ((founders >> ((fb & 0xFFFFu) % 6u)) & 1u);
                    const uint32_t a2 = ((nz >> 16) < 1311u) ? ((h >> 16) < t) : ((founders >> ((fb >> 16) % 6u)) & 1u);
                    g = a1 + a2;
                } else
                    g = ((h & 0xFFFFu) < t) + ((h >> 16) < t);
                if (miss32 && synth_mix32(h ^ 0x68E31DA4u) < miss32) g = 3u;
                if (force >= 0) g = (uint32_t)force;
            }
            out |= g << (2 * k);
        }
        row[b] = (uint8_t)out;
    }
}

int launch_synth_block(hipStream_t st, uint8_t *dst, int64_t n_samp, int64_t snp_begin, int64_t n_snp, uint32_t seed,
                       uint32_t miss32, int spectrum, int special)
{
    if (n_snp <= 0) return 0;
    const int64_t rb = (n_samp + 3) / 4;
    int gx = (int)((rb + 255) / 256);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(synth_block_kernel, dim3((unsigned)gx, (unsigned)n_snp), dim3(256), 0, st, dst, n_samp, rb,
                       snp_begin, seed, miss32, spectrum, special);
    SNPGPU_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace snpgpu
