// Ablation of the bit-plane pair kernel (tools only): how far is the compiled VALU stream from the
// instruction-throughput bound, and what do the loads cost?  MODE 0 = IBS (7 ops), 1 = KING (11 ops).
//   LOADS 0: operands loaded once (pure VALU stream)   1: real scalar+vector loads every word
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define BITOP3_A_OR_BC 0xF8
#define BITOP3_AXB_AND_C 0x28
template <int MODE> __device__ __forceinline__ void run(const uint4 &r, const uint4 &c, uint32_t *cnt)
{
    if (MODE == 0) {
        const uint32_t t0 = r.x & c.x;
        cnt[0] += __popc(t0);
        cnt[1] += __popc(__builtin_amdgcn_bitop3_b32(c.y, r.y, t0, BITOP3_AXB_AND_C));
        cnt[2] += __popc(__builtin_amdgcn_bitop3_b32(r.z & c.w, r.w, c.z, BITOP3_A_OR_BC));
    } else {
        cnt[0] += __popc(r.x & c.x);
        const uint32_t a = r.y & c.x, b = r.x & c.y;
        cnt[3] += __popc(a); cnt[4] += __popc(b); cnt[1] += __popc(a ^ b);
        cnt[2] += __popc(__builtin_amdgcn_bitop3_b32(r.z & c.w, r.w, c.z, BITOP3_A_OR_BC));
    }
}
template <int MODE, int LOADS, int A, int BC>
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ rowp, const uint4 *__restrict__ colp, int KW,
                                         int64_t ncols, uint32_t *__restrict__ out)
{
    constexpr int C = MODE == 0 ? 3 : 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // big = realistic footprint: 10240 samples, supertile-ish order (8x8 tiles of 32x(64*BC))
    int tr, tc;
    if (KW < 0) { tr = 0; tc = 0; }
    else if (ncols >= 10240) {
        const int per = 64, st = blockIdx.x / per, w = blockIdx.x % per;
        const int nsc = (int)(ncols / (64 * BC)) / 8;
        tr = (st / nsc) * 8 + w / 8; tc = (st % nsc) * 8 + w % 8;
        tr %= (10240 / (4 * A));
    } else { tr = blockIdx.x % 64; tc = (blockIdx.x / 64) % 32; }
    const uint4 *__restrict__ rp = rowp + (int64_t)(tr * 4 + wave) * KW * A;
    const uint4 *__restrict__ cp = colp + tc * (64 * BC) + lane;
    uint32_t cnt[A][BC][C];
    for (int a = 0; a < A; a++) for (int b = 0; b < BC; b++) for (int c = 0; c < C; c++) cnt[a][b][c] = 0;
    uint4 r[A], c[BC];
    for (int a = 0; a < A; a++) r[a] = rp[a];
    for (int b = 0; b < BC; b++) c[b] = cp[b * 64];
    for (int kw = 0; kw < KW; kw++) {
        if (LOADS) {
#pragma unroll
            for (int a = 0; a < A; a++) r[a] = rp[(int64_t)kw * A + a];
#pragma unroll
            for (int b = 0; b < BC; b++) c[b] = cp[(int64_t)kw * ncols + b * 64];
        } else {
#pragma unroll
            for (int b = 0; b < BC; b++) { c[b].x += kw; c[b].z ^= kw; }   // keep the stream data dependent on kw
        }
#pragma unroll
        for (int a = 0; a < A; a++)
#pragma unroll
            for (int b = 0; b < BC; b++) run<MODE>(r[a], c[b], cnt[a][b]);
    }
    uint32_t s = 0;
    for (int a = 0; a < A; a++) for (int b = 0; b < BC; b++) for (int c2 = 0; c2 < C; c2++) s += cnt[a][b][c2];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int LOADS, int A, int BC>
void go(const char *name, const uint4 *rowp, const uint4 *colp, int KW, int64_t ncols, uint32_t *out)
{
    const int blocks = 64 * 32 * 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, LOADS, A, BC>), dim3(blocks), dim3(256), 0, 0, rowp, colp, KW, ncols, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, LOADS, A, BC>), dim3(blocks), dim3(256), 0, 0, rowp, colp, KW, ncols, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double pw = (double)blocks * 256 * A * BC * KW;     // lane pair-words
    const double simple = MODE == 0 ? 4 : 6, pop = MODE == 0 ? 3 : 5;
    const double ideal_ms = pw * (simple / 63.8e12 + pop / 37.2e12) * 1e3;   // measured instruction rates
    printf("%-34s A=%d BC=%d  %7.3f ms  %6.2f Tpair-word-lanes/s  (%.0f%% of the instruction-throughput bound)\n", name, A, BC,
           ms, pw / ms / 1e9, ideal_ms / ms * 100);
}
int main()
{
    const int KW = 512; const int64_t ncols = (getenv("BIG") ? 10240 : 32 * 256);
    std::vector<uint32_t> h((size_t)KW * ncols * 4);
    for (auto &x : h) x = (uint32_t)rand() * 2654435761u;
    uint4 *rowp, *colp; uint32_t *out;
    hipMalloc(&rowp, h.size() * 4); hipMalloc(&colp, h.size() * 4); hipMalloc(&out, 64 * 32 * 4 * 256 * 4);
    hipMemcpy(rowp, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(colp, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    go<0, 0, 8, 2>("IBS  no loads", rowp, colp, KW, ncols, out);
    go<0, 1, 8, 2>("IBS  loads", rowp, colp, KW, ncols, out);
    go<0, 0, 8, 4>("IBS  no loads", rowp, colp, KW, ncols, out);
    go<0, 1, 8, 4>("IBS  loads", rowp, colp, KW, ncols, out);
    go<1, 0, 8, 2>("KING no loads", rowp, colp, KW, ncols, out);
    go<1, 1, 8, 2>("KING loads", rowp, colp, KW, ncols, out);
    go<1, 0, 4, 4>("KING no loads", rowp, colp, KW, ncols, out);
    go<1, 1, 4, 4>("KING loads", rowp, colp, KW, ncols, out);
    go<1, 0, 4, 2>("KING no loads", rowp, colp, KW, ncols, out);
    go<1, 1, 4, 2>("KING loads", rowp, colp, KW, ncols, out);
    return 0;
}
