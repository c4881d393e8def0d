// Feasibility of the pair counters on the fp4 matrix path (v_mfma_scale_f32_32x32x64_f8f6f4, E2M1 operands, unit scales):
// operands {0, 1, 2, -1} are exact in fp4, sums < 2^24 are exact in the fp32 accumulators.  (1) correctness of a 32x32x64
// product of random {0,1,2} / {-1,0,1} operands against the CPU; the k order inside a lane does not matter as long as A
// and B use the same one.  (2) rate of a register-only stream, to compare with v_mfma_i32_32x32x32_i8 (tools only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v16f mfma_fp4(v8i a, v8i b, v16f c)
{
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 127, 0, 127);
}

__global__ void check_kernel(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, float *__restrict__ d)
{
    const int l = threadIdx.x;
    v8i A = {0, 0, 0, 0, 0, 0, 0, 0}, B = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < 4; t++) { A[t] = (int)a[l * 4 + t]; B[t] = (int)b[l * 4 + t]; }
    v16f c;
    for (int r = 0; r < 16; r++) c[r] = 0.f;
    c = mfma_fp4(A, B, c);
    for (int r = 0; r < 16; r++) d[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

template <int FP4>
__global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, uint32_t seed)
{
    v8i A[4], B[4];
    for (int i = 0; i < 4; i++)
        for (int t = 0; t < 8; t++) {
            // nibbles 0 / 2 / 4 = 0, 1, 2 in E2M1; as int8 bytes the same words read 0x20 0x42 ... : only the rate matters there
            const uint32_t x = (threadIdx.x * 2654435761u + i * 97u + t * 13u + seed) * 2246822519u;
            A[i][t] = (int)((x & 0x22222222u) | ((x >> 1) & 0x44444444u & ~((x & 0x22222222u) << 1)));
            B[i][t] = (int)(((x >> 3) & 0x22222222u));
        }
    v16f c[4][4];
    v16i ci[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            for (int r = 0; r < 16; r++) { c[i][j][r] = 0.f; ci[i][j][r] = 0; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (FP4) c[i][j] = mfma_fp4(A[i], B[j], c[i][j]);
                else {
                    v4i a4 = {A[i][0], A[i][1], A[i][2], A[i][3]}, b4 = {B[j][0], B[j][1], B[j][2], B[j][3]};
                    ci[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, ci[i][j], 0, 0, 0);
                }
            }
    }
    float s = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            for (int r = 0; r < 16; r++) s += FP4 ? c[i][j][r] : (float)ci[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    // (1) correctness
    std::vector<int> av(32 * 64), bv(32 * 64);
    srand(7);
    for (auto &x : av) x = rand() % 3;             // {0, 1, 2}
    for (auto &x : bv) x = rand() % 3 - 1;         // {-1, 0, 1}
    auto enc = [](int v) -> uint32_t { return v == 0 ? 0u : v == 1 ? 2u : v == 2 ? 4u : 0xAu; };   // E2M1: 1.0 = 0010, 2.0 = 0100, -1.0 = 1010
    std::vector<uint32_t> ah(64 * 4, 0), bh(64 * 4, 0);
    for (int l = 0; l < 64; l++)
        for (int t = 0; t < 32; t++) {
            const int k = 32 * (l >> 5) + t;
            ah[l * 4 + t / 8] |= enc(av[(l & 31) * 64 + k]) << (4 * (t % 8));
            bh[l * 4 + t / 8] |= enc(bv[(l & 31) * 64 + k]) << (4 * (t % 8));
        }
    uint32_t *da, *db; float *dd;
    hipMalloc(&da, ah.size() * 4); hipMalloc(&db, bh.size() * 4); hipMalloc(&dd, 32 * 32 * 4);
    hipMemcpy(da, ah.data(), ah.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, bh.data(), bh.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
    std::vector<float> dh(32 * 32);
    hipMemcpy(dh.data(), dd, dh.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            int ref = 0;
            for (int k = 0; k < 64; k++) ref += av[i * 64 + k] * bv[j * 64 + k];
            if ((float)ref != dh[i * 32 + j]) { if (bad < 5) printf("mismatch (%d,%d): %g vs %d\n", i, j, dh[i * 32 + j], ref); bad++; }
        }
    printf("fp4 32x32x64 product of {0,1,2} x {-1,0,1}: %d mismatches of 1024\n", bad);
    // (2) rate
    float *out;
    hipMalloc(&out, 256 * 4 * 256 * 8 * sizeof(float));
    for (int fp4 = 1; fp4 >= 0; fp4--) {
        const int iters = 20000, blocks = 256 * 2;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0, 0);
            if (fp4) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
            else hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double ops = (double)blocks * 4 * iters * 16 * (fp4 ? 2.0 * 32 * 32 * 64 : 2.0 * 32 * 32 * 32);
            printf("%s: %.2f ms, %.0f TOP/s\n", fp4 ? "fp4 32x32x64" : "i8 32x32x32", ms, ops / ms / 1e9);
        }
    }
    return 0;
}
