// What the MFMA pipes sustain under the socket power cap (tools only): a register-only stream of
// v_mfma_f32_32x32x16_f16 / v_mfma_i32_32x32x32_i8 on 8 independent accumulators per wave, 2 waves per SIMD,
// operands filled with data shaped like the real kernels' (fp16 standardised genotypes and their lo parts /
// int8 values in {-1,0,1}) or with zeros.  Prints the achieved rate; run tools/clock_watch-style polling of
// rocm-smi beside it for clock and power.      usage: mfma_power_ubench <f16|i8> <real|zero> [seconds]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));

constexpr int NT = 8, ITERS = 4096;

__global__ __launch_bounds__(256, 2) void f16_kernel(const h8 *__restrict__ src, float *__restrict__ out)
{
    h8 a[4], b[4];
    for (int i = 0; i < 4; i++) {
        a[i] = src[(threadIdx.x + 256 * i) & 1023];
        b[i] = src[1024 + ((threadIdx.x + 256 * i + 77) & 1023)];
    }
    f16v acc[NT];
    for (int t = 0; t < NT; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t & 3], b[(t >> 1) & 3], acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < NT; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    if (s == 1.2345f) out[0] = s;
}

__global__ __launch_bounds__(256, 2) void i8_kernel(const i4 *__restrict__ src, int *__restrict__ out)
{
    i4 a[4], b[4];
    for (int i = 0; i < 4; i++) {
        a[i] = src[(threadIdx.x + 256 * i) & 1023];
        b[i] = src[1024 + ((threadIdx.x + 256 * i + 77) & 1023)];
    }
    i16v acc[NT];
    for (int t = 0; t < NT; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t & 3], b[(t >> 1) & 3], acc[t], 0, 0, 0);
    }
    int s = 0;
    for (int t = 0; t < NT; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    if (s == 0x12345677) out[0] = s;
}

int main(int argc, char **argv)
{
    const bool f16 = argc > 1 && !strcmp(argv[1], "f16");
    const bool real = !(argc > 2 && !strcmp(argv[2], "zero"));
    const bool exact = argc > 2 && !strcmp(argv[2], "exact");   // a: g - 1 in {-1,0,1} (f16) / {0,1} (i8), b: as "real"
    const double secs = argc > 3 ? atof(argv[3]) : 4.0;
    std::vector<uint8_t> h(2048 * 16, 0);      // entries 0..1023: a operands, 1024..2047: b operands
    srand(7);
    if (real) {
        if (f16) {
            _Float16 *p = (_Float16 *)h.data();
            for (int i = 0; i < 2048 * 8; i++) {
                // half of the operand registers hold hi parts (|z| up to ~4), half hold lo parts (~2^-11 of that)
                const float z = ((rand() % 2001) - 1000) / 400.0f;
                p[i] = (_Float16)(((i >> 3) & 1) ? z * 4.8e-4f : z);
                // "exact": the a registers (entries tid + 256 i, i.e. all of the first 1024 lanes' reads) hold g - 1
                if (exact && i < 1024 * 8) p[i] = (_Float16)(float)((rand() % 3) - 1);
            }
        } else {
            int8_t *p = (int8_t *)h.data();
            for (int i = 0; i < 2048 * 16; i++) p[i] = exact ? (int8_t)(rand() % 2) : (int8_t)((rand() % 3) - 1);
        }
    }
    void *d_src, *d_out;
    hipMalloc(&d_src, h.size());
    hipMalloc(&d_out, 64);
    hipMemcpy(d_src, h.data(), h.size(), hipMemcpyHostToDevice);
    const int blocks = 256 * 2 * 8;     // 2 workgroups of 4 waves per CU resident, 8 rounds
    auto launch = [&]() {
        if (f16) hipLaunchKernelGGL(f16_kernel, dim3(blocks), dim3(256), 0, 0, (const h8 *)d_src, (float *)d_out);
        else hipLaunchKernelGGL(i8_kernel, dim3(blocks), dim3(256), 0, 0, (const i4 *)d_src, (int *)d_out);
    };
    for (int i = 0; i < 3; i++) launch();
    hipDeviceSynchronize();
    const double ops_per_launch = 2.0 * (f16 ? 32.0 * 32 * 16 : 32.0 * 32 * 32) * NT * ITERS * 4.0 * blocks;
    double t_total = 0; long n = 0;
    auto t0 = std::chrono::steady_clock::now();
    while (t_total < secs) {
        for (int i = 0; i < 10; i++) launch();
        hipDeviceSynchronize();
        n += 10;
        t_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    printf("%s %s: %.1f T%s/s over %.1f s (%ld launches)\n", f16 ? "f16 32x32x16" : "i8 32x32x32", exact ? "exact" : real ? "real" : "zero",
           ops_per_launch * n / t_total / 1e12, f16 ? "FLOP" : "OP", t_total, n);
    return 0;
}
