// Diagnostics behind include/snpgpu.h (no reference counterpart): what THIS device's matrix pipe sustains right now.
//
// The SYRK / pair-counter kernels run against the socket power cap, not against the 2.4 GHz peak: a register-only
// MFMA stream with operands shaped like theirs settles at 1.7 - 1.85 GHz, one with zero operands at 2.39 GHz, and the figure moves
// by a few per cent from box to box.  bench.py therefore measures it in the run it reports (roofline.sustained_peak_measured)
// instead of quoting a constant: snpgpu_diag_mfma_rate streams one MFMA instruction from registers -- 8 independent accumulators
// per wave, 2 waves per SIMD, so the pipe never waits -- for `seconds` and returns the rate over the second half of that time.
#include "snpgpu_internal.h"

#include <chrono>
#include <vector>

namespace snpgpu {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));

constexpr int DIAG_NT = 8, DIAG_ITERS = 2048;

__global__ __launch_bounds__(256, 2) void diag_f16_kernel(const h8 *__restrict__ src, float *__restrict__ out)
{
    h8 a[4], b[4];
    for (int i = 0; i < 4; i++) {
        a[i] = src[(threadIdx.x + 256 * i) & 1023];
        b[i] = src[1024 + ((threadIdx.x + 256 * i + 77) & 1023)];
    }
    f16v acc[DIAG_NT];
    for (int t = 0; t < DIAG_NT; t++)
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    for (int it = 0; it < DIAG_ITERS; it++) {
#pragma unroll
        for (int t = 0; t < DIAG_NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t & 3], b[(t >> 1) & 3], acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < DIAG_NT; t++)
        for (int r = 0; r < 16; r++) s += acc[t][r];
    if (s == 1.2345f) out[0] = s;
}

// the same flops through v_mfma_f32_16x16x32_f16 (a quarter of the accumulator registers per instruction, half the cycles)
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void diag_f16_16x16x32_kernel(const h8 *__restrict__ src, float *__restrict__ out)
{
    h8 a[4], b[4];
    for (int i = 0; i < 4; i++) {
        a[i] = src[(threadIdx.x + 256 * i) & 1023];
        b[i] = src[1024 + ((threadIdx.x + 256 * i + 77) & 1023)];
    }
    f4v acc[2 * DIAG_NT];
    for (int t = 0; t < 2 * DIAG_NT; t++)
        for (int r = 0; r < 4; r++) acc[t][r] = 0.f;
    for (int it = 0; it < DIAG_ITERS; it++) {
#pragma unroll
        for (int t = 0; t < 2 * DIAG_NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t & 3], b[(t >> 2) & 3], acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < 2 * DIAG_NT; t++)
        for (int r = 0; r < 4; r++) s += acc[t][r];
    if (s == 1.2345f) out[0] = s;
}

__global__ __launch_bounds__(256, 2) void diag_fp4_kernel(const v8i *__restrict__ src, float *__restrict__ out)
{
    v8i a[4], b[4];
    for (int i = 0; i < 4; i++) {
        a[i] = src[(threadIdx.x + 256 * i) & 1023];
        b[i] = src[1024 + ((threadIdx.x + 256 * i + 77) & 1023)];
    }
    f16v acc[DIAG_NT];
    for (int t = 0; t < DIAG_NT; t++)
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    for (int it = 0; it < DIAG_ITERS; it++) {
#pragma unroll
        for (int t = 0; t < DIAG_NT; t++)
            acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t & 3], b[(t >> 1) & 3], acc[t], 4, 4, 0, 127, 0, 127);
    }
    float s = 0;
    for (int t = 0; t < DIAG_NT; t++)
        for (int r = 0; r < 16; r++) s += acc[t][r];
    if (s == 1.2345f) out[0] = s;
}

__global__ __launch_bounds__(256, 2) void diag_fp4_16x16x128_kernel(const v8i *__restrict__ src, float *__restrict__ out)
{
    v8i a[4], b[4];
    for (int i = 0; i < 4; i++) {
        a[i] = src[(threadIdx.x + 256 * i) & 1023];
        b[i] = src[1024 + ((threadIdx.x + 256 * i + 77) & 1023)];
    }
    f4v acc[2 * DIAG_NT];
    for (int t = 0; t < 2 * DIAG_NT; t++)
        for (int r = 0; r < 4; r++) acc[t][r] = 0.f;
    for (int it = 0; it < DIAG_ITERS; it++) {
#pragma unroll
        for (int t = 0; t < 2 * DIAG_NT; t++)
            acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[t & 3], b[(t >> 2) & 3], acc[t], 4, 4, 0, 127, 0, 127);
    }
    float s = 0;
    for (int t = 0; t < 2 * DIAG_NT; t++)
        for (int r = 0; r < 4; r++) s += acc[t][r];
    if (s == 1.2345f) out[0] = s;
}

uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

}  // namespace
}  // namespace snpgpu

using namespace snpgpu;

extern "C" int snpgpu_diag_mfma_rate(int device, int mode, double seconds, double *tflops, double *implied_mhz)
{
    if (!tflops || !(seconds > 0.0) || seconds > 30.0 || mode < 0 || mode > SNPGPU_DIAG_F16_EXACT_ROW_16X16X32) {
        set_error("snpgpu_diag_mfma_rate: invalid arguments");
        return 1;
    }
    SNPGPU_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    SNPGPU_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        set_error(std::string("snpgpu_diag_mfma_rate: device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only");
        return 1;
    }
    // operand images: entries 0..1023 = a registers, 1024..2047 = b registers (16 bytes for fp16 x 8, 32 bytes for 64 nibbles padded)
    const bool fp4 = mode == SNPGPU_DIAG_FP4 || mode == SNPGPU_DIAG_FP4_16X16X128;
    const size_t entry = fp4 ? 32 : 16;
    std::vector<uint8_t> h(2048 * entry, 0);
    uint32_t s = 7u;
    if (mode == SNPGPU_DIAG_F16_EXACT_ROW || mode == SNPGPU_DIAG_F16_UV || mode == SNPGPU_DIAG_F16_UV_16X16X32 || mode == SNPGPU_DIAG_F16_EXACT_ROW_16X16X32) {
        _Float16 *p = (_Float16 *)h.data();
        for (int e = 0; e < 2048; e++) {
            // single-product kernel (SNPGPU_DIAG_F16_UV): both operands (g - c) x an fp16 factor of the SNP weight, g in {0,1,2},
            // c in {0,1,2}; exact-row kernel: row operand (g - c), column operand a real number y^2 (g - avg) (hi part)
            for (int k = 0; k < 8; k++) {
                const int gg = (int)(lcg(s) % 3u), cc = (int)(lcg(s) % 3u);
                const float g = (float)(gg - cc);
                const float f = 0.75f + (float)(lcg(s) % 1024u) / 1024.0f;
                const bool row = e < 1024;
                float v;
                if (mode != SNPGPU_DIAG_F16_EXACT_ROW && mode != SNPGPU_DIAG_F16_EXACT_ROW_16X16X32) v = g * f;
                else v = row ? g : ((float)(lcg(s) % 2001u) - 1000.0f) / 400.0f;
                p[e * 8 + k] = (_Float16)v;
            }
        }
    } else if (fp4) {
        uint32_t *p = (uint32_t *)h.data();
        for (int e = 0; e < 2048; e++)
            for (int w = 0; w < 4; w++) {        // 32 nibbles used per lane (the upper four dwords of the operand are ignored for fp4)
                uint32_t x = 0;
                for (int k = 0; k < 8; k++) {
                    // e2m1 codes of the counter kernels' value types: a in {0, 1/2, 1} (nibble of g / 2), b in {0, +-1}
                    const uint32_t g = lcg(s) % 3u;
                    const uint32_t code = e < 1024 ? g : (g == 0 ? 0u : g == 1 ? 2u : 0xAu);
                    x |= code << (4 * k);
                }
                p[e * 8 + w] = x;
            }
    }
    void *d_src = nullptr, *d_out = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int rc = 1;
    do {
        if (hipMalloc(&d_src, h.size()) != hipSuccess || hipMalloc(&d_out, 64) != hipSuccess) { set_error("snpgpu_diag_mfma_rate: hipMalloc failed"); break; }
        if (hipMemcpy(d_src, h.data(), h.size(), hipMemcpyHostToDevice) != hipSuccess) { set_error("snpgpu_diag_mfma_rate: copy failed"); break; }
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { set_error("snpgpu_diag_mfma_rate: stream"); break; }
        bool ok = true;
        for (auto &e : ev) ok = ok && hipEventCreate(&e) == hipSuccess;
        if (!ok) { set_error("snpgpu_diag_mfma_rate: events"); break; }
        const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        const int blocks = cus * 2 * 4;       // 2 workgroups of 4 waves per CU resident, 4 rounds per launch (~3 ms)
        auto launch = [&]() {
            if (mode == SNPGPU_DIAG_FP4_16X16X128)
                hipLaunchKernelGGL(diag_fp4_16x16x128_kernel, dim3(blocks), dim3(256), 0, st, (const v8i *)d_src, (float *)d_out);
            else if (fp4) hipLaunchKernelGGL(diag_fp4_kernel, dim3(blocks), dim3(256), 0, st, (const v8i *)d_src, (float *)d_out);
            else if (mode == SNPGPU_DIAG_F16_UV_16X16X32 || mode == SNPGPU_DIAG_F16_EXACT_ROW_16X16X32)
                hipLaunchKernelGGL(diag_f16_16x16x32_kernel, dim3(blocks), dim3(256), 0, st, (const h8 *)d_src, (float *)d_out);
            else hipLaunchKernelGGL(diag_f16_kernel, dim3(blocks), dim3(256), 0, st, (const h8 *)d_src, (float *)d_out);
        };
        const double flop_per_launch = 2.0 * (fp4 ? 32.0 * 32 * 64 : 32.0 * 32 * 16) * DIAG_NT * DIAG_ITERS * 4.0 * blocks;
        // first half: the clock settles under the power cap; second half: timed with HIP events on this stream
        auto run_for = [&](double secs, long &n) {
            const auto t0 = std::chrono::steady_clock::now();
            n = 0;
            do {
                for (int i = 0; i < 8; i++) launch();
                n += 8;
                if (hipStreamSynchronize(st) != hipSuccess) return false;
            } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs);
            return true;
        };
        long n1 = 0, n2 = 0;      // n1: launches of the settling half (not timed)
        if (!run_for(seconds * 0.5, n1)) { set_error("snpgpu_diag_mfma_rate: kernel failed"); break; }
        (void)hipEventRecord(ev[0], st);
        if (!run_for(seconds * 0.5, n2)) { set_error("snpgpu_diag_mfma_rate: kernel failed"); break; }
        (void)hipEventRecord(ev[1], st);
        if (hipEventSynchronize(ev[1]) != hipSuccess) { set_error("snpgpu_diag_mfma_rate: sync failed"); break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[0], ev[1]);
        if (!(ms > 0.f)) { set_error("snpgpu_diag_mfma_rate: no time measured"); break; }
        const double rate = flop_per_launch * (double)n2 / ((double)ms * 1e-3) / 1e12;
        *tflops = rate;
        // the stream is issue-bound on the matrix pipe: rate / (flop per clock of the whole device) = the shader clock it ran at
        if (implied_mhz) *implied_mhz = rate * 1e12 / ((double)cus * 4.0 * (fp4 ? 4096.0 : 1024.0)) / 1e6;
        rc = 0;
    } while (0);
    for (auto &e : ev) if (e) (void)hipEventDestroy(e);
    if (st) (void)hipStreamDestroy(st);
    if (d_src) (void)hipFree(d_src);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

extern "C" int snpgpu_diag_device_pci(int device, char *buf, int len)
{
    if (!buf || len < 16) { set_error("snpgpu_diag_device_pci: buffer of at least 16 bytes needed"); return 1; }
    SNPGPU_HIP_CHECK(hipDeviceGetPCIBusId(buf, len, device));
    return 0;
}
