// Do VALU ops overlap with int8 MFMAs on MI355X?  (tools only)
// Loop body: 4 independent v_mfma_i32_32x32x32_i8 + NV independent VALU ops (v_perm_b32 / v_and)
// that do not feed the MFMAs.  Reports SIMD cycles per loop iteration for 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NV, int KIND>
__global__ __launch_bounds__(256) void k(int iters, int *out, uint32_t seed)
{
    v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {3, 2, 1, (int)threadIdx.x};
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = seed + threadIdx.x * (i + 1);
    for (int it = 0; it < iters; it++) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            if (KIND == 0) x[v & 7] = __builtin_amdgcn_perm(0u, 0x00FF0001u, x[v & 7]);
            else asm volatile("v_and_b32 %0, 0x3030303, %0" : "+v"(x[v & 7]));
        }
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    int s = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r] + c3[r];
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int KIND> void go(int *out)
{
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;          // 4 waves per block -> wps waves per SIMD
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL((k<NV, KIND>), dim3(blocks), dim3(256), 0, 0, 100, out, 1u);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k<NV, KIND>), dim3(blocks), dim3(256), 0, 0, iters, out, 1u);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        const double cyc = ms * 1e-3 * 2.4e9 / iters / wps;      // SIMD cycles per wave-iteration
        printf("%s NV=%2d wps=%d  %7.1f cycles per (4 MFMA + NV VALU)\n", KIND ? "v_and " : "v_perm", NV, wps, cyc);
    }
}

// same question for the fp32 MFMA of the SYRK kernel: 4 x v_mfma_f32_32x32x2_f32 (64 cycles each) + NV VALU
// ops (KIND 0: v_perm, 1: v_add_u32_sdwa-like add, 2: ds_read_b64 from a lane-dependent address + add)
typedef float v16f __attribute__((ext_vector_type(16)));
template <int NV, int KIND>
__global__ __launch_bounds__(256) void kf(int iters, float *out, uint32_t seed)
{
    __shared__ float2 tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) tab[i] = make_float2((float)i, 1.f);
    __syncthreads();
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float a = (float)threadIdx.x, b = 1.5f;
    uint32_t x[8];
    float2 acc2 = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = seed + threadIdx.x * (i + 1);
    for (int it = 0; it < iters; it++) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            if (KIND == 0) x[v & 7] = __builtin_amdgcn_perm(0u, 0x00FF0001u, x[v & 7]);
            else if (KIND == 1) asm volatile("v_add_u32 %0, 0x3030303, %0" : "+v"(x[v & 7]));
            else {
                const float2 t = tab[(x[v & 7] + it) & 2047];
                acc2.x += t.x; x[v & 7] += 8;
            }
        }
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    float s = acc2.x;
#pragma unroll
    for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r] + c3[r];
#pragma unroll
    for (int i = 0; i < 8; i++) s += (float)x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int KIND> void gof(float *out)
{
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL((kf<NV, KIND>), dim3(blocks), dim3(256), 0, 0, 100, out, 1u);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((kf<NV, KIND>), dim3(blocks), dim3(256), 0, 0, iters, out, 1u);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        const double cyc = ms * 1e-3 * 2.4e9 / iters / wps;
        printf("f32 MFMA + %s NV=%2d wps=%d  %7.1f cycles per (4 MFMA + NV ops)\n",
               KIND == 0 ? "v_perm" : KIND == 1 ? "v_add " : "ds_read_b64+2valu", NV, wps, cyc);
    }
}

int main()
{
    {
        float *outf; (void)hipMalloc(&outf, 1024 * 256 * 4);
        gof<0, 0>(outf); gof<8, 0>(outf); gof<16, 0>(outf); gof<32, 0>(outf); gof<48, 0>(outf); gof<64, 0>(outf);
        gof<16, 1>(outf); gof<32, 1>(outf);
        gof<4, 2>(outf); gof<8, 2>(outf); gof<16, 2>(outf);
    }
    int *out; (void)hipMalloc(&out, 1024 * 256 * 4);
    go<0, 0>(out); go<8, 0>(out); go<16, 0>(out); go<24, 0>(out); go<32, 0>(out); go<48, 0>(out);
    go<16, 1>(out); go<32, 1>(out);
    return 0;
}
