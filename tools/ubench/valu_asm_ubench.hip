// Pure instruction-stream VALU throughput on MI355X (tools only): 16 independent chains,
// 64 instructions per loop body, 8 waves/SIMD resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define BODY(INS)                                                                                   \
    asm volatile(                                                                                   \
        REP4(INS(%0) INS(%1) INS(%2) INS(%3) INS(%4) INS(%5) INS(%6) INS(%7)                        \
             INS(%8) INS(%9) INS(%10) INS(%11) INS(%12) INS(%13) INS(%14) INS(%15))                 \
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
          "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) \
        : "v"(x), "v"(y))
#define I_AND(r) "v_and_b32 " #r ", %16, " #r "\n"
#define I_XOR(r) "v_xor_b32 " #r ", %16, " #r "\n"
#define I_BCNT(r) "v_bcnt_u32_b32 " #r ", %16, " #r "\n"
#define I_ANDOR(r) "v_and_or_b32 " #r ", " #r ", %16, %17\n"
#define I_BITOP3(r) "v_bitop3_b32 " #r ", " #r ", %16, %17 bitop3:0x6c\n"
#define I_FMA(r) "v_fma_f32 " #r ", " #r ", %16, %17\n"
#define I_ADD(r) "v_add_u32 " #r ", %16, " #r "\n"
// KING per-pair sequence (6 logic + 5 bcnt) and IBS (4 logic + 3 bcnt), operands all VGPR
#define KING_SEQ(t0, t1, c0, c1, c2, c3, c4) \
    "v_and_b32 " #t0 ", %16, " #c0 "\n v_bcnt_u32_b32 " #c0 ", " #t0 ", " #c0 "\n" \
    "v_and_b32 " #t0 ", %17, " #c1 "\n v_and_b32 " #t1 ", %16, " #c2 "\n" \
    "v_bcnt_u32_b32 " #c1 ", " #t0 ", " #c1 "\n v_bcnt_u32_b32 " #c2 ", " #t1 ", " #c2 "\n" \
    "v_xor_b32 " #t0 ", " #t0 ", " #t1 "\n v_bcnt_u32_b32 " #c3 ", " #t0 ", " #c3 "\n" \
    "v_and_b32 " #t0 ", %17, " #c4 "\n v_bitop3_b32 " #t0 ", " #t0 ", %16, %17 bitop3:0xf8\n v_bcnt_u32_b32 " #c4 ", " #t0 ", " #c4 "\n"
#define IBS_SEQ(t0, t1, c0, c1, c2) \
    "v_and_b32 " #t0 ", %16, " #c0 "\n v_bcnt_u32_b32 " #c0 ", " #t0 ", " #c0 "\n" \
    "v_bitop3_b32 " #t1 ", " #t0 ", %16, %17 bitop3:0x28\n v_bcnt_u32_b32 " #c1 ", " #t1 ", " #c1 "\n" \
    "v_and_b32 " #t0 ", %17, " #c2 "\n v_bitop3_b32 " #t0 ", " #t0 ", %16, %17 bitop3:0xf8\n v_bcnt_u32_b32 " #c2 ", " #t0 ", " #c2 "\n"
#define MIXBODY(SEQ2)                                                                               \
    asm volatile(SEQ2                                                                               \
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
          "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) \
        : "v"(x), "v"(y))
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a[16];
    for (int i = 0; i < 16; i++) a[i] = seed * (i + 3) + threadIdx.x;
    uint32_t x = seed * 7 + threadIdx.x, y = seed * 13;
    for (int it = 0; it < iters; it++) {
        if (OP == 0) BODY(I_AND);
        if (OP == 1) BODY(I_XOR);
        if (OP == 2) BODY(I_BCNT);
        if (OP == 3) BODY(I_ANDOR);
        if (OP == 4) BODY(I_BITOP3);
        if (OP == 5) BODY(I_FMA);
        if (OP == 6) BODY(I_ADD);
        // 8 and + 8 bcnt: alternating vs batched (independent registers)
        if (OP == 9) MIXBODY(REP4(
            "v_and_b32 %0, %16, %0\n v_bcnt_u32_b32 %8, %17, %8\n v_and_b32 %1, %16, %1\n v_bcnt_u32_b32 %9, %17, %9\n"
            "v_and_b32 %2, %16, %2\n v_bcnt_u32_b32 %10, %17, %10\n v_and_b32 %3, %16, %3\n v_bcnt_u32_b32 %11, %17, %11\n"
            "v_and_b32 %4, %16, %4\n v_bcnt_u32_b32 %12, %17, %12\n v_and_b32 %5, %16, %5\n v_bcnt_u32_b32 %13, %17, %13\n"
            "v_and_b32 %6, %16, %6\n v_bcnt_u32_b32 %14, %17, %14\n v_and_b32 %7, %16, %7\n v_bcnt_u32_b32 %15, %17, %15\n"));
        if (OP == 10) MIXBODY(REP4(
            "v_and_b32 %0, %16, %0\n v_and_b32 %1, %16, %1\n v_and_b32 %2, %16, %2\n v_and_b32 %3, %16, %3\n"
            "v_and_b32 %4, %16, %4\n v_and_b32 %5, %16, %5\n v_and_b32 %6, %16, %6\n v_and_b32 %7, %16, %7\n"
            "v_bcnt_u32_b32 %8, %17, %8\n v_bcnt_u32_b32 %9, %17, %9\n v_bcnt_u32_b32 %10, %17, %10\n v_bcnt_u32_b32 %11, %17, %11\n"
            "v_bcnt_u32_b32 %12, %17, %12\n v_bcnt_u32_b32 %13, %17, %13\n v_bcnt_u32_b32 %14, %17, %14\n v_bcnt_u32_b32 %15, %17, %15\n"));
        if (OP == 11) MIXBODY(REP4(   // pairs: 2 and, 2 bcnt
            "v_and_b32 %0, %16, %0\n v_and_b32 %1, %16, %1\n v_bcnt_u32_b32 %8, %17, %8\n v_bcnt_u32_b32 %9, %17, %9\n"
            "v_and_b32 %2, %16, %2\n v_and_b32 %3, %16, %3\n v_bcnt_u32_b32 %10, %17, %10\n v_bcnt_u32_b32 %11, %17, %11\n"
            "v_and_b32 %4, %16, %4\n v_and_b32 %5, %16, %5\n v_bcnt_u32_b32 %12, %17, %12\n v_bcnt_u32_b32 %13, %17, %13\n"
            "v_and_b32 %6, %16, %6\n v_and_b32 %7, %16, %7\n v_bcnt_u32_b32 %14, %17, %14\n v_bcnt_u32_b32 %15, %17, %15\n"));
        if (OP == 7) MIXBODY(REP4(KING_SEQ(%0, %1, %2, %3, %4, %5, %6) KING_SEQ(%7, %8, %9, %10, %11, %12, %13)));
        if (OP == 8) MIXBODY(REP4(IBS_SEQ(%0, %1, %2, %3, %4) IBS_SEQ(%5, %6, %7, %8, %9) IBS_SEQ(%10, %11, %12, %13, %14)));
    }
    uint32_t r = 0;
    for (int i = 0; i < 16; i++) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP>
void run(const char *name, int instr_per_iter = 64, double ideal_cycles_per_iter = 0)
{
    uint32_t *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    const int blocks = 256 * 8, iters = 4000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 123u, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 123u, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * 256 * iters * instr_per_iter;
    printf("%-18s %8.3f ms  %7.2f Tlane-op/s  (%.2f cycles per wave64 instruction at 2.4 GHz)", name, ms,
           ops / ms / 1e9, 256.0 * 4 * 64 * 2.4e9 / (ops / (ms * 1e-3)));
    if (ideal_cycles_per_iter > 0) {
        const double cyc_per_iter = (ms * 1e-3) * 2.4e9 * 1024 / ((double)blocks * 4 * iters);
        printf("  [%.1f cycles per iteration vs %.1f from the per-instruction rates]", cyc_per_iter, ideal_cycles_per_iter);
    }
    printf("\n");
    hipFree(out);
}
int main()
{
    run<0>("v_and_b32"); run<1>("v_xor_b32"); run<2>("v_bcnt_u32_b32"); run<3>("v_and_or_b32");
    run<4>("v_bitop3_b32"); run<5>("v_fma_f32"); run<6>("v_add_u32");
    run<9>("and/bcnt alternating", 64, 32 * 2.4 + 32 * 4.2);
    run<10>("and x8 then bcnt x8", 64, 32 * 2.4 + 32 * 4.2);
    run<11>("and x2, bcnt x2", 64, 32 * 2.4 + 32 * 4.2);
    run<7>("KING mix x8", 88, 8 * (6 * 2.4 + 5 * 4.2));
    run<8>("IBS mix x12", 84, 12 * (4 * 2.4 + 3 * 4.2));
    return 0;
}
