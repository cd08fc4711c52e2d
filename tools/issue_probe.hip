// Developer probe: what a single resident wave per SIMD loses to (a) taken branches (loop back-edges), (b) dependent
// FP64 instructions close together, (c) 8-byte encodings that straddle an 8-byte boundary.  Every variant is a loop of
// hand-placed v_fma_f64 / v_fmac_f64_e32 in inline assembly, 1 024 waves = one per SIMD, timed with HIP events and with
// s_memtime inside the wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o tools/issue_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

// 16 accumulators a0..a15 (operands %0..%15), multiplier %16, addend %17
#define ACC_OPS "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), \
                "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
#define F64(i) "v_fma_f64 %" #i ", %" #i ", %16, %17\n\t"        /* 8-byte encoding */
#define F32E(i) "v_fmac_f64_e32 %" #i ", %16, %17\n\t"            /* 4-byte encoding: acc += m * c */
#define ALL16 F64(0) F64(1) F64(2) F64(3) F64(4) F64(5) F64(6) F64(7) F64(8) F64(9) F64(10) F64(11) F64(12) F64(13) F64(14) F64(15)
#define DEP1_16 F64(0) F64(0) F64(0) F64(0) F64(0) F64(0) F64(0) F64(0) F64(0) F64(0) F64(0) F64(0) F64(0) F64(0) F64(0) F64(0)
#define DEP2_16 F64(0) F64(1) F64(0) F64(1) F64(0) F64(1) F64(0) F64(1) F64(0) F64(1) F64(0) F64(1) F64(0) F64(1) F64(0) F64(1)
#define DEP3_15 F64(0) F64(1) F64(2) F64(0) F64(1) F64(2) F64(0) F64(1) F64(2) F64(0) F64(1) F64(2) F64(0) F64(1) F64(2)
#define DEP4_16 F64(0) F64(1) F64(2) F64(3) F64(0) F64(1) F64(2) F64(3) F64(0) F64(1) F64(2) F64(3) F64(0) F64(1) F64(2) F64(3)
// one 4-byte instruction, then 15 eight-byte ones that all start 4 bytes off an 8-byte boundary; next group re-aligns
#define ODD16 F32E(0) F64(1) F64(2) F64(3) F64(4) F64(5) F64(6) F64(7) F64(8) F64(9) F64(10) F64(11) F64(12) F64(13) F64(14) F64(15)
// two 4-byte instructions first: the 14 eight-byte ones stay aligned
#define EVEN16 F32E(0) F32E(1) F64(2) F64(3) F64(4) F64(5) F64(6) F64(7) F64(8) F64(9) F64(10) F64(11) F64(12) F64(13) F64(14) F64(15)
// all 4-byte
#define E32_16 F32E(0) F32E(1) F32E(2) F32E(3) F32E(4) F32E(5) F32E(6) F32E(7) F32E(8) F32E(9) F32E(10) F32E(11) F32E(12) F32E(13) F32E(14) F32E(15)

// what makes the 8-byte form slower: the same accumulate with the long encoding (size only), a two-source multiply,
// a scalar multiplier (fewer VGPR reads), packed float32 (VOP3P)
#define G(i) "v_fmac_f64_e64 %" #i ", %16, %17\n\t"
#define M(i) "v_mul_f64 %" #i ", %" #i ", %16\n\t"
#define SC(i) "v_fma_f64 %" #i ", %" #i ", 0.5, %17\n\t"
#define PK(i) "v_pk_fma_f32 %" #i ", %" #i ", %16, %17\n\t"
#define X16(f) f(0) f(1) f(2) f(3) f(4) f(5) f(6) f(7) f(8) f(9) f(10) f(11) f(12) f(13) f(14) f(15)

// interleavings of 4-byte (A) and 8-byte (B) encodings at a fixed 1 : 1 mix, and two other mixes
#define AB_16 F32E(0) F64(1) F32E(2) F64(3) F32E(4) F64(5) F32E(6) F64(7) F32E(8) F64(9) F32E(10) F64(11) F32E(12) F64(13) F32E(14) F64(15)
#define A2B2_16 F32E(0) F32E(1) F64(2) F64(3) F32E(4) F32E(5) F64(6) F64(7) F32E(8) F32E(9) F64(10) F64(11) F32E(12) F32E(13) F64(14) F64(15)
#define A4B4_16 F32E(0) F32E(1) F32E(2) F32E(3) F64(4) F64(5) F64(6) F64(7) F32E(8) F32E(9) F32E(10) F32E(11) F64(12) F64(13) F64(14) F64(15)
#define A8B8_16 F32E(0) F32E(1) F32E(2) F32E(3) F32E(4) F32E(5) F32E(6) F32E(7) F64(8) F64(9) F64(10) F64(11) F64(12) F64(13) F64(14) F64(15)
#define A16 E32_16
#define B16 ALL16
#define ABB_15 F32E(0) F64(1) F64(2) F32E(3) F64(4) F64(5) F32E(6) F64(7) F64(8) F32E(9) F64(10) F64(11) F32E(12) F64(13) F64(14)
#define ABBB_16 F32E(0) F64(1) F64(2) F64(3) F32E(4) F64(5) F64(6) F64(7) F32E(8) F64(9) F64(10) F64(11) F32E(12) F64(13) F64(14) F64(15)
#define AAB_15 F32E(0) F32E(1) F64(2) F32E(3) F32E(4) F64(5) F32E(6) F32E(7) F64(8) F32E(9) F32E(10) F64(11) F32E(12) F32E(13) F64(14)

#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R32(x) R16(x) R16(x)

#define KERNEL(NAME, BODY, PER_ITER)                                                                              \
    __global__ void __launch_bounds__(64) NAME(double *out, long long *cyc, int iters, double m, double c) {      \
        double a[16];                                                                                              \
        for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3 + i;                                                \
        const long long t0 = __builtin_readcyclecounter();                                                         \
        for (int it = 0; it < iters; ++it) asm volatile(".p2align 3\n\t" BODY : ACC_OPS : "v"(m), "v"(c));                         \
        const long long t1 = __builtin_readcyclecounter();                                                         \
        double s = 0;                                                                                              \
        for (int i = 0; i < 16; ++i) s += a[i];                                                                    \
        out[blockIdx.x * 64 + threadIdx.x] = s;                                                                    \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                           \
    }                                                                                                              \
    static const int NAME##_per_iter = PER_ITER;

KERNEL(ind16, ALL16, 16)
KERNEL(ind32, R2(ALL16), 32)
KERNEL(ind64, R4(ALL16), 64)
KERNEL(ind128, R8(ALL16), 128)
KERNEL(ind256, R16(ALL16), 256)
KERNEL(ind512, R32(ALL16), 512)
KERNEL(dep1, R16(DEP1_16), 256)
KERNEL(dep2, R16(DEP2_16), 256)
KERNEL(dep3, R16(DEP3_15), 240)
KERNEL(dep4, R16(DEP4_16), 256)
KERNEL(odd256, R16(ODD16), 256)
KERNEL(even256, R16(EVEN16), 256)
KERNEL(e32_256, R16(E32_16), 256)

#define OFF4 "s_nop 0\n\t"      /* after the .p2align 3 of the loop body: everything that follows starts 4 bytes off */
KERNEL(ind256_off4, OFF4 R16(ALL16), 256)
KERNEL(pk_off4, OFF4 R16(X16(PK)), 256)
KERNEL(ab_off4, OFF4 R16(AB_16), 256)
KERNEL(a4b4_off4, OFF4 R16(A4B4_16), 256)
KERNEL(fmac_e64, R16(X16(G)), 256)
KERNEL(mul_e64, R16(X16(M)), 256)
KERNEL(fma_const, R16(X16(SC)), 256)
KERNEL(pk_fma, R16(X16(PK)), 256)
KERNEL(ab, R16(AB_16), 256)
KERNEL(a2b2, R16(A2B2_16), 256)
KERNEL(a4b4, R16(A4B4_16), 256)
KERNEL(a8b8, R16(A8B8_16), 256)
KERNEL(a16b16, R8(A16 B16), 256)
KERNEL(a32b32, R4(A16 A16 B16 B16), 256)
KERNEL(abb, R16(ABB_15), 240)
KERNEL(abbb, R16(ABBB_16), 256)
KERNEL(aab, R16(AAB_15), 240)

// the whole loop in assembly, so that the position of its first instruction inside a 64-byte line is chosen here:
// PAD = number of 4-byte s_nop between a 64-byte boundary and the loop label (executed once, before the loop)
#define ASM_LOOP_KERNEL(NAME, PAD, BODY, PER_ITER)                                                                \
    __global__ void __launch_bounds__(64) NAME(double *out, long long *cyc, int iters, double m, double c) {      \
        double a[16];                                                                                              \
        for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3 + i;                                                \
        const long long t0 = __builtin_readcyclecounter();                                                         \
        asm volatile("s_mov_b32 s20, %18\n\t.p2align 6\n\t" PAD "1:\n\t" BODY                                     \
                     "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\t"                       \
                     : ACC_OPS : "v"(m), "v"(c), "s"(iters) : "s20", "scc");                                       \
        const long long t1 = __builtin_readcyclecounter();                                                         \
        double s = 0;                                                                                              \
        for (int i = 0; i < 16; ++i) s += a[i];                                                                    \
        out[blockIdx.x * 64 + threadIdx.x] = s;                                                                    \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                           \
    }                                                                                                              \
    static const int NAME##_per_iter = PER_ITER;
#define NOP2 "s_nop 0\n\ts_nop 0\n\t"
#define NOP4 NOP2 NOP2
#define NOP8 NOP4 NOP4
ASM_LOOP_KERNEL(loop16_at0, "", ALL16, 16)
ASM_LOOP_KERNEL(loop16_at8, NOP2, ALL16, 16)
ASM_LOOP_KERNEL(loop16_at16, NOP4, ALL16, 16)
ASM_LOOP_KERNEL(loop16_at32, NOP8, ALL16, 16)
ASM_LOOP_KERNEL(loop16_at48, NOP8 NOP4, ALL16, 16)
ASM_LOOP_KERNEL(loop16_at56, NOP8 NOP4 NOP2, ALL16, 16)
ASM_LOOP_KERNEL(loop64_at0, "", R4(ALL16), 64)
ASM_LOOP_KERNEL(loop64_at32, NOP8, R4(ALL16), 64)
ASM_LOOP_KERNEL(loop64_at56, NOP8 NOP4 NOP2, R4(ALL16), 64)

typedef void (*kern_t)(double *, long long *, int, double, double);

static void run(const char *name, kern_t k, int per_iter, int waves_per_simd, long long total_instr) {
    const int blocks = 1024 * waves_per_simd;
    const int iters = (int)(total_instr / per_iter);
    double *out;
    long long *cyc;
    hipMalloc(&out, (size_t)blocks * 64 * sizeof(double));
    hipMalloc(&cyc, (size_t)blocks * sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0000001, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    long long lo = h[0], hi = h[0];
    for (long long v : h) { mean += (double)v; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    mean /= blocks;
    const double n = (double)iters * per_iter;
    // s_memtime ticks per instruction of one wave, and the same from the wall clock at 2.4 GHz
    printf("%-12s waves/SIMD=%d %4d instr/iter %8.3f ms  ticks/instr %.4f (min %.4f max %.4f)  ticks/iter %.1f  wall ns/instr %.4f\n",
           name, waves_per_simd, per_iter, best, mean / n, lo / n, hi / n, mean / iters, best * 1e6 / (n * waves_per_simd));
    hipFree(out);
    hipFree(cyc);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

#define RUN(NAME, W) run(#NAME, NAME, NAME##_per_iter, W, total)

int main(int argc, char **argv) {
    const long long total = argc > 1 ? atoll(argv[1]) : 4000000LL;   // FP64 instructions per wave
    for (int w : {1}) {
        RUN(ind16, w); RUN(ind32, w); RUN(ind64, w); RUN(ind128, w); RUN(ind256, w); RUN(ind512, w);
        RUN(dep1, w); RUN(dep2, w); RUN(dep3, w); RUN(dep4, w);
        RUN(odd256, w); RUN(even256, w); RUN(e32_256, w);
        RUN(ind256_off4, w); RUN(pk_off4, w); RUN(ab_off4, w); RUN(a4b4_off4, w);
        RUN(loop16_at0, w); RUN(loop16_at8, w); RUN(loop16_at16, w); RUN(loop16_at32, w); RUN(loop16_at48, w); RUN(loop16_at56, w);
        RUN(loop64_at0, w); RUN(loop64_at32, w); RUN(loop64_at56, w);
        RUN(fmac_e64, w); RUN(mul_e64, w); RUN(fma_const, w); RUN(pk_fma, w);
        RUN(ab, w); RUN(a2b2, w); RUN(a4b4, w); RUN(a8b8, w); RUN(a16b16, w); RUN(a32b32, w); RUN(abb, w); RUN(abbb, w); RUN(aab, w);
    }
    return 0;
}
