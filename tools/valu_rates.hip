// Micro-benchmark: issue cost of the vector instructions the decoders lean on (gfx950).
// Each kernel runs N dependent-free copies of one instruction per loop trip on every SIMD with
// 4 waves per SIMD; prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int OP> __global__ void k(double *out, int iters, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, f4 = (float)a4, f5 = (float)a5, f6 = (float)a6, f7 = (float)a7;
    double c = seed * 0.5;
    float cf = (float)c;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { asm volatile(REP8("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
        if (OP == 1) { asm volatile(REP8("v_min_f64 %0, %0, %8\n v_min_f64 %1, %1, %8\n v_min_f64 %2, %2, %8\n v_min_f64 %3, %3, %8\n v_min_f64 %4, %4, %8\n v_min_f64 %5, %5, %8\n v_min_f64 %6, %6, %8\n v_min_f64 %7, %7, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
        if (OP == 2) { asm volatile(REP8("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
        if (OP == 3) { asm volatile(REP8("v_cmp_eq_f64 vcc, %0, %8\n v_cmp_eq_f64 vcc, %1, %8\n v_cmp_eq_f64 vcc, %2, %8\n v_cmp_eq_f64 vcc, %3, %8\n v_cmp_eq_f64 vcc, %4, %8\n v_cmp_eq_f64 vcc, %5, %8\n v_cmp_eq_f64 vcc, %6, %8\n v_cmp_eq_f64 vcc, %7, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc"); }
        if (OP == 4) { asm volatile(REP8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf)); }
        if (OP == 5) { asm volatile(REP8("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf)); }
        if (OP == 6) { asm volatile(REP8("v_med3_f32 %0, %0, %8, %8\n v_med3_f32 %1, %1, %8, %8\n v_med3_f32 %2, %2, %8, %8\n v_med3_f32 %3, %3, %8, %8\n v_med3_f32 %4, %4, %8, %8\n v_med3_f32 %5, %5, %8, %8\n v_med3_f32 %6, %6, %8, %8\n v_med3_f32 %7, %7, %8, %8\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf)); }
        if (OP == 7) { asm volatile(REP8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf) : "vcc"); }
        if (OP == 8) { asm volatile(REP8("v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf) : "vcc"); }
        if (OP == 9) { asm volatile(REP8("v_cmp_eq_u64 vcc, %0, %8\n v_cmp_eq_u64 vcc, %1, %8\n v_cmp_eq_u64 vcc, %2, %8\n v_cmp_eq_u64 vcc, %3, %8\n v_cmp_eq_u64 vcc, %4, %8\n v_cmp_eq_u64 vcc, %5, %8\n v_cmp_eq_u64 vcc, %6, %8\n v_cmp_eq_u64 vcc, %7, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc"); }
        if (OP == 10) { asm volatile(REP8("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
        if (OP == 11) { asm volatile(REP8("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
        if (OP == 12) { asm volatile(REP8("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_lt_u32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_u32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf) : "vcc"); }
        if (OP == 13) { asm volatile(REP8("v_add_u32 %0, %0, %8\n v_min_u32_sdwa %1, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n v_add_u32 %2, %2, %8\n v_min_u32_sdwa %3, %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n v_add_u32 %4, %4, %8\n v_min_u32_sdwa %5, %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n v_add_u32 %6, %6, %8\n v_min_u32_sdwa %7, %6, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf)); }
        if (OP == 14) { asm volatile(REP8("v_and_or_b32 %0, %0, %8, %8\n v_and_or_b32 %1, %1, %8, %8\n v_and_or_b32 %2, %2, %8, %8\n v_and_or_b32 %3, %3, %8, %8\n v_and_or_b32 %4, %4, %8, %8\n v_and_or_b32 %5, %5, %8, %8\n v_and_or_b32 %6, %6, %8, %8\n v_and_or_b32 %7, %7, %8, %8\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf)); }
        if (OP == 15) { asm volatile(REP8("v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %8, 31\n v_alignbit_b32 %2, %2, %8, 31\n v_alignbit_b32 %3, %3, %8, 31\n v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %8, 31\n v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %8, 31\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf)); }
        if (OP == 16) { asm volatile(REP8("v_cmp_eq_f32_e64 vcc, |%0|, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_eq_f32_e64 vcc, |%2|, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_eq_f32_e64 vcc, |%4|, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_eq_f32_e64 vcc, |%6|, %8\n v_cndmask_b32 %7, %7, %8, vcc\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf) : "vcc"); }
        if (OP == 17) { asm volatile(REP8("v_min_f32 %0, %0, |%8|\n v_min_f32 %1, %1, |%8|\n v_min_f32 %2, %2, |%8|\n v_min_f32 %3, %3, |%8|\n v_min_f32 %4, %4, |%8|\n v_min_f32 %5, %5, |%8|\n v_min_f32 %6, %6, |%8|\n v_min_f32 %7, %7, |%8|\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf)); }
        if (OP == 18) { asm volatile(REP8("v_cmp_lt_u32 vcc, %0, %8\n v_add_u32 %2, %2, %8\n v_xor_b32 %4, %4, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_u32 vcc, %6, %8\n v_add_u32 %3, %3, %8\n v_xor_b32 %5, %5, %8\n v_cndmask_b32 %7, %7, %8, vcc\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf) : "vcc"); }
        if (OP == 19) { asm volatile(REP8("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf)); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
template <int OP> void run(const char *name, double *out, int threads = 256) {
    const int iters = 2000, blocks = 256 * 4;  // threads = 256: 4 waves per SIMD on every CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(out, 10, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(out, iters, 1.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves * iters * 64 instructions
    double inst_per_simd = (threads / 64) * 1.0 * iters * 64;
    printf("%-16s %8.3f ms  %6.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
}
int main() {
    double *out; hipMalloc(&out, 256 * 4 * 256 * sizeof(double));
    run<4>("v_add_f32", out); run<5>("v_xor_b32", out); run<6>("v_med3_f32", out); run<7>("v_cndmask_b32", out);
    run<8>("v_cmp_lt_u32", out); run<11>("v_pk_add_f32", out);
    run<0>("v_add_f64", out); run<1>("v_min_f64", out); run<2>("v_fma_f64", out); run<10>("v_mul_f64", out);
    run<3>("v_cmp_eq_f64", out); run<9>("v_cmp_eq_u64", out);
    run<12>("cmp_u32+cndmask", out); run<13>("add+min_sdwa", out); run<14>("v_and_or_b32", out);
    run<15>("v_alignbit_b32", out); run<16>("cmp_f32|x|+cnd", out); run<17>("v_min_f32 |x|", out);
    run<18>("cmp,add,xor,cnd", out); run<19>("v_lshlrev_b32", out);
    printf("-- 3 waves per SIMD (192 threads per workgroup, one workgroup per CU slot)\n");
    run<4>("v_add_f32", out, 192); run<12>("cmp_u32+cndmask", out, 192); run<13>("add+min_sdwa", out, 192);
    run<6>("v_med3_f32", out, 192);
    return 0;
}
