// Micro-benchmark: issue cost of candidate vector instructions for the layered kernel (gfx950).
// N independent copies of one instruction (or a short mix) per loop trip, W waves per SIMD on
// every CU; prints cycles per wave-instruction per SIMD at an assumed 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(x) x x x x x x x x
#define OPS8(fmt) R8(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7))
#define REGS "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf), "s"(sm)

#define I_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define I_SUB(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define I_MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define I_MINABS(i) "v_min_f32 %" #i ", %" #i ", |%8|\n"
#define I_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define I_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %8\n"
#define I_MED3ABS(i) "v_med3_f32 %" #i ", %" #i ", %8, |%8|\n"
#define I_MIN3(i) "v_min3_f32 %" #i ", %" #i ", %8, %8\n"
#define I_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define I_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define I_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define I_ANDLIT(i) "v_and_b32 %" #i ", 0x7fffffff, %" #i "\n"
#define I_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define I_BFI(i) "v_bfi_b32 %" #i ", %8, %" #i ", %8\n"
#define I_BFILIT(i) "v_bfi_b32 %" #i ", s20, %" #i ", %8\n"
#define I_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %8\n"
#define I_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define I_XOR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %8\n"
#define I_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define I_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define I_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define I_ASHR(i) "v_ashrrev_i32 %" #i ", 31, %" #i "\n"
#define I_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define I_CNDVCC(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define I_CNDS(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, %9\n"
#define I_CMPU(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n"
#define I_CMPUS(i) "v_cmp_lt_u32_e64 s[20:21], %" #i ", %8\n"
#define I_CMPF(i) "v_cmp_eq_f32 vcc, %" #i ", %8\n"
#define I_CMPFABS(i) "v_cmp_eq_f32_e64 vcc, |%" #i "|, %8\n"
#define I_CMPLIT(i) "v_cmp_eq_u32 vcc, 5, %" #i "\n"
#define I_ADDC(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n"
#define I_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define I_PKADD32(i) "v_pk_add_f32 %" #i ", %" #i ", %10\n"
#define I_PKMUL32(i) "v_pk_mul_f32 %" #i ", %" #i ", %10\n"
#define I_PKFMA32(i) "v_pk_fma_f32 %" #i ", %" #i ", %10, %10\n"
#define I_PKADD16(i) "v_pk_add_f16 %" #i ", %" #i ", %8\n"
#define I_PKMIN16(i) "v_pk_min_f16 %" #i ", %" #i ", %8\n"
#define I_PKMAX16(i) "v_pk_max_f16 %" #i ", %" #i ", %8\n"
#define I_PKMUL16(i) "v_pk_mul_f16 %" #i ", %" #i ", %8\n"
#define I_PKFMA16(i) "v_pk_fma_f16 %" #i ", %" #i ", %8, %8\n"
#define I_PKMINU16(i) "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define I_PKMAXI16(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define I_PKADDU16(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define I_PKSUBI16(i) "v_pk_sub_i16 %" #i ", %" #i ", %8\n"
#define I_PKLSHL16(i) "v_pk_lshlrev_b16 %" #i ", 1, %" #i "\n"
#define I_PKASHR16(i) "v_pk_ashrrev_i16 %" #i ", 15, %" #i "\n"
#define I_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %8\n"
#define I_MINI16(i) "v_min_i16 %" #i ", %" #i ", %8\n"
#define I_MINU32(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define I_MIN3U(i) "v_min3_u32 %" #i ", %" #i ", %8, %8\n"
#define I_MED3U(i) "v_med3_u32 %" #i ", %" #i ", %8, %8\n"
#define I_SADU8(i) "v_sad_u8 %" #i ", %" #i ", %8, %8\n"
#define I_DPPMOV(i) "v_mov_b32_dpp %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_PKMADU16(i) "v_pk_mad_u16 %" #i ", %" #i ", %8, %8\n"
#define I_PKMADU16C(i) "v_pk_mad_u16 %" #i ", %" #i ", 2, %8 op_sel_hi:[1,0,1]\n"
#define I_PKSUBSAT(i) "v_pk_sub_u16 %" #i ", 1, %" #i " op_sel_hi:[0,1] clamp\n"
#define I_PKSUBU16(i) "v_pk_sub_u16 %" #i ", %" #i ", %8\n"
#define I_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %8 bitop3:0x6c\n"
#define I_PKLSHR16(i) "v_pk_lshrrev_b16 %" #i ", 15, %" #i " op_sel_hi:[0,1]\n"
#define I_PKMADI16(i) "v_pk_mad_i16 %" #i ", %" #i ", %8, %8\n"
#define I_MADU16(i) "v_mad_u16 %" #i ", %" #i ", %8, %8\n"
#define I_MADU32U24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %8\n"
#define I_PKMULLO(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n"
#define I_SDWAAND(i) "v_and_b32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n"

template <int OP> __global__ void k(float *out, int iters, float seed) {
    float f0 = seed + threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    float cf = seed * 0.5f;
    unsigned long long sm = 0x5555555555555555ull;
    double d0 = f0, d1 = f1, d2 = f2, d3 = f3, d4 = f4, d5 = f5, d6 = f6, d7 = f7, cd = cf;
    for (int i = 0; i < iters; ++i) {
#define CASE(N, MAC) if (OP == N) { asm volatile(OPS8(MAC) : REGS : "vcc", "s20", "s21"); }
#define CASED(N, MAC) if (OP == N) { asm volatile(OPS8(MAC) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(cf), "s"(sm), "v"(cd) : "vcc"); }
        CASE(0, I_ADD) CASE(1, I_SUB) CASE(2, I_MIN) CASE(3, I_MINABS) CASE(4, I_MAX) CASE(5, I_MED3) CASE(6, I_MED3ABS)
        CASE(7, I_MIN3) CASE(8, I_FMA) CASE(9, I_MUL) CASE(10, I_XOR) CASE(11, I_AND) CASE(12, I_ANDLIT) CASE(13, I_OR)
        CASE(14, I_BFI) CASE(15, I_BFILIT) CASE(16, I_ANDOR) CASE(17, I_LSHLOR) CASE(18, I_XOR3) CASE(19, I_ALIGN)
        CASE(20, I_LSHL) CASE(21, I_LSHR) CASE(22, I_ASHR) CASE(23, I_MOV) CASE(24, I_CNDVCC) CASE(25, I_CNDS)
        CASE(26, I_CMPU) CASE(27, I_CMPUS) CASE(28, I_CMPF) CASE(29, I_CMPFABS) CASE(30, I_CMPLIT) CASE(31, I_ADDC)
        CASE(32, I_ADDU) CASED(33, I_PKADD32) CASED(34, I_PKMUL32) CASED(35, I_PKFMA32) CASE(36, I_PKADD16)
        CASE(37, I_PKMIN16) CASE(38, I_PKMAX16) CASE(39, I_PKMUL16) CASE(40, I_PKFMA16) CASE(41, I_PKMINU16)
        CASE(42, I_PKMAXI16) CASE(43, I_PKADDU16) CASE(44, I_PKSUBI16) CASE(45, I_PKLSHL16) CASE(46, I_PKASHR16)
        CASE(47, I_PERM) CASE(48, I_MINI16) CASE(49, I_MINU32) CASE(50, I_MIN3U) CASE(51, I_MED3U) CASE(52, I_SADU8)
        CASE(53, I_DPPMOV) CASE(54, I_SDWAAND) CASE(55, I_PKMADU16) CASE(56, I_PKMADU16C) CASE(57, I_PKSUBSAT)
        CASE(58, I_PKSUBU16) CASE(59, I_BITOP3) CASE(60, I_PKLSHR16) CASE(61, I_PKMADI16) CASE(62, I_MADU16) CASE(63, I_MADU32U24) CASE(64, I_PKMULLO)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
template <int OP> void run(const char *name, float *out, int threads) {
    const int iters = 2000, blocks = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(out, 10, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(out, iters, 1.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // blocks/256 workgroups per CU in turn, each threads/64 waves spread over 4 SIMDs
    double inst_per_simd = (blocks / 256.0) * (threads / 64) / 4.0 * iters * 64;
    printf("%-22s w/SIMD-at-once %d  %8.3f ms  %5.2f cycles per wave-instruction per SIMD\n", name, threads / 64 / 4 * 1, ms,
           ms * 1e6 / inst_per_simd * 2.4);
}
int main() {
    float *out; hipMalloc(&out, 256 * 4 * 1024 * sizeof(float));
#define RUN(N, NAME) run<N>(NAME, out, 1024);
    // 1024 threads per workgroup = 4 waves per SIMD resident per workgroup (4 workgroups per CU in turn: 2 fit at once)
    RUN(0, "v_add_f32") RUN(1, "v_sub_f32") RUN(2, "v_min_f32") RUN(3, "v_min_f32 |x|") RUN(4, "v_max_f32") RUN(5, "v_med3_f32")
    RUN(6, "v_med3_f32 |x|") RUN(7, "v_min3_f32") RUN(8, "v_fma_f32") RUN(9, "v_mul_f32") RUN(10, "v_xor_b32") RUN(11, "v_and_b32")
    RUN(12, "v_and_b32 literal") RUN(13, "v_or_b32") RUN(14, "v_bfi_b32") RUN(15, "v_bfi_b32 literal") RUN(16, "v_and_or_b32")
    RUN(17, "v_lshl_or_b32") RUN(18, "v_xor3_b32") RUN(19, "v_alignbit_b32") RUN(20, "v_lshlrev_b32") RUN(21, "v_lshrrev_b32")
    RUN(22, "v_ashrrev_i32") RUN(23, "v_mov_b32") RUN(24, "v_cndmask vcc") RUN(25, "v_cndmask sgpr") RUN(26, "v_cmp_lt_u32 vcc")
    RUN(27, "v_cmp_lt_u32 sgpr") RUN(28, "v_cmp_eq_f32") RUN(29, "v_cmp_eq_f32 |x|") RUN(30, "v_cmp_eq_u32 inline") RUN(31, "v_addc_co_u32")
    RUN(32, "v_add_u32") RUN(33, "v_pk_add_f32") RUN(34, "v_pk_mul_f32") RUN(35, "v_pk_fma_f32") RUN(36, "v_pk_add_f16")
    RUN(37, "v_pk_min_f16") RUN(38, "v_pk_max_f16") RUN(39, "v_pk_mul_f16") RUN(40, "v_pk_fma_f16") RUN(41, "v_pk_min_u16")
    RUN(42, "v_pk_max_i16") RUN(43, "v_pk_add_u16") RUN(44, "v_pk_sub_i16") RUN(45, "v_pk_lshlrev_b16") RUN(46, "v_pk_ashrrev_i16")
    RUN(47, "v_perm_b32") RUN(48, "v_min_i16") RUN(49, "v_min_u32") RUN(50, "v_min3_u32") RUN(51, "v_med3_u32") RUN(52, "v_sad_u8")
    RUN(53, "v_mov_b32 dpp") RUN(54, "v_and_b32 sdwa") RUN(55, "v_pk_mad_u16") RUN(56, "v_pk_mad_u16 x2+") RUN(57, "v_pk_sub_u16 clamp")
    RUN(58, "v_pk_sub_u16") RUN(59, "v_bitop3_b32") RUN(60, "v_pk_lshrrev_b16") RUN(61, "v_pk_mad_i16") RUN(62, "v_mad_u16") RUN(63, "v_mad_u32_u24") RUN(64, "v_pk_mul_lo_u16")
    printf("-- 3 waves per SIMD (768 threads)\n");
    run<0>("v_add_f32", out, 768); run<3>("v_min_f32 |x|", out, 768); run<5>("v_med3_f32", out, 768); run<24>("v_cndmask vcc", out, 768);
    run<37>("v_pk_min_f16", out, 768);
    return 0;
}
