// Micro-benchmark: does a chain of DEPENDENT vector instructions issue as fast as independent ones
// (gfx950)?  One instruction kind per run, 64 per loop trip, either all on one register (every
// instruction needs the previous result) or round-robin over eight; W waves per SIMD on every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(x) x x x x x x x x
#define IND(fmt) R8(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7))
#define DEP(fmt) R8(fmt(0) fmt(0) fmt(0) fmt(0) fmt(0) fmt(0) fmt(0) fmt(0))
#define REGS "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf)
#define I_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define I_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, |%8|\n"
#define I_MINABS(i) "v_min_f32 %" #i ", %" #i ", |%8|\n"
#define I_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define I_CMPSEL(i) "v_cmp_eq_f32_e64 vcc, |%" #i "|, %8\n\tv_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
template <int OP> __global__ void k(float *out, int iters, float seed) {
    float f0 = seed + threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    float cf = seed * 0.5f;
    for (int i = 0; i < iters; ++i) {
#define CASE(N, BODY) if (OP == N) { asm volatile(BODY : REGS : "vcc"); }
        CASE(0, IND(I_ADD)) CASE(1, DEP(I_ADD)) CASE(2, IND(I_MED3)) CASE(3, DEP(I_MED3)) CASE(4, IND(I_MINABS)) CASE(5, DEP(I_MINABS))
        CASE(6, IND(I_ALIGN)) CASE(7, DEP(I_ALIGN)) CASE(8, IND(I_CMPSEL)) CASE(9, DEP(I_CMPSEL))
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
template <int OP> void run(const char *name, float *out, int threads, int per_trip) {
    const int iters = 2000, blocks = 256;  // one workgroup per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(out, 10, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(out, iters, 1.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (threads / 64) / 4.0 * iters * per_trip;
    printf("%-28s %d waves/SIMD  %7.3f ms  %5.2f cycles per wave-instruction per SIMD, %6.2f per instruction of one wave\n", name,
           threads / 256, ms, ms * 1e6 / inst_per_simd * 2.4, ms * 1e6 / (iters * (double)per_trip) * 2.4);
}
int main() {
    float *out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    for (int threads : {256, 512, 768, 1024}) {
        run<0>("v_add_f32 independent", out, threads, 64); run<1>("v_add_f32 dependent", out, threads, 64);
        run<2>("v_med3_f32 independent", out, threads, 64); run<3>("v_med3_f32 dependent", out, threads, 64);
        run<4>("v_min_f32 independent", out, threads, 64); run<5>("v_min_f32 dependent", out, threads, 64);
        run<6>("v_alignbit independent", out, threads, 64); run<7>("v_alignbit dependent", out, threads, 64);
        run<8>("cmp+cndmask independent", out, threads, 128); run<9>("cmp+cndmask dependent", out, threads, 128);
    }
    return 0;
}
