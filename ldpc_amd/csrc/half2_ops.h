// Packed half precision: two frames' numbers side by side in one 32-bit word, and the packed
// instructions that work on both (reduced-precision fast modes, compute_dtype LDPC_F16).
#pragma once
#include "spec_common.h"

namespace ldpc {

typedef unsigned h2;  // two half-precision numbers: frame 2p in the low half, frame 2p + 1 in the high half

constexpr unsigned kAbsMask = 0x7fff7fffu, kSignMask = 0x80008000u;
constexpr float kHalfClamp = 2000.0f;
constexpr h2 kHalfClamp2 = 0x67d067d0u;   // 2000.0 twice
constexpr h2 kHalfMax2 = 0x7bff7bffu;     // 65504 twice: "no input yet"

__device__ __forceinline__ h2 pk_min(h2 a, h2 b) { h2 d; asm("v_pk_min_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ h2 pk_max(h2 a, h2 b) { h2 d; asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ h2 pk_add(h2 a, h2 b) { h2 d; asm("v_pk_add_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ h2 pk_sub(h2 a, h2 b)
{
    h2 d;
    asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ h2 pk_mul(h2 a, h2 b) { h2 d; asm("v_pk_mul_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
// c - a * b
__device__ __forceinline__ h2 pk_fnma(h2 a, h2 b, h2 c)
{
    h2 d;
    asm("v_pk_fma_f16 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ h2 pack_half2(float lo, float hi)
{
    lo = __builtin_amdgcn_fmed3f(lo, -kHalfClamp, kHalfClamp);
    hi = __builtin_amdgcn_fmed3f(hi, -kHalfClamp, kHalfClamp);
    typedef __fp16 v2h __attribute__((ext_vector_type(2)));
    const v2h p = __builtin_amdgcn_cvt_pkrtz(lo, hi);
    return *reinterpret_cast<const h2 *>(&p);
}
// the same with the clamp after the conversion (a value beyond the half-precision range converts to
// an infinity, which the clamp brings back): the bound is a packed register the caller holds anyway
__device__ __forceinline__ h2 pack_half2_clamped(float lo, float hi, h2 bound2)
{
    h2 d;
    asm("v_cvt_pkrtz_f16_f32 %0, %1, %2\n\tv_pk_max_f16 %0, %0, %3 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_min_f16 %0, %0, %3"
        : "=&v"(d) : "v"(lo), "v"(hi), "v"(bound2));
    return d;
}
// "value > 0" of one half (hard decision `0 if b > 0 else 1`, bpsk.py:240-243)
__device__ __forceinline__ bool half_positive(h2 v, int h)
{
    return h ? (int)v > 0xffff : ((int)(v << 16) > 0);
}

// Packed 16-bit integer helpers (both halves at once).  Inline constants of packed instructions are
// read from the low half by both lanes (op_sel_hi 0 for that operand).
// 1 - x, saturating at 0: 1 where x == 0, else 0
__device__ __forceinline__ unsigned pk_is_zero(unsigned x)
{
    unsigned d; asm("v_pk_sub_u16 %0, 1, %1 op_sel_hi:[0,1] clamp" : "=v"(d) : "v"(x)); return d;
}
__device__ __forceinline__ unsigned pk_mad_u16(unsigned a, unsigned b, unsigned c)
{
    unsigned d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d;
}
// 2 w + s: a funnel shift by one bit per half when s is 0 or 1
__device__ __forceinline__ unsigned pk_funnel(unsigned w, unsigned s)
{
    unsigned d; asm("v_pk_mad_u16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(w), "v"(s)); return d;
}
__device__ __forceinline__ unsigned pk_sub_u16(unsigned a, unsigned b)
{
    unsigned d; asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d;
}
template <int N>
__device__ __forceinline__ unsigned pk_shl(unsigned x)
{
    static_assert(N >= 0 && N < 16, "shift of a 16-bit half");
    if constexpr (N == 0) return x;
    else { unsigned d; asm("v_pk_lshlrev_b16 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "n"(N), "v"(x)); return d; }
}
template <int N>
__device__ __forceinline__ unsigned pk_shr(unsigned x)
{
    static_assert(N >= 0 && N < 16, "shift of a 16-bit half");
    if constexpr (N == 0) return x;
    else { unsigned d; asm("v_pk_lshrrev_b16 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "n"(N), "v"(x)); return d; }
}
// the top bit of each half spread over the half
__device__ __forceinline__ unsigned pk_sign_fill(unsigned x)
{
    unsigned d; asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(x)); return d;
}
// (a & mask) | c and (a & mask) ^ b as one instruction each
__device__ __forceinline__ unsigned and_or(unsigned a, unsigned mask, unsigned c)
{
    unsigned d; asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(mask), "v"(c)); return d;
}
__device__ __forceinline__ unsigned and_xor(unsigned a, unsigned b, unsigned mask)
{
    unsigned d; asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x6c" : "=v"(d) : "v"(a), "v"(b), "s"(mask)); return d;
}

// One input edge of a packed layered check as ONE block (the compiler puts a wait state between
// separate inline blocks that depend on each other).  Take the edge's previous message off the pair
// of posteriors q -- magnitude B where the edge held the minimum last time, else A, as A + e (B - A)
// on the bit patterns, e = 1 where the running counter c (position of that minimum minus the edge's
// own position, per half) is zero; sign in the top bit of each half of ws -- then book |q| in the
// running smallest / second smallest (m1 <= m2), funnel sign(q) into ws (the old signs leave at the
// top, the new ones enter at the bottom) and fold q into the parity word sx.  FIRST: the check's
// first edge (m1, m2 and sx start here; big = the largest half-precision number twice).
#define LDPC_HEDGE_SEL_CNT "v_pk_sub_u16 %[t], 1, %[c] op_sel_hi:[0,1] clamp\n\tv_pk_sub_u16 %[c], %[c], 1 op_sel_hi:[1,0]\n\t"
#define LDPC_HEDGE_SEL_HOT "v_pk_lshrrev_b16 %[t], 15, %[c] op_sel_hi:[0,1]\n\tv_pk_add_u16 %[c], %[c], %[c]\n\t"
#define LDPC_HEDGE_OLD "v_pk_mad_u16 %[t], %[t], %[dba], %[a]\n\tv_and_or_b32 %[t], %[ws], %[S], %[t]\n\t" \
                       "v_pk_add_f16 %[q], %[q], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define LDPC_HEDGE_BOOK "v_and_b32 %[aq], 0x7fff7fff, %[q]\n\tv_pk_max_f16 %[u], %[m1], %[aq]\n\tv_pk_min_f16 %[m2], %[m2], %[u]\n\t" \
                        "v_pk_min_f16 %[m1], %[m1], %[aq]\n\tv_pk_lshrrev_b16 %[t], 15, %[q] op_sel_hi:[0,1]\n\t"                  \
                        "v_pk_mad_u16 %[ws], %[ws], 2, %[t] op_sel_hi:[1,0,1]\n\tv_xor_b32 %[sx], %[sx], %[q]"
#define LDPC_HEDGE_BOOK1 "v_and_b32 %[aq], 0x7fff7fff, %[q]\n\tv_mov_b32 %[m1], %[aq]\n\tv_mov_b32 %[m2], %[big]\n\tv_pk_lshrrev_b16 %[t], 15, %[q] op_sel_hi:[0,1]\n\t" \
                         "v_pk_mad_u16 %[ws], %[ws], 2, %[t] op_sel_hi:[1,0,1]\n\tv_mov_b32 %[sx], %[q]"
// HOT: c is not a counter but one bit per edge ("this edge held the minimum"), the current edge's in
// the top bit of each half, shifted up edge by edge.
template <bool FIRST, bool HOT>
__device__ __forceinline__ void hedge_in(h2 &q, h2 &aq, unsigned &c, h2 a, unsigned dba, unsigned &ws, h2 &m1, h2 &m2, unsigned &sx, h2 big)
{
    unsigned t, u;
#define LDPC_HEDGE_FIRST_OPS : [q] "+v"(q), [aq] "=&v"(aq), [c] "+v"(c), [t] "=&v"(t), [ws] "+v"(ws), [m1] "=&v"(m1), [m2] "=&v"(m2), [sx] "=&v"(sx) \
                             : [a] "v"(a), [dba] "v"(dba), [S] "s"(kSignMask), [big] "v"(big)
#define LDPC_HEDGE_NEXT_OPS : [q] "+v"(q), [aq] "=&v"(aq), [c] "+v"(c), [t] "=&v"(t), [u] "=&v"(u), [ws] "+v"(ws), [m1] "+v"(m1), [m2] "+v"(m2), [sx] "+v"(sx) \
                            : [a] "v"(a), [dba] "v"(dba), [S] "s"(kSignMask)
    if constexpr (FIRST && HOT) asm(LDPC_HEDGE_SEL_HOT LDPC_HEDGE_OLD LDPC_HEDGE_BOOK1 LDPC_HEDGE_FIRST_OPS);
    else if constexpr (FIRST) asm(LDPC_HEDGE_SEL_CNT LDPC_HEDGE_OLD LDPC_HEDGE_BOOK1 LDPC_HEDGE_FIRST_OPS);
    else if constexpr (HOT) asm(LDPC_HEDGE_SEL_HOT LDPC_HEDGE_OLD LDPC_HEDGE_BOOK LDPC_HEDGE_NEXT_OPS);
    else asm(LDPC_HEDGE_SEL_CNT LDPC_HEDGE_OLD LDPC_HEDGE_BOOK LDPC_HEDGE_NEXT_OPS);
#undef LDPC_HEDGE_FIRST_OPS
#undef LDPC_HEDGE_NEXT_OPS
}
// the same for an edge whose input is given (no previous message to take off); the counter still moves on
template <bool FIRST, bool HOT>
__device__ __forceinline__ void hedge_in_given(h2 q, h2 &aq, unsigned &c, unsigned &ws, h2 &m1, h2 &m2, unsigned &sx, h2 big)
{
    unsigned t, u;
#define LDPC_HEDGE_STEP_CNT "v_pk_sub_u16 %[c], %[c], 1 op_sel_hi:[1,0]\n\t"
#define LDPC_HEDGE_STEP_HOT "v_pk_add_u16 %[c], %[c], %[c]\n\t"
#define LDPC_HEDGE_FIRST_OPS : [aq] "=&v"(aq), [c] "+v"(c), [t] "=&v"(t), [ws] "+v"(ws), [m1] "=&v"(m1), [m2] "=&v"(m2), [sx] "=&v"(sx) : [q] "v"(q), [big] "v"(big)
#define LDPC_HEDGE_NEXT_OPS : [aq] "=&v"(aq), [c] "+v"(c), [t] "=&v"(t), [u] "=&v"(u), [ws] "+v"(ws), [m1] "+v"(m1), [m2] "+v"(m2), [sx] "+v"(sx) : [q] "v"(q)
    if constexpr (FIRST && HOT) asm(LDPC_HEDGE_STEP_HOT LDPC_HEDGE_BOOK1 LDPC_HEDGE_FIRST_OPS);
    else if constexpr (FIRST) asm(LDPC_HEDGE_STEP_CNT LDPC_HEDGE_BOOK1 LDPC_HEDGE_FIRST_OPS);
    else if constexpr (HOT) asm(LDPC_HEDGE_STEP_HOT LDPC_HEDGE_BOOK LDPC_HEDGE_NEXT_OPS);
    else asm(LDPC_HEDGE_STEP_CNT LDPC_HEDGE_BOOK LDPC_HEDGE_NEXT_OPS);
#undef LDPC_HEDGE_STEP_CNT
#undef LDPC_HEDGE_STEP_HOT
#undef LDPC_HEDGE_FIRST_OPS
#undef LDPC_HEDGE_NEXT_OPS
}
#undef LDPC_HEDGE_SEL_CNT
#undef LDPC_HEDGE_SEL_HOT
#undef LDPC_HEDGE_OLD
#undef LDPC_HEDGE_BOOK
#undef LDPC_HEDGE_BOOK1
// Second pass of an edge: e = 1 where |q| == m1, as the saturating difference (m1 + 1) - |q| of the
// bit patterns (|q| >= m1); new message magnitude An + e (Bn - An) with the parity of the inputs in
// its sign bits (anp), times sign(q); e funnelled into mn; returns q + message.
__device__ __forceinline__ h2 hedge_out(h2 q, h2 aq, unsigned m1p1, unsigned anp, unsigned dba, unsigned &mn)
{
    unsigned t;
    h2 out;
    asm("v_pk_sub_u16 %[t], %[m1p1], %[aq] clamp\n\t"
        "v_pk_mad_u16 %[mn], %[mn], 2, %[t] op_sel_hi:[1,0,1]\n\tv_pk_mad_u16 %[t], %[t], %[dba], %[anp]\n\t"
        "v_bitop3_b32 %[t], %[q], %[t], %[S] bitop3:0x6c\n\tv_pk_add_f16 %[out], %[q], %[t]"
        : [out] "=v"(out), [t] "=&v"(t), [mn] "+v"(mn)
        : [q] "v"(q), [aq] "v"(aq), [m1p1] "v"(m1p1), [anp] "v"(anp), [dba] "v"(dba), [S] "s"(kSignMask));
    return out;
}

// the same without the magnitude at hand: |q| xor m1 is formed first (one instruction more, one register less)
__device__ __forceinline__ h2 hedge_out_nomag(h2 q, h2 m1, unsigned anp, unsigned dba, unsigned &mn)
{
    unsigned t;
    h2 out;
    asm("v_bitop3_b32 %[t], %[m1], %[q], %[A] bitop3:0x78\n\tv_pk_sub_u16 %[t], 1, %[t] op_sel_hi:[0,1] clamp\n\t"
        "v_pk_mad_u16 %[mn], %[mn], 2, %[t] op_sel_hi:[1,0,1]\n\tv_pk_mad_u16 %[t], %[t], %[dba], %[anp]\n\t"
        "v_bitop3_b32 %[t], %[q], %[t], %[S] bitop3:0x6c\n\tv_pk_add_f16 %[out], %[q], %[t]"
        : [out] "=v"(out), [t] "=&v"(t), [mn] "+v"(mn)
        : [q] "v"(q), [m1] "v"(m1), [anp] "v"(anp), [dba] "v"(dba), [S] "s"(kSignMask), [A] "s"(kAbsMask));
    return out;
}

}  // namespace ldpc
