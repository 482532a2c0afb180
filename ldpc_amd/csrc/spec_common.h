// Helpers shared by the code-specialised kernels: compile-time loops over the generated circulant
// tables, sign/magnitude primitives written as single instructions, rotated LDS offsets.
#pragma once
#include "bp_common.h"
#include "bp_math.h"
#include "gen/codes_gen.h"

namespace ldpc {

template <int I> struct IC { static constexpr int value = I; constexpr operator int() const { return I; } };
template <int B, int E, class Fn>
__device__ __forceinline__ void static_for(Fn &&fn)
{
    if constexpr (B < E) { fn(IC<B>{}); static_for<B + 1, E>(fn); }
}

template <typename T> struct SNum;
template <> struct SNum<float> {
    typedef float2 Pair;
    __device__ static __forceinline__ unsigned sign_word(float x) { return __float_as_uint(x); }
    __device__ static __forceinline__ float with_sign(float mag, unsigned s) {
        return __uint_as_float(__float_as_uint(mag) | (s & 0x80000000u));
    }
    // written out as instructions: the min/max builtins put a quieting canonicalisation in front
    // of every operand that comes from memory
    __device__ static __forceinline__ float flip_sign(float x, unsigned signbit) {
        return __uint_as_float(__float_as_uint(x) ^ signbit);
    }
    __device__ static __forceinline__ float min_abs(float m, float b) {
        float d; asm("v_min_f32 %0, %1, |%2|" : "=v"(d) : "v"(m), "v"(b)); return d;
    }
    __device__ static __forceinline__ float second(float m1, float m2, float b) {  // m1 <= m2
        float d; asm("v_med3_f32 %0, %1, %2, |%3|" : "=v"(d) : "v"(m1), "v"(m2), "v"(b)); return d;
    }
    __device__ static __forceinline__ float max_abs(float m, float b) {
        float d; asm("v_max_f32 %0, %1, |%2|" : "=v"(d) : "v"(m), "v"(b)); return d;
    }
    // book |b| of a check's P-th input in the running smallest / second smallest (m1 <= m2, both +infinity
    // before the first input): the first two inputs need one and two instructions instead of three
    template <int P>
    __device__ static __forceinline__ void book(float &m1, float &m2, float b) {
        if constexpr (P == 0) { m1 = __builtin_fabsf(b); }
        else if constexpr (P == 1) { m2 = max_abs(m1, b); m1 = min_abs(m1, b); }
        else { m2 = second(m1, m2, b); m1 = min_abs(m1, b); }
    }
    __device__ static __forceinline__ Pair make(float a, float b) { return make_float2(a, b); }
    // mag = |q| == m1 ? b : a, and the outcome of the comparison is shifted into `bits` from the
    // right (carry-in of an add), so that no lane mask and no constant has to be kept in registers
    __device__ static __forceinline__ float pick_min(float q, float m1, float a, float b, unsigned &bits) {
        float mag;
        asm("v_cmp_eq_f32_e64 vcc, |%2|, %3\n\tv_cndmask_b32_e32 %0, %4, %5, vcc\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc"
            : "=&v"(mag), "+v"(bits) : "v"(q), "v"(m1), "v"(a), "v"(b) : "vcc");
        return mag;
    }
    // q + sign(q) * (|q| == m1 ? b : a), the comparison booked in `bits` as above.  One block: the
    // compiler puts a wait state between such a block and a vector instruction that reads its result
    // at once, and here the only reader left is an LDS store.
    __device__ static __forceinline__ float pick_add(float q, float m1, float a, float b, unsigned &bits) {
        float out, t;
        asm("v_cmp_eq_f32_e64 vcc, |%3|, %4\n\tv_cndmask_b32_e32 %1, %5, %6, vcc\n\tv_addc_co_u32_e32 %2, vcc, %2, %2, vcc\n\t"
            "v_bitop3_b32 %1, %3, %1, %7 bitop3:0x6c\n\tv_add_f32_e32 %0, %3, %1"
            : "=v"(out), "=&v"(t), "+v"(bits) : "v"(q), "v"(m1), "v"(a), "v"(b), "s"(0x80000000u) : "vcc");
        return out;
    }
    // One input edge of a layered check as one block: take the edge's previous message (magnitude b
    // if this edge held the minimum last time, else a; sign in bit 31 of ws) off the posterior q and
    // book |q| and sign(q) in the running smallest / second smallest (m1 <= m2) and in the sign word.
    // ws is ONE word for both sets of signs: the previous messages' signs leave it at the top, one per
    // edge, while the signs of the new inputs enter at the bottom (a funnel shift by one), so after
    // the check's last edge it holds the new signs, first edge highest.  FIRST: the check's first
    // edge (m1, m2 start here; inf = a register that holds +infinity).
#define LDPC_EDGE_OLD "v_cmp_eq_u32_e32 vcc, %[p], %[idx]\n\tv_cndmask_b32_e32 %[t], %[a], %[b], vcc\n\t" \
                      "v_and_or_b32 %[t], %[ws], %[S], %[t]\n\tv_sub_f32_e32 %[q], %[q], %[t]\n\t"
#define LDPC_EDGE_BOOK "v_med3_f32 %[m2], %[m1], %[m2], |%[q]|\n\tv_min_f32 %[m1], %[m1], |%[q]|\n\t" \
                       "v_alignbit_b32 %[ws], %[ws], %[q], 31"
#define LDPC_EDGE_BOOK1 "v_med3_f32 %[m2], %[inf], %[inf], |%[q]|\n\tv_min_f32 %[m1], %[inf], |%[q]|\n\t" \
                        "v_alignbit_b32 %[ws], %[ws], %[q], 31"
    template <int P, bool FIRST>
    __device__ static __forceinline__ void edge_in(float &q, unsigned idx_old, float a, float b, unsigned &ws, float &m1,
                                                   float &m2, float inf) {
        float t;
        if constexpr (FIRST)
            asm(LDPC_EDGE_OLD LDPC_EDGE_BOOK1
                : [q] "+v"(q), [t] "=&v"(t), [ws] "+v"(ws), [m1] "=&v"(m1), [m2] "=&v"(m2)
                : [p] "i"(P), [idx] "v"(idx_old), [a] "v"(a), [b] "v"(b), [S] "s"(0x80000000u), [inf] "v"(inf) : "vcc");
        else
            asm(LDPC_EDGE_OLD LDPC_EDGE_BOOK
                : [q] "+v"(q), [t] "=&v"(t), [ws] "+v"(ws), [m1] "+v"(m1), [m2] "+v"(m2)
                : [p] "i"(P), [idx] "v"(idx_old), [a] "v"(a), [b] "v"(b), [S] "s"(0x80000000u) : "vcc");
    }
    // the same for an edge whose input q is given (no previous message to take off)
    template <bool FIRST>
    __device__ static __forceinline__ void edge_in_given(float q, unsigned &ws, float &m1, float &m2, float inf) {
        if constexpr (FIRST)
            asm(LDPC_EDGE_BOOK1 : [m1] "=&v"(m1), [m2] "=&v"(m2), [ws] "+v"(ws) : [q] "v"(q), [inf] "v"(inf));
        else
            asm(LDPC_EDGE_BOOK : [m1] "+v"(m1), [m2] "+v"(m2), [ws] "+v"(ws) : [q] "v"(q));
    }
#undef LDPC_EDGE_OLD
#undef LDPC_EDGE_BOOK
#undef LDPC_EDGE_BOOK1
};
template <> struct SNum<double> {
    typedef double2 Pair;
    __device__ static __forceinline__ unsigned sign_word(double x) { return (unsigned)__double2hiint(x); }
    __device__ static __forceinline__ double with_sign(double mag, unsigned s) {
        return __hiloint2double(__double2hiint(mag) | (int)(s & 0x80000000u), __double2loint(mag));
    }
    __device__ static __forceinline__ double flip_sign(double x, unsigned signbit) {
        return __hiloint2double(__double2hiint(x) ^ (int)signbit, __double2loint(x));
    }
    __device__ static __forceinline__ double min_abs(double m, double b) {
        double d; asm("v_min_f64 %0, %1, |%2|" : "=v"(d) : "v"(m), "v"(b)); return d;
    }
    __device__ static __forceinline__ double second(double m1, double m2, double b) {
        double t, d;
        asm("v_max_f64 %0, %1, |%2|" : "=v"(t) : "v"(m1), "v"(b));
        asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(m2), "v"(t));
        return d;
    }
    __device__ static __forceinline__ double max_abs(double m, double b) {
        double d; asm("v_max_f64 %0, %1, |%2|" : "=v"(d) : "v"(m), "v"(b)); return d;
    }
    template <int P>
    __device__ static __forceinline__ void book(double &m1, double &m2, double b) {
        if constexpr (P == 0) { m1 = __builtin_fabs(b); }
        else if constexpr (P == 1) { m2 = max_abs(m1, b); m1 = min_abs(m1, b); }
        else { m2 = second(m1, m2, b); m1 = min_abs(m1, b); }
    }
    __device__ static __forceinline__ Pair make(double a, double b) { return make_double2(a, b); }
    __device__ static __forceinline__ double pick_min(double q, double m1, double a, double b, unsigned &bits) {
        int lo, hi;
        asm("v_cmp_eq_f64_e64 vcc, |%3|, %4\n\tv_cndmask_b32_e32 %0, %5, %7, vcc\n\tv_cndmask_b32_e32 %1, %6, %8, vcc\n\t"
            "v_addc_co_u32_e32 %2, vcc, %2, %2, vcc"
            : "=&v"(lo), "=&v"(hi), "+v"(bits)
            : "v"(q), "v"(m1), "v"(__double2loint(a)), "v"(__double2hiint(a)), "v"(__double2loint(b)), "v"(__double2hiint(b))
            : "vcc");
        return __hiloint2double(hi, lo);
    }
    template <int P, bool FIRST>
    __device__ static __forceinline__ void edge_in(double &q, unsigned idx_old, double a, double b, unsigned &ws, double &m1,
                                                   double &m2, double inf) {
        q = q - with_sign((int)idx_old == P ? b : a, ws);
        edge_in_given<FIRST>(q, ws, m1, m2, inf);
    }
    template <bool FIRST>
    __device__ static __forceinline__ void edge_in_given(double q, unsigned &ws, double &m1, double &m2, double inf) {
        if constexpr (FIRST) { m1 = inf; m2 = inf; }
        m2 = second(m1, m2, q);
        m1 = min_abs(m1, q);
        ws = __builtin_amdgcn_alignbit(ws, sign_word(q), 31);  // old sign out at the top, new sign in at the bottom
    }
    __device__ static __forceinline__ double pick_add(double q, double m1, double a, double b, unsigned &bits) {
        const double mag = pick_min(q, m1, a, b, bits);
        return q + flip_sign(mag, sign_word(q) & 0x80000000u);
    }
};

// Edges of shift 0 connect check (br, z) with variable (bc, z), which the same lane owns: where a
// lane's check states fit registers (base graphs of at most four block rows) the single-wave
// flooding kernels pass such an edge's message and check state through registers, not LDS.
template <class C>
constexpr bool spec_local_edges()
{
#ifdef LDPC_NO_LOCAL_EDGES
    return false;
#else
    return C::MB <= 4;
#endif
}

constexpr bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

// byte offset of element ((z + s) mod Z) of a Z-vector with ELEM-byte elements, given zb = z*ELEM
template <int Z, int ELEM, int S>
__device__ __forceinline__ unsigned rot_fwd(unsigned zb)
{
    if constexpr (S == 0) return zb;
    else if constexpr (is_pow2(Z)) return (zb + S * ELEM) & (unsigned)(Z * ELEM - 1);
    else { const unsigned t = zb + S * ELEM; return min(t, t - (unsigned)(Z * ELEM)); }
}
// ... of element ((z - s) mod Z)
template <int Z, int ELEM, int S>
__device__ __forceinline__ unsigned rot_bwd(unsigned zb)
{
    return rot_fwd<Z, ELEM, (Z - S) % Z>(zb);
}

// The NEXT frame's channel LLRs, fetched straight into an LDS staging area [NB][64 lanes] while the
// current frame is decoded (single-wave workgroups, float LLRs).  global_load_lds moves the data
// without holding registers while it is on its way; without it a wave idles for a memory round
// trip at the start of every frame.  Written as instructions: the compiler's own bookkeeping for
// LDS-DMA makes every later LDS store wait for the transfer, which would put the round trip back
// on the critical path.  The caller waits (s_waitcnt 0) before reading the area and again before
// refilling it.  Lanes without a frame read frame 0, which exists.
// Worth it only while LDS is not what limits residency: with the staging area at least twelve
// single-wave workgroups (three per SIMD) must still fit a compute unit.
// (a kernel whose registers allow fewer workgroups anyway passes that number as `wanted`)
template <class C, typename TL>
constexpr bool llr_prefetch_pays(int lds_bytes, int wanted = 12)
{
    return sizeof(TL) == 4 && (160 * 1024) / (lds_bytes + C::NB * 256) >= wanted;
}
template <class C, int F>
__device__ __forceinline__ void llr_prefetch(float *stage, const float *llr, long long group, long long frames,
                                             bool lane_on, int f, int z)
{
    const long long fr = group * F + f;
    const float *src = llr + (lane_on && fr < frames ? fr : 0) * (C::NB * C::Z) + z;
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float *)stage;
    static_for<0, C::NB>([&](auto bc) {
        const float *from = src + bc * C::Z;
        const unsigned to = base + bc * 256u;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"
                     :: "v"(from), "s"(to) : "memory", "m0");
    });
}
template <class C, int F>
__device__ __forceinline__ void llr_prefetch(float *, const double *, long long, long long, bool, int, int) {}
// the same for one frame given by its index (lanes without a frame pass `valid` = false and read frame 0)
template <class C>
__device__ __forceinline__ void llr_prefetch_frame(float *stage, const float *llr, long long frame, bool valid, int z)
{
    const float *src = llr + (valid ? frame : 0) * (C::NB * C::Z) + z;
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float *)stage;
    static_for<0, C::NB>([&](auto bc) {
        const float *from = src + bc * C::Z;
        const unsigned to = base + bc * 256u;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"
                     :: "v"(from), "s"(to) : "memory", "m0");
    });
}

// Hard decisions as ballots: one 64-bit lane mask per block column; the syndrome of a block row is
// the XOR of those masks rotated by the circulant shifts (scalar instructions only).
template <int Z, int F>
constexpr unsigned long long replicate_field(unsigned long long x)
{
    unsigned long long r = 0;
    for (int f = 0; f < F; ++f) r |= x << (f * Z);
    return r;
}
// bit (f*Z + z) of the result = bit (f*Z + (z + S) mod Z) of m, for every slot f
template <int Z, int F, int S>
__device__ __forceinline__ unsigned long long rot_fields(unsigned long long m)
{
    if constexpr (S == 0) return m;
    else if constexpr (Z == 64) return (m >> S) | (m << (64 - S));
    else {
        constexpr unsigned long long field = (1ull << Z) - 1ull;
        constexpr unsigned long long low = replicate_field<Z, F>(field >> S);
        constexpr unsigned long long high = replicate_field<Z, F>(field & ~(field >> S));
        return ((m >> S) & low) | ((m << (Z - S)) & high);
    }
}

template <class C, int F>
__device__ __forceinline__ bool r1_slot_ok(const unsigned long long (&hm)[C::NB], unsigned long long slot_mask)
{
    unsigned long long bad = 0ull;
    static_for<0, C::MB>([&](auto br) {
        constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0;
        unsigned long long acc = 0ull;
        static_for<0, D>([&](auto p) {
            acc ^= rot_fields<C::Z, F, C::row_shift[e0 + p]>(hm[C::row_col[e0 + p]]);
        });
        bad |= acc;
    });
    return (bad & slot_mask) == 0ull;
}


// Check node of the tanh rules R2 / R3 for all D inputs of a check at once (bpsk.py:325-339, channel.py:471-474):
//     out_p = c * 2 atanh(clip(prod over j != p of tanh(clip(v_j / 2)))),   c = 1 (R2) or the scale (R3).
// With e_j = exp(-|v_j|): tanh(|v_j| / 2) = (1 - e_j) / (1 + e_j) = a_j / b_j, and for N = prod a_j, M = prod b_j
// over the other inputs 2 atanh(N / M) = log((M + N) / (M - N)): one exponential, one division and one logarithm
// per edge where tanh and atanh evaluated one after the other take an exponential, a logarithm, two divisions and
// two small-argument series.  The inner clip |v_j / 2| <= c is |v_j| <= 2 c, the product clip N / M <= lim is
// N <= lim * M.  Signs travel apart as sign bits; a zero input makes N = 0 and the message +-0, as tanh(0) = 0
// does in the reference.  Against the reference's own evaluation the results differ by rounding only (1 - e_j
// carries an absolute error of one ulp of 1, so messages near zero are exact to that, not to their own ulp).
template <typename T, int D, int NV>
__device__ __forceinline__ void spa_check_tanh(const T (&v)[NV], T (&out)[NV], T clip_arg, T clip_prod, T scale, bool scaled)
{
    if constexpr (D > 0) {
        T a[D], b[D], sa[D], sb[D];
        unsigned sx = 0;
        const T lim = clip_prod < Lim<T>::below_one() ? clip_prod : Lim<T>::below_one();
        static_for<0, D>([&](auto p) {
            T m = t_abs(v[p]);
            if (clip_arg > (T)0) m = m < (T)2 * clip_arg ? m : (T)2 * clip_arg;
            // saturated, infinite or NaN inputs: e^-700 is far below an ulp of 1 already, and the range reduction of the
            // double-precision exponential is built for finite arguments (the hardware exp2 of the float path is not)
            if constexpr (sizeof(T) == 8) m = m < (T)700 ? m : (T)700;
            const T e = f_exp(-m);
            a[p] = (T)1 - e;
            b[p] = (T)1 + e;
            sx ^= SNum<T>::sign_word(v[p]);
        });
        T ra = (T)1, rb = (T)1;
        static_for<0, D>([&](auto pp) {
            constexpr int p = D - 1 - pp;
            sa[p] = ra; sb[p] = rb;
            ra *= a[p]; rb *= b[p];
        });
        T pa = (T)1, pb = (T)1;
        T k;  // float: the logarithm comes to base two, its factor log(2) joins the scale
        if constexpr (sizeof(T) == 4) k = scaled ? scale * 0.693147181f : 0.693147181f;
        else k = scale;
        static_for<0, D>([&](auto p) {
            T n = pa * sa[p];
            const T m = pb * sb[p];
            pa *= a[p]; pb *= b[p];
            const T cap = lim * m;
            n = n < cap ? n : cap;
            const T r = f_div(m + n, m - n);
            T mag;
            if constexpr (sizeof(T) == 4) mag = k * __builtin_amdgcn_logf(r);
            else { mag = f_log(r); if (scaled) mag = k * mag; }
            out[p] = SNum<T>::with_sign(mag, sx ^ SNum<T>::sign_word(v[p]));
        });
    }
}

}  // namespace ldpc
