// Circulant decoder for rule R0 (ldpc.py:137-227), float or double, with compressed check state.
//
// R0 is not extrinsic: every check reads the posteriors b of its variables, and its message to
// variable i is  sign-product-of-the-others * min-magnitude-of-the-others  (ldpc.py:184-187).
// That message is a function of three numbers of the check (smallest magnitude m1, second
// smallest m2, parity of the signs) and of b_i itself:
//      |r| = m2 if |b_i| == m1 else m1        sign(r) = parity xor sign(b_i)
// (if two inputs tie for the smallest, m2 == m1 and both forms agree).  So the check-node phase
// stores only (m1, m2), both with the parity in their sign bit, per check, and the variable-node phase
// rebuilds each message while summing in ascending check order (ldpc.py:173,188), then forms
// b = 0.75 * (S + L) (ldpc.py:196).  Per frame the LDS holds b, L and the check pairs: 12 KiB for
// the CCSDS (512,256) code in double precision instead of 26 KiB with one message per edge.
//
// Thread mapping, slots and the circulant tables are those of decode_qc.hip.  The tables are read
// through the constant address space so that they arrive by scalar loads.
//
// Sign handling: channel LLRs are stored as L + 0.0, so no value in the decoder is ever -0.0
// (sums start from +0.0) and "negative" is exactly "sign bit set"; the hard decision
// `0 if b < 0 else 1` (ldpc.py:198-201) is then the inverted sign bit.
#include <cstdlib>

#include "bp_common.h"
#include "bp_math.h"

namespace ldpc {

typedef const unsigned __attribute__((address_space(4))) *const_u32;
typedef const int __attribute__((address_space(4))) *const_i32;
__device__ __forceinline__ const_u32 as_const(const unsigned *p) { return (const_u32)(unsigned long long)p; }
__device__ __forceinline__ const_i32 as_const(const int *p) { return (const_i32)(unsigned long long)p; }

template <typename T> struct Num;
template <> struct Num<float> {
    typedef float2 Pair;
    __device__ static __forceinline__ unsigned sign_word(float x) { return __float_as_uint(x); }
    __device__ static __forceinline__ float with_sign(float mag, unsigned s) {
        return __uint_as_float(__float_as_uint(mag) | (s & 0x80000000u));
    }
    // min(m, |b|); the instruction is written out because the builtin inserts a quieting
    // canonicalisation in front of every operand that comes from memory
    __device__ static __forceinline__ float flip_sign(float x, unsigned signbit) {
        return __uint_as_float(__float_as_uint(x) ^ signbit);
    }
    __device__ static __forceinline__ float min_abs(float m, float b) {
        float d; asm("v_min_f32 %0, %1, |%2|" : "=v"(d) : "v"(m), "v"(b)); return d;
    }
    // second smallest after seeing b: median of (m1, m2, |b|) given m1 <= m2
    __device__ static __forceinline__ float second(float m1, float m2, float b) {
        float d; asm("v_med3_f32 %0, %1, %2, |%3|" : "=v"(d) : "v"(m1), "v"(m2), "v"(b)); return d;
    }
    __device__ static __forceinline__ Pair make(float a, float b) { return make_float2(a, b); }
};
template <> struct Num<double> {
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
    __device__ static __forceinline__ Pair make(double a, double b) { return make_double2(a, b); }
};

struct R0Ctx {
    int Z, FZ, mb, nb;
    int tid, fz0, z;  // tid = f*Z + z, fz0 = f*Z
    int w0, wstep;    // this wave takes block rows / block columns w0, w0 + wstep, ...
    int mact;         // checks with index >= mact never send a message (ldpc.py:30,175: the loop runs to int(self.m))
};

// rotated index inside the slot: (z + s) mod Z without a compare (unsigned wrap-around)
__device__ __forceinline__ int rot_add(int z, int s, int Z) {
    const unsigned t = (unsigned)(z + s);
    return (int)min(t, t - (unsigned)Z);
}
__device__ __forceinline__ int rot_sub(int z, int s, int Z) {
    const unsigned t = (unsigned)(z - s);
    return (int)min(t, t + (unsigned)Z);
}

// ---- one check of degree D: summary (m1 | parity, m2); returns 1 if the check fails
template <typename T, int D, bool FLOATCHK>
__device__ __forceinline__ int r0_check(const R0Ctx &x, const_u32 info, int e0, const T *post,
                                        typename Num<T>::Pair *cs, int br)
{
    T m1 = Lim<T>::inf(), m2 = Lim<T>::inf();
    unsigned sx = 0;
    T fsum = (T)0;
#pragma unroll
    for (int p = 0; p < D; ++p) {
        const unsigned w = info[e0 + p];
        const int bc = (int)(w >> 16), s = (int)(w & 0xfffu);
        const T b = post[bc * x.FZ + x.fz0 + rot_add(x.z, s, x.Z)];
        m2 = Num<T>::second(m1, m2, b);
        m1 = Num<T>::min_abs(m1, b);
        sx ^= Num<T>::sign_word(b);
        if (FLOATCHK) fsum += b;
    }
    if (D < 2) { m1 = (T)0; m2 = (T)0; }  // ldpc.py:178: a check with one variable sends nothing
    unsigned sp = sx;
    if (br * x.Z + x.z >= x.mact) { m1 = (T)0; m2 = (T)0; sp = 0; }  // beyond int(self.m): its messages stay 0.0
    cs[br * x.FZ + x.tid] = Num<T>::make(Num<T>::with_sign(m1, sp), Num<T>::with_sign(m2, sp));
    if (FLOATCHK) return !even_integer(fsum);  // ldpc.py:158,165 on the raw LLRs
    // parity of the hard decisions: hard = 1 - signbit, so the XOR is (D - #negative) mod 2
    return (int)(((unsigned)D ^ (sx >> 31)) & 1u);
}

// ---- one variable of degree D: rebuild its messages, sum ascending, new posterior
template <typename T, int D>
__device__ __forceinline__ void r0_var(const R0Ctx &x, T scale, const_u32 info, int k0, int bc,
                                       T *post, const T *L, const typename Num<T>::Pair *cs)
{
    const int vi = bc * x.FZ + x.tid;
    const T own = post[vi];
    const unsigned so = Num<T>::sign_word(own) & 0x80000000u;
    const T ao = t_abs(own);
    T r[D > 0 ? D : 1];
    int zz[D > 0 ? D : 1];
    bool pair[D > 0 ? D : 1];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const unsigned w = info[k0 + k];
        const int br = (int)(w >> 16), s = (int)(w & 0xfffu);
        pair[k] = (w & 0x8000u) != 0;
        zz[k] = rot_sub(x.z, s, x.Z);
        const typename Num<T>::Pair c = cs[br * x.FZ + x.fz0 + zz[k]];
        r[k] = ao == t_abs(c.x) ? c.y : c.x;  // candidates carry the parity; the variable's own sign goes on the sum
    }
#pragma unroll
    for (int k = 0; k + 1 < D; ++k) {
        if (pair[k]) {  // two circulants in one block: lower check index first
            const bool sw = zz[k] > zz[k + 1];
            const T a = r[k], b = r[k + 1];
            r[k] = sw ? b : a;
            r[k + 1] = sw ? a : b;
        }
    }
    T s = (T)0;
#pragma unroll
    for (int k = 0; k < D; ++k) s += r[k];
    post[vi] = scale * (Num<T>::flip_sign(s, so) + L[vi]);
}

#define R0_CASE_ROW(N)                                                                              \
    case N:                                                                                         \
        if constexpr (N <= DC) bad |= r0_check<T, N, FLOATCHK>(x, row_info, e0, post, cs, br);      \
        break;
#define R0_CASE_COL(N)                                                                              \
    case N:                                                                                         \
        if constexpr (N <= DV) r0_var<T, N>(x, scale, col_info, k0, bc, post, L, cs);               \
        break;

template <typename T, int DC, bool FLOATCHK>
__device__ __forceinline__ int r0_check_phase(const R0Ctx &x, const_i32 row_ptr, const_u32 row_info,
                                              const T *post, typename Num<T>::Pair *cs)
{
    int bad = 0;
    for (int br = x.w0; br < x.mb; br += x.wstep) {
        const int e0 = row_ptr[br], d = row_ptr[br + 1] - e0;
        switch (d) {
            R0_CASE_ROW(1) R0_CASE_ROW(2) R0_CASE_ROW(3) R0_CASE_ROW(4) R0_CASE_ROW(5)
            R0_CASE_ROW(6) R0_CASE_ROW(7) R0_CASE_ROW(8) R0_CASE_ROW(9) R0_CASE_ROW(10)
            R0_CASE_ROW(11) R0_CASE_ROW(12) R0_CASE_ROW(13) R0_CASE_ROW(14) R0_CASE_ROW(15)
            R0_CASE_ROW(16) R0_CASE_ROW(17) R0_CASE_ROW(18) R0_CASE_ROW(19) R0_CASE_ROW(20)
            default: break;
        }
    }
    return bad;
}

template <typename T, int DV>
__device__ __forceinline__ void r0_var_phase(const R0Ctx &x, T scale, const_i32 col_ptr, const_u32 col_info,
                                             T *post, const T *L, const typename Num<T>::Pair *cs)
{
    for (int bc = x.w0; bc < x.nb; bc += x.wstep) {
        const int k0 = col_ptr[bc], d = col_ptr[bc + 1] - k0;
        switch (d) {
            case 0: r0_var<T, 0>(x, scale, col_info, k0, bc, post, L, cs); break;
            R0_CASE_COL(1) R0_CASE_COL(2) R0_CASE_COL(3) R0_CASE_COL(4) R0_CASE_COL(5)
            R0_CASE_COL(6) R0_CASE_COL(7) R0_CASE_COL(8) R0_CASE_COL(9) R0_CASE_COL(10)
            R0_CASE_COL(11) R0_CASE_COL(12) R0_CASE_COL(13) R0_CASE_COL(14) R0_CASE_COL(15)
            R0_CASE_COL(16) R0_CASE_COL(17) R0_CASE_COL(18) R0_CASE_COL(19) R0_CASE_COL(20)
            R0_CASE_COL(21) R0_CASE_COL(22) R0_CASE_COL(23) R0_CASE_COL(24) R0_CASE_COL(25)
            R0_CASE_COL(26) R0_CASE_COL(27) R0_CASE_COL(28) R0_CASE_COL(29) R0_CASE_COL(30)
            R0_CASE_COL(31) R0_CASE_COL(32)
            default: break;
        }
    }
}

// ------------------------------------------------------------------------------------------
template <typename T, typename TL, int DC, int DV>
__global__ void __launch_bounds__(512)
decode_qc_r0_kernel(QcDev q, Cfg<T> g, int F, int W, const TL *__restrict__ llr, long long frames,
                    uint8_t *__restrict__ hard_out, int *__restrict__ iters_out,
                    uint8_t *__restrict__ conv_out, T *__restrict__ post_out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    typedef typename Num<T>::Pair Pair;
    R0Ctx x;
    x.Z = q.Z; x.FZ = F * q.Z; x.mb = q.mb; x.nb = q.nb;
    x.mact = g.num_active_checks < 0 ? q.mb * q.Z : g.num_active_checks;
    // thread mapping as in decode_qc.hip: a wave covers 64 z of one slot (or 64/Z slots side by side),
    // W wave sets split the block rows / columns of a slot group, G groups share the workgroup
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int S = q.Z >= 64 ? 1 : 64 / q.Z, Zw = (q.Z + 63) / 64;
    const int group = wave / (W * Zw), within = wave % (W * Zw);
    const int sub = q.Z >= 64 ? 0 : lane / q.Z;
    x.w0 = within / Zw;
    x.wstep = W;
    x.z = q.Z >= 64 ? (within % Zw) * 64 + lane : lane - sub * q.Z;
    const bool lane_on = q.Z >= 64 ? x.z < q.Z : sub < S;
    const int f = lane_on ? group * S + sub : 0;
    if (!lane_on) x.z = 0;
    x.fz0 = f * x.Z;
    x.tid = x.fz0 + x.z;
    const bool reporter = lane_on && x.z == 0 && x.w0 == 0;
    const int n = q.nb * q.Z;
    const const_i32 row_ptr = as_const(q.row_ptr), col_ptr = as_const(q.col_ptr);
    const const_u32 row_info = as_const(q.row_info), col_info = as_const(q.colr_info);

    Pair *cs = reinterpret_cast<Pair *>(smem);
    T *post = reinterpret_cast<T *>(cs + (size_t)q.mb * x.FZ);
    T *L = post + (size_t)q.nb * x.FZ;
    int *slot_bad = reinterpret_cast<int *>(L + (size_t)q.nb * x.FZ);

    const long long groups = (frames + F - 1) / F;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long frame = grp * F + f;
        const bool active = lane_on && frame < frames;
        if (lane_on) {
            for (int bc = x.w0; bc < x.nb; bc += x.wstep) {
                const int vi = bc * x.FZ + x.tid;
                const T l = (active ? (T)llr[frame * n + bc * x.Z + x.z] : (T)0) + (T)0;  // -0.0 -> +0.0
                L[vi] = l;
                post[vi] = l;  // ldpc.py:154
            }
        }
        if ((int)threadIdx.x < F) slot_bad[threadIdx.x] = 0;
        __syncthreads();

        bool done = !active;
        int t = 0;
        for (; t < g.max_iters; ++t) {
            const bool float_check = t == 0 && g.float_syndrome0 && g.early_exit;
            if (!done) {
                int bad = float_check ? r0_check_phase<T, DC, true>(x, row_ptr, row_info, post, cs)
                                      : r0_check_phase<T, DC, false>(x, row_ptr, row_info, post, cs);
                if (bad && g.early_exit) slot_bad[f] = 1;
            }
            __syncthreads();
            if (g.early_exit) {
                if (!done && !slot_bad[f]) {  // ldpc.py:165-170
                    for (int bc = x.w0; bc < x.nb; bc += x.wstep) {
                        const T v = post[bc * x.FZ + x.tid];
                        hard_out[frame * n + bc * x.Z + x.z] =
                            float_check ? (v == (T)0 ? 0 : (v == (T)1 ? 1 : 2)) : (uint8_t)hard_bit<0>(v);
                        if (post_out) post_out[frame * n + bc * x.Z + x.z] = v;
                    }
                    if (reporter) { iters_out[frame] = t; conv_out[frame] = 1; }
                    done = true;
                }
                if (!__syncthreads_or(!done)) break;
                if ((int)threadIdx.x < F) slot_bad[threadIdx.x] = 0;
            }
            if (!done) r0_var_phase<T, DV>(x, g.scale, col_ptr, col_info, post, L, cs);
            __syncthreads();
        }
        if (!g.early_exit) {  // fixed work: report the syndrome of the final hard decisions
            int bad = 0;
            if (!done) bad = r0_check_phase<T, DC, false>(x, row_ptr, row_info, post, cs);
            if (bad) slot_bad[f] = 1;
            __syncthreads();
        }
        if (!done) {  // ldpc.py:210-227
            for (int bc = x.w0; bc < x.nb; bc += x.wstep) {
                const T v = post[bc * x.FZ + x.tid];
                hard_out[frame * n + bc * x.Z + x.z] = (uint8_t)hard_bit<0>(v);
                if (post_out) post_out[frame * n + bc * x.Z + x.z] = v;
            }
            if (reporter) {
                iters_out[frame] = g.early_exit ? g.max_iters - 1 : g.max_iters;
                conv_out[frame] = g.early_exit ? 0 : (uint8_t)!slot_bad[f];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
static const size_t kLdsLimit = 160 * 1024;

template <typename T>
static size_t r0_lds_bytes(const QcDev &q, int F)
{
    const size_t FZ = (size_t)F * q.Z;
    return (size_t)q.mb * FZ * 2 * sizeof(T) + (size_t)q.nb * FZ * 2 * sizeof(T) + (size_t)F * sizeof(int) + 16;
}

struct R0Shape { int F, W, threads; };
static R0Shape pick_shape(const QcDev &q, size_t bytes_per_slot)
{
    const int S = q.Z >= 64 ? 1 : 64 / q.Z, Zw = (q.Z + 63) / 64;
    int G = 1;
    while (G < 4 && (size_t)(G + 1) * S * bytes_per_slot <= kLdsLimit / 2 && (G + 1) * Zw <= 8) ++G;
    int W = 8 / (G * Zw);
    if (W < 1) W = 1;
    const int most = q.mb < q.nb ? q.mb : q.nb;
    if (W > most) W = most;
    return R0Shape{G * S, W, G * W * Zw * 64};
}

template <typename T, typename TL, int DC, int DV>
static hipError_t launch_r0_t(const QcDev &q, const ldpc_decode_config &cfg, const Launch &io,
                              int num_cus, const char **name, bool *supported)
{
    const int S = q.Z >= 64 ? 1 : 64 / q.Z;
    const size_t per_slot = r0_lds_bytes<T>(q, 1);
    const R0Shape sh = pick_shape(q, per_slot);
    const size_t lds = r0_lds_bytes<T>(q, sh.F);
    if ((size_t)S * per_slot > kLdsLimit || lds > kLdsLimit || sh.threads > 512) { *supported = false; return hipSuccess; }
    *supported = true;
    const int F = sh.F;
    size_t per_cu = kLdsLimit / lds;
    const size_t wave_cap = 32 / (sh.threads / 64);
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu < 1) per_cu = 1;
    const long long groups = (io.frames + F - 1) / F;
    long long grid = (long long)num_cus * (long long)per_cu * 4;
    if (grid > groups) grid = groups;
    if (grid < 1) grid = 1;
    Cfg<T> g;
    g.rule = 0; g.max_iters = cfg.max_iters; g.early_exit = cfg.early_exit;
    g.float_syndrome0 = cfg.float_syndrome0; g.num_active_checks = cfg.num_active_checks; g.num_refresh_vars = -1;
    g.scale = (T)cfg.scale; g.clip_tanh_arg = (T)0; g.clip_prod = (T)0; g.clip_post = (T)0; g.offset = (T)0;
    auto k = decode_qc_r0_kernel<T, TL, DC, DV>;
    hipError_t e = kernel_setup(reinterpret_cast<const void *>(k), sh.threads, lds, nullptr);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(sh.threads), lds, io.stream, q, g, F, sh.W,
                       reinterpret_cast<const TL *>(io.llr), io.frames, io.hard, io.iters,
                       io.converged, reinterpret_cast<T *>(io.posterior));
    *name = sizeof(T) == 8 ? "qc_r0_f64" : "qc_r0_f32";
    return hipGetLastError();
}

template <typename T, typename TL>
static hipError_t launch_r0_deg(const QcDev &q, const ldpc_decode_config &cfg, const Launch &io,
                                int num_cus, const char **name, bool *supported)
{
    if (q.max_dc <= 8 && q.max_dv <= 12)
        return launch_r0_t<T, TL, 8, 12>(q, cfg, io, num_cus, name, supported);
    if (q.max_dc <= 10 && q.max_dv <= 24)
        return launch_r0_t<T, TL, 10, 24>(q, cfg, io, num_cus, name, supported);
    if (q.max_dc <= 20 && q.max_dv <= 32)
        return launch_r0_t<T, TL, 20, 32>(q, cfg, io, num_cus, name, supported);
    *supported = false;
    return hipSuccess;
}

hipError_t launch_qc_rule_0(const QcDev &q, const ldpc_decode_config &cfg, int compute_dtype,
                            const Launch &io, int num_cus, const char **name, bool *supported)
{
    if (compute_dtype == LDPC_F64) {
        if (io.llr_dtype == LDPC_F64) return launch_r0_deg<double, double>(q, cfg, io, num_cus, name, supported);
        return launch_r0_deg<double, float>(q, cfg, io, num_cus, name, supported);
    }
    if (io.llr_dtype == LDPC_F32) return launch_r0_deg<float, float>(q, cfg, io, num_cus, name, supported);
    *supported = false;
    return hipSuccess;
}

}  // namespace ldpc
