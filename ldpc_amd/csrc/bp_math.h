// Device arithmetic shared by the edge-list and the circulant kernels.
//
// Every helper reproduces one expression of the reference decoders; the file:line it follows is
// given at each helper.  T is float or double; with T = double and rules R0/R1 the results are
// the reference's float64 values bit for bit (same operations in the same order).
#pragma once
#include <hip/hip_runtime.h>

#include <limits>

namespace ldpc {

template <typename T> struct Lim;
template <> struct Lim<float> {
    __device__ static float inf() { return __builtin_huge_valf(); }
    // largest float below 1: 0.999999999 and 0.99999999 (bpsk.py:337, channel.py:473) both round
    // to 1.0f, whose atanh is infinite, so float arithmetic saturates one ulp lower
    __device__ static float below_one() { return 0.99999994f; }
    // exp(x) overflows float above ~88.7 and (inf+1)/(inf-1) is NaN; phi is exactly 0 from
    // x ~ 17 on in float, so capping the argument at 80 changes no result
    __device__ static float phi_arg_max() { return 80.0f; }
};
template <> struct Lim<double> {
    __device__ static double inf() { return __builtin_huge_val(); }
    __device__ static double below_one() { return 0.99999999999999989; }
    __device__ static double phi_arg_max() { return 100.0; }  // channel.py:396
};

__device__ inline float t_abs(float x) { return fabsf(x); }
__device__ inline double t_abs(double x) { return fabs(x); }
__device__ inline float t_tanh(float x) { return tanhf(x); }
__device__ inline double t_tanh(double x) { return tanh(x); }
__device__ inline float t_atanh(float x) { return atanhf(x); }
__device__ inline double t_atanh(double x) { return atanh(x); }
__device__ inline float t_exp(float x) { return expf(x); }
__device__ inline double t_exp(double x) { return exp(x); }
__device__ inline float t_log(float x) { return logf(x); }
__device__ inline double t_log(double x) { return log(x); }
__device__ inline float t_fmod(float x, float y) { return fmodf(x, y); }
__device__ inline double t_fmod(double x, double y) { return fmod(x, y); }
// `x % 2 == 0` of the reference's iteration-0 syndrome on raw LLR sums (ldpc.py:79 on floats): true
// exactly when x is finite and an even integer (+-0 included), which is what fmod(x, 2) == 0 says.
// Halving is exact, so x is an even integer iff twice the truncated half gives x back (a tiny x
// whose half rounds to zero fails the comparison, as it should); three instructions where the
// library's fmod is a loop.
__device__ inline bool even_integer(float x) { const float t = truncf(x * 0.5f); return t + t == x && fabsf(x) < __builtin_inff(); }
__device__ inline bool even_integer(double x) { const double t = trunc(x * 0.5); return t + t == x && fabs(x) < __builtin_inf(); }

template <typename T>
__device__ inline T clip(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }

// ---------------------------------------------------------------------------------------------
// Transcendentals of the sum-product rules.  The library versions spend 60-150 instructions per
// call on cases these decoders never produce (huge arguments, denormals, exact rounding); the
// versions below are plain range reduction + polynomial, a few ulp in double and hardware
// exp2/log2 in float.  The parity bar for these rules is 1e-4 relative on posteriors with
// identical decisions and iteration counts, which is many orders of magnitude looser.
// Compile with -DLDPC_LIBM_TRANSCENDENTALS to get the library functions back.
#ifndef LDPC_LIBM_TRANSCENDENTALS
// a / b, b normal and not tiny: reciprocal seed + two Newton steps + one residual correction
__device__ inline double f_div(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}
__device__ inline float f_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
// e^x for |x| < 700
__device__ inline double f_exp(double x) {
    const double n = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
    double p = 2.08767569878680989792e-09;  // 1/12!
    p = __builtin_fma(p, r, 2.50521083854417187751e-08);
    p = __builtin_fma(p, r, 2.75573192239858906526e-07);
    p = __builtin_fma(p, r, 2.75573192239858906526e-06);
    p = __builtin_fma(p, r, 2.48015873015873015873e-05);
    p = __builtin_fma(p, r, 1.98412698412698412698e-04);
    p = __builtin_fma(p, r, 1.38888888888888888889e-03);
    p = __builtin_fma(p, r, 8.33333333333333333333e-03);
    p = __builtin_fma(p, r, 4.16666666666666666667e-02);
    p = __builtin_fma(p, r, 1.66666666666666666667e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}
__device__ inline float f_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504f); }
// log(x), x normal and positive
__device__ inline double f_log(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double s = f_div(m - 1.0, m + 1.0);  // |s| <= 0.1716
    const double z = s * s;
    double p = 1.0 / 23.0;
    p = __builtin_fma(p, z, 1.0 / 21.0);
    p = __builtin_fma(p, z, 1.0 / 19.0);
    p = __builtin_fma(p, z, 1.0 / 17.0);
    p = __builtin_fma(p, z, 1.0 / 15.0);
    p = __builtin_fma(p, z, 1.0 / 13.0);
    p = __builtin_fma(p, z, 1.0 / 11.0);
    p = __builtin_fma(p, z, 1.0 / 9.0);
    p = __builtin_fma(p, z, 1.0 / 7.0);
    p = __builtin_fma(p, z, 1.0 / 5.0);
    p = __builtin_fma(p, z, 1.0 / 3.0);
    p = p * z;                                  // log(m) = 2 s (1 + p)
    const double lm = __builtin_fma(s + s, p, s + s);
    const double de = (double)e;
    return __builtin_fma(de, 6.93147180369123816490e-01, __builtin_fma(de, 1.90821492927058770002e-10, lm));
}
__device__ inline float f_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147181f; }
// tanh: odd series below 2^-6 (double) / 0.06 (float), (1 - e^-2|x|) / (1 + e^-2|x|) above
__device__ inline double f_tanh(double x) {
    const double ax = __builtin_fabs(x), x2 = x * x;
    const double small = x * __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, -17.0 / 315.0, 2.0 / 15.0), -1.0 / 3.0), 1.0);
    const double e = f_exp(-2.0 * (ax < 40.0 ? ax : 40.0));
    const double big = __builtin_copysign(f_div(1.0 - e, 1.0 + e), x);
    return ax < 0.015625 ? small : big;
}
__device__ inline float f_tanh(float x) {
    const float ax = __builtin_fabsf(x), x2 = x * x;
    const float small = x * __builtin_fmaf(x2, __builtin_fmaf(x2, 2.0f / 15.0f, -1.0f / 3.0f), 1.0f);
    const float e = f_exp(-2.0f * (ax < 20.0f ? ax : 20.0f));
    const float big = __builtin_copysignf(f_div(1.0f - e, 1.0f + e), x);
    return ax < 0.06f ? small : big;
}
// atanh for |x| < 1: odd series below 2^-6 (double) / 0.06 (float), log((1 + |x|) / (1 - |x|)) / 2 above
__device__ inline double f_atanh(double x) {
    const double ax = __builtin_fabs(x), x2 = x * x;
    double p = 1.0 / 9.0;
    p = __builtin_fma(p, x2, 1.0 / 7.0);
    p = __builtin_fma(p, x2, 1.0 / 5.0);
    p = __builtin_fma(p, x2, 1.0 / 3.0);
    const double small = __builtin_fma(x * x2, p, x);
    const double big = __builtin_copysign(0.5 * f_log(f_div(1.0 + ax, 1.0 - ax)), x);
    return ax < 0.015625 ? small : big;
}
__device__ inline float f_atanh(float x) {
    const float ax = __builtin_fabsf(x), x2 = x * x;
    const float small = __builtin_fmaf(x * x2, __builtin_fmaf(x2, 0.2f, 1.0f / 3.0f), x);
    const float big = __builtin_copysignf(0.5f * f_log(f_div(1.0f + ax, 1.0f - ax)), x);
    return ax < 0.06f ? small : big;
}
#else
__device__ inline double f_div(double a, double b) { return a / b; }
__device__ inline float f_div(float a, float b) { return a / b; }
__device__ inline double f_exp(double x) { return exp(x); }
__device__ inline float f_exp(float x) { return expf(x); }
__device__ inline double f_log(double x) { return log(x); }
__device__ inline float f_log(float x) { return logf(x); }
__device__ inline double f_tanh(double x) { return tanh(x); }
__device__ inline float f_tanh(float x) { return tanhf(x); }
__device__ inline double f_atanh(double x) { return atanh(x); }
__device__ inline float f_atanh(float x) { return atanhf(x); }
#endif

// Running state of a min-sum check node: smallest and second smallest magnitude, position of the
// smallest, sign bits.  Equivalent to the reference's per-target min over the other inputs
// (ldpc.py:184-187, bpsk.py:221-230): the target holding the smallest magnitude gets the second
// smallest, everyone else the smallest; ties give equal values either way.
template <typename T>
struct MinSumState {
    T min1, min2;
    int idx;
    unsigned long long neg;  // bit p set when input p is negative
    __device__ void init() { min1 = Lim<T>::inf(); min2 = Lim<T>::inf(); idx = -1; neg = 0ull; }
    __device__ void add(int p, T v) {
        T a = t_abs(v);
        if (a < min1) { min2 = min1; min1 = a; idx = p; }
        else if (a < min2) { min2 = a; }
        neg |= (unsigned long long)(v < (T)0) << p;
    }
    // product of the other inputs' signs is -1 when an odd number of them is negative.  A zero
    // input makes the other targets' magnitude 0, which is what numpy.sign(0) = 0 yields in
    // ldpc.py:186; bpsk.py:227 maps sign 0 to +1, and 0 is not negative here either.
    __device__ bool negative_excluding(int p) const {
        return (__popcll(neg & ~(1ull << p)) & 1) != 0;
    }
    __device__ T magnitude_excluding(int p) const { return p == idx ? min2 : min1; }
};

// R0 message: minLLR * messageSign (ldpc.py:187)
template <typename T>
__device__ inline T r0_message(const MinSumState<T> &s, int p) {
    T mag = s.magnitude_excluding(p);
    return s.negative_excluding(p) ? -mag : mag;
}
// R1 message: 0.75 * minLLR * signProd, evaluated left to right (bpsk.py:233)
template <typename T>
__device__ inline T r1_message(const MinSumState<T> &s, int p, T scale) {
    T mag = scale * s.magnitude_excluding(p);
    return s.negative_excluding(p) ? -mag : mag;
}

// Offset min-sum magnitude (build extension, LDPC_RULE_OMS): max(scale * m - offset, 0), in that order
template <typename T>
__device__ inline T oms_magnitude(T m, T scale, T offset) {
    const T t = scale * m - offset;
    return t > (T)0 ? t : (T)0;
}
template <typename T>
__device__ inline T oms_message(const MinSumState<T> &s, int p, T scale, T offset) {
    T mag = oms_magnitude(s.magnitude_excluding(p), scale, offset);
    return s.negative_excluding(p) ? -mag : mag;
}

// tanh(clip(q / 2, -c, c)) (bpsk.py:333; channel.py:471 has no clip, c <= 0)
template <typename T>
__device__ inline T spa_tanh_half(T q, T clip_arg) {
    T a = q / (T)2;
    if (clip_arg > (T)0) a = clip(a, -clip_arg, clip_arg);
    return f_tanh(a);
}
// R2: 2*arctanh(clip(prod)) (bpsk.py:337-339); R3: 0.85*2*arctanh(clip(prod)) (channel.py:473-474)
template <typename T>
__device__ inline T spa_message(T prod, T clip_prod, T scale, bool scaled) {
    T lim = clip_prod < Lim<T>::below_one() ? clip_prod : Lim<T>::below_one();
    prod = clip(prod, -lim, lim);
    return scaled ? (scale * (T)2) * f_atanh(prod) : (T)2 * f_atanh(prod);
}
// phi(x) = log((e^x + 1) / (e^x - 1)) on x clipped to [1e-12, 100] (channel.py:392-398)
template <typename T>
__device__ inline T phi(T x) {
    x = clip(x, (T)1e-12, Lim<T>::phi_arg_max());
    T ex = f_exp(x);
    return f_log(f_div(ex + (T)1, ex - (T)1));
}
// R4 message magnitude: phi(clip(sum of phi, 1e-12, 20)) (channel.py:347-349)
template <typename T>
__device__ inline T phi_message(T phi_sum, bool negative) {
    T mag = phi(clip(phi_sum, (T)1e-12, (T)20));
    return negative ? -mag : mag;
}

// hard decision: R0 `0 if b < 0 else 1` (ldpc.py:198-201); R1-R4 `0 if b > 0 else 1` (bpsk.py:240-243)
template <int RULE, typename T>
__device__ inline unsigned hard_bit(T b) {
    return RULE == 0 ? (b < (T)0 ? 0u : 1u) : (b > (T)0 ? 0u : 1u);  // the build's extra rules (5, 6) follow R1
}

}  // namespace ldpc
