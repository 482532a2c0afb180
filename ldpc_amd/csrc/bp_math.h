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

template <typename T>
__device__ inline T clip(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }

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

// tanh(clip(q / 2, -c, c)) (bpsk.py:333; channel.py:471 has no clip, c <= 0)
template <typename T>
__device__ inline T spa_tanh_half(T q, T clip_arg) {
    T a = q / (T)2;
    if (clip_arg > (T)0) a = clip(a, -clip_arg, clip_arg);
    return t_tanh(a);
}
// R2: 2*arctanh(clip(prod)) (bpsk.py:337-339); R3: 0.85*2*arctanh(clip(prod)) (channel.py:473-474)
template <typename T>
__device__ inline T spa_message(T prod, T clip_prod, T scale, bool scaled) {
    T lim = clip_prod < Lim<T>::below_one() ? clip_prod : Lim<T>::below_one();
    prod = clip(prod, -lim, lim);
    return scaled ? (scale * (T)2) * t_atanh(prod) : (T)2 * t_atanh(prod);
}
// phi(x) = log((e^x + 1) / (e^x - 1)) on x clipped to [1e-12, 100] (channel.py:392-398)
template <typename T>
__device__ inline T phi(T x) {
    x = clip(x, (T)1e-12, Lim<T>::phi_arg_max());
    T ex = t_exp(x);
    return t_log((ex + (T)1) / (ex - (T)1));
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
    return RULE == 0 ? (b < (T)0 ? 0u : 1u) : (b > (T)0 ? 0u : 1u);
}

}  // namespace ldpc
