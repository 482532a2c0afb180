// The steps either side of the decode loop, as device functions that a decoder kernel can run
// itself ("fused sweep"): the BPSK + white-noise channel with its LLR formula in front
// (/root/reference/ldpc.py:142-146, 241-266), the error accounting behind
// (ldpc.py:166-170, 210-216; FER loop ldpc.py:644-667).  channel.hip builds its stand-alone kernel
// from the same functions, so a fused launch and the channel -> decode -> count sequence of three
// launches see identical LLRs and give identical counters.
//
// Noise: Philox4x32-10 keyed by the caller's seed, counter = (absolute frame index, position / 4),
// two Box-Muller pairs per counter value (the reference draws from numpy's global generator, which
// nothing can reproduce).
#pragma once
#include "bp_common.h"

namespace ldpc {

struct Philox4 { unsigned x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                                 unsigned k0, unsigned k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

// two standard normals from two 32-bit words: u1 in (0, 1), radius up to 6.7 sigma
__device__ __forceinline__ void box_muller(unsigned a, unsigned b, float &n0, float &n1)
{
    const float u1 = ((float)a + 0.5f) * 2.3283064365386963e-10f;  // 2^-32; never 0, may round to 1
    const float u2 = ((float)b + 0.5f) * 2.3283064365386963e-10f;
    // hardware log2 / sin / cos (the latter take their argument in turns): a few 1e-7 absolute,
    // far below the quantisation of the uniforms themselves
    const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1 < 1.0f ? u1 : 0.99999994f));
    n0 = r * __builtin_amdgcn_cosf(u2);
    n1 = r * __builtin_amdgcn_sinf(u2);
}

// Channel of one launch.  amp = sinc(cfo): loss of the integrate-and-dump filter under a carrier
// offset of cfo cycles per symbol (see ldpc_channel_llr_doppler).
struct ChannelDev {
    float sigma, gain, prefix_llr, cfo, phase0, amp;
    int plus_is_bit, prefix_len;
    unsigned k0, k1;
    unsigned long long first_frame;
};

// LLRs of positions 4 qd .. 4 qd + 3 of the frame with absolute index af; byte j of `bits` holds the
// transmitted bit of position 4 qd + j.
__device__ __forceinline__ void channel_quad(const ChannelDev &c, unsigned long long af, int qd, unsigned bits, float (&v)[4])
{
    const bool doppler = c.cfo != 0.0f || c.phase0 != 0.0f;  // the same for every thread
    const Philox4 rnd = philox4x32_10((unsigned)af, (unsigned)(af >> 32), (unsigned)qd, 0u, c.k0, c.k1);
    float z[4];
    box_muller(rnd.x, rnd.y, z[0], z[1]);
    box_muller(rnd.z, rnd.w, z[2], z[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pos = qd * 4 + j, b = (int)((bits >> (8 * j)) & 1u);
        float x = c.plus_is_bit ? (float)(2 * b - 1) : (float)(1 - 2 * b);
        if (doppler) {
            float turns = c.cfo * ((float)pos + 0.5f) + c.phase0;
            turns -= __builtin_floorf(turns);  // the hardware cosine takes turns, |argument| < 256
            x *= c.amp * __builtin_amdgcn_cosf(turns);
        }
        v[j] = pos < c.prefix_len ? c.prefix_llr : c.gain * (x + c.sigma * z[j]);
    }
}

// What a fused launch gets instead of LLR and result arrays.
struct SweepDev {
    ChannelDev ch;
    const uint8_t *ref;            // transmitted codewords [frames, n] (ref_stride = n), one word for all
    long long ref_stride;          //   frames (ref_stride = 0), or NULL = the all-zero word
    unsigned long long *counters;  // [5], added to atomically: see count_errors.hip
    int ref_packed;                // the words are bits, eight positions per byte, lowest bit first; ref_stride in bytes
};

// The transmitted bits of positions 4 qd .. 4 qd + 3 of a frame, one per byte of the result.  Every kernel
// reads the reference word here and nowhere else: what the accounting compares with is the copy the
// staging step keeps beside the LLRs.
__device__ __forceinline__ unsigned sweep_ref_quad(const SweepDev &sp, long long frame, int qd)
{
    if (!sp.ref) return 0u;
    if (sp.ref_packed) {
        const unsigned nib = ((unsigned)sp.ref[frame * sp.ref_stride + (qd >> 1)] >> ((qd & 1) * 4)) & 15u;
        return (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
    }
    return *reinterpret_cast<const unsigned *>(sp.ref + frame * sp.ref_stride + qd * 4) & 0x01010101u;
}

// Per-lane partial counters of a fused launch; summed over the workgroup and added to the
// counters in device memory once, when the workgroup has run out of frames.
struct SweepAcc {
    unsigned frames = 0, nonconv = 0, wrong = 0, biterr = 0, itsum = 0;
    // one frame: `errs` = this lane's share of the frame's bit errors, `any_err` = the frame has one
    // somewhere, `owner` = this lane books the per-frame counters (one lane per frame)
    __device__ __forceinline__ void book(unsigned errs, bool any_err, bool owner, int iters, int conv)
    {
        biterr += errs;
        if (owner) {
            frames += 1;
            nonconv += conv ? 0u : 1u;
            wrong += (conv && any_err) ? 1u : 0u;
            itsum += (unsigned)iters;
        }
    }
    // all lanes of a wave; lane 0 adds the wave's sums to device memory
    __device__ __forceinline__ void flush_wave(unsigned long long *counters)
    {
        unsigned v[5] = {frames, nonconv, wrong, biterr, itsum};
        unsigned long long s[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            // per-lane counts stay far below 2^26 (a workgroup's share of one launch): the wave sum fits 32 bits
            unsigned x = v[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            s[i] = x;
        }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (s[i]) atomicAdd(&counters[i], s[i]);
        }
    }
};

// Single-wave workgroups: the channel LLRs of the F frames of group `grp` into an LDS staging
// area indexed by position ([F][n] floats), and the transmitted bits beside it ([F][n] bytes).
// n is a multiple of four for every compiled-in code.
template <int N, int F>
__device__ __forceinline__ void sweep_stage_wave(const SweepDev &sp, float *stage, uint8_t *refstage, long long grp,
                                                 long long frames, int lane)
{
    static_assert(N % 4 == 0, "positions are generated four at a time");
    constexpr int QUADS = N / 4;
    for (int i = lane; i < F * QUADS; i += 64) {
        const int f = F == 1 ? 0 : i / QUADS, qd = i - f * QUADS;
        long long frame = grp * F + f;
        if (frame >= frames) frame = frames - 1;  // idle slot: any frame that exists
        const unsigned bits = sweep_ref_quad(sp, frame, qd);
        float v[4];
        channel_quad(sp.ch, sp.ch.first_frame + (unsigned long long)frame, qd, bits, v);
        *reinterpret_cast<float4 *>(stage + f * N + qd * 4) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<unsigned *>(refstage + f * N + qd * 4) = bits;
    }
}

// Multi-wave workgroups: the channel LLRs of the F frames of group `grp` into the workgroup's own slice
// of a scratch buffer in device memory ([F][N] floats, then the frames' transmitted bits as [F][N] bytes;
// reused group after group), drawn by all threads of the workgroup, four positions per draw.  Ends with
// the fences and the barrier that make the slice readable by every wave; the caller's previous group must
// have ended with a barrier.
template <int N, int F>
constexpr size_t sweep_slice_bytes() { return (size_t)F * N * 5; }
template <int N, int F>
__device__ __forceinline__ const uint8_t *sweep_slice_ref(const float *slice) { return reinterpret_cast<const uint8_t *>(slice + F * N); }

template <int N, int F>
__device__ __forceinline__ const float *sweep_stage_group(const SweepDev &sp, float *scratch, long long grp, long long frames)
{
    static_assert(N % 4 == 0, "positions are generated four at a time");
    constexpr int QUADS = N / 4;
    float *mine = scratch + (size_t)blockIdx.x * (sweep_slice_bytes<N, F>() / 4);
    uint8_t *mine_ref = reinterpret_cast<uint8_t *>(mine + F * N);
    // four draws per round, their transmitted bits requested together: a workgroup of these kernels has few
    // waves per SIMD, and one load per draw would leave every draw waiting for its own memory round trip
    constexpr int U = 4, TOTAL = F * QUADS;
    for (int i0 = threadIdx.x; i0 < TOTAL; i0 += U * (int)blockDim.x) {
        unsigned bits[U];
        unsigned long long af[U];
        int at[U], qds[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int i = i0 + u * (int)blockDim.x;
            if (i >= TOTAL) i = TOTAL - 1;  // past the end: a draw that exists, not stored
            const int f = F == 1 ? 0 : i / QUADS, qd = i - f * QUADS;
            long long frame = grp * F + f;
            if (frame >= frames) frame = frames - 1;  // idle slot: any frame that exists
            bits[u] = sweep_ref_quad(sp, frame, qd);
            af[u] = sp.ch.first_frame + (unsigned long long)frame;
            at[u] = f * N + qd * 4;
            qds[u] = qd;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * (int)blockDim.x < TOTAL) {
                float v[4];
                channel_quad(sp.ch, af[u], qds[u], bits[u], v);
                *reinterpret_cast<float4 *>(mine + at[u]) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<unsigned *>(mine_ref + at[u]) = bits[u];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return mine;
}

// host side: channel parameters of a launch (shared by ldpc_channel_llr and the fused entry point)
inline ChannelDev make_channel_dev(double sigma2, double gain, int plus_is_bit, int prefix_len, double prefix_llr,
                                   unsigned long long seed, unsigned long long first_frame, double cfo, double phase0)
{
    ChannelDev c;
    const double pe = 3.14159265358979323846 * cfo;
    c.sigma = (float)__builtin_sqrt(sigma2);
    c.gain = (float)gain;
    c.prefix_llr = (float)prefix_llr;
    c.cfo = (float)cfo;
    c.phase0 = (float)phase0;
    c.amp = (float)(cfo == 0.0 ? 1.0 : __builtin_sin(pe) / pe);  // sinc: loss of the integrate-and-dump filter
    c.plus_is_bit = plus_is_bit;
    c.prefix_len = prefix_len;
    c.k0 = (unsigned)seed;
    c.k1 = (unsigned)(seed >> 32);
    c.first_frame = first_frame;
    return c;
}

// decode_spec.hip / decode_lnms.hip compiled with -DLDPC_SPEC_FUSED: one launch = channel + decode +
// error accounting for `io.frames` frames (io.llr and the result arrays of `io` are unused)
hipError_t launch_spec_fused(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                             const SweepDev &sp, int num_cus, Scratch &workspace, const char **kernel_name,
                             bool *supported);
// decode_spec_h.hip: rule R1 in packed half precision
hipError_t launch_spec_half_fused(int spec_id, const ldpc_decode_config &cfg, const Launch &io, const SweepDev &sp,
                                  int num_cus, const char **kernel_name, bool *supported);
// decode_spec_r1w.hip: the multi-wave flooding kernels (rules R0, R1, OMS on the larger compiled-in codes)
hipError_t launch_spec_r0w_fused(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io, const SweepDev &sp,
                                 int num_cus, Scratch &workspace, const char **kernel_name, bool *supported);
hipError_t launch_spec_r1w_fused(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io, const SweepDev &sp,
                                 int num_cus, Scratch &workspace, const char **kernel_name, bool *supported);
hipError_t launch_spec_omsw_fused(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io, const SweepDev &sp,
                                  int num_cus, Scratch &workspace, const char **kernel_name, bool *supported);
// decode_spec_spa.hip / decode_spec_spaw.hip: the sum-product rules R2, R3, R4 (single wave; several waves per frame group)
hipError_t launch_spec_spa_fused(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io, const SweepDev &sp,
                                 int num_cus, const char **kernel_name, bool *supported);
hipError_t launch_spec_spaw_fused_rule_2(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io, const SweepDev &sp,
                                         int num_cus, Scratch &workspace, const char **kernel_name, bool *supported);
hipError_t launch_spec_spaw_fused_rule_3(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io, const SweepDev &sp,
                                         int num_cus, Scratch &workspace, const char **kernel_name, bool *supported);
hipError_t launch_spec_spaw_fused_rule_4(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io, const SweepDev &sp,
                                         int num_cus, Scratch &workspace, const char **kernel_name, bool *supported);
// decode_lnms_h.hip: the layered rule in packed half precision
hipError_t launch_spec_lnms_half_fused(int spec_id, const ldpc_decode_config &cfg, const Launch &io, const SweepDev &sp,
                                       int num_cus, Scratch &workspace, const char **kernel_name, bool *supported);
hipError_t launch_spec_lnms_fused(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                                  const SweepDev &sp, int num_cus, Scratch &workspace, const char **kernel_name,
                                  bool *supported);

}  // namespace ldpc
