// Reduced-precision fast mode of rule R1 (bpsk.py:154-260): packed half precision, TWO frames per
// lane (compute_dtype LDPC_F16).  No counterpart in the reference, which computes in float64; this
// mode is NOT bit-exact with anything and is validated statistically only (frame error rates against
// the float32 kernel on the same frames, tests/test_gpu_half.py).
//
// Same organisation as spec_r1_kernel (decode_spec.hip): single-wave workgroups, thread (f, z) owns
// check (br, z) of every block row and variable (bc, z) of every block column of frame slot f,
// variable-to-check messages in LDS in variable order, compressed check state, hard decisions as
// ballots, syndrome by rotating the ballot masks.  A slot now holds a PAIR of frames: every 32-bit
// word carries the two frames' half-precision numbers side by side, and every packed instruction
// (v_pk_min_f16, v_pk_max_f16, v_pk_add_f16, v_pk_fma_f16, and plain 32-bit logic on the two sign
// bits) works on both.  Compares and selects have no packed form; the one select of the rule,
//     message = (|q| == m1) ? m2 : m1            (m1 <= m2 smallest / second smallest input magnitude),
// is done arithmetically: an input that is not the smallest is at least m2, so
//     message = min(m2, max(m1, m1 + m2 - |q|))
// (|q| = m1 gives m2; |q| >= m2 gives m1 + (m2 - |q|) <= m1, hence m1), up to one rounding of the
// half-precision sum.  With the factor 0.75 folded in (A = 0.75 m1, B = 0.75 m2):
//     message = min(B, max(A, (A + B) - 0.75 |q|)).
// Extrinsic sums are total - own message (one packed subtraction per edge) instead of the reference's
// ordered sums of the others.  Magnitudes are clamped (channel values and check outputs to 2000) so
// that no sum of up to 30 messages leaves the half-precision range.
#include "spec_common.h"
#include "half2_ops.h"
#include "sweep_fused.h"

namespace ldpc {

template <class C, int F>
struct HalfLayout {
    static constexpr int THREADS = ((F * C::Z + 63) / 64) * 64;
    static constexpr bool QREG = C::EB <= 72;  // own messages in registers
    // edges of shift 0 never leave the lane: their messages live in registers only and have no place
    // in LDS; row(e) = place of the message of edge e (variable order) in qv
    static constexpr bool LOCAL = spec_local_edges<C>() && QREG;
    static constexpr bool local(int e) { return LOCAL && C::col_shift[e] == 0; }
    static constexpr int row(int e)
    {
        int k = 0;
        for (int i = 0; i < e; ++i) k += !local(i);
        return k;
    }
    static constexpr int QV_BYTES = ((row(C::EB) * C::Z * 4 + 15) / 16) * 16;
    static constexpr int CS_BYTES = C::MB * C::Z * 16;  // A, B, A + B, sign parities
    static constexpr int SLOT_BYTES = ((QV_BYTES + CS_BYTES + 127) / 128) * 128 + (F > 1 ? 32 : 0);
    static constexpr int LDS_BYTES = F * SLOT_BYTES;
    static constexpr bool OK = THREADS == 64 && C::EB <= 100 && LDS_BYTES <= 64 * 1024;
    // the next pair's LLRs are fetched into an LDS staging area during the decode (llr_prefetch_frame)
    // where that still leaves twelve workgroups per compute unit
    static constexpr bool PREFETCH = F == 1 && (160 * 1024) / (LDS_BYTES + 2 * C::NB * 256) >= 12;
};

// FUSED: the channel in front and the error accounting behind run inside the kernel (sweep_fused.h):
// the LLRs of a slot's two frames are drawn into an LDS staging area, results go into the counters;
// llr and the result arrays are unused.
template <class C, int F, bool FUSED = false>
__global__ void __launch_bounds__(64)
spec_r1h_kernel(int max_iters, int early_exit, float scale_f, const float *__restrict__ llr, long long frames,
                uint8_t *__restrict__ hard_out, int *__restrict__ iters_out, uint8_t *__restrict__ conv_out, SweepDev sp)
{
    typedef HalfLayout<C, F> LY;
    constexpr int Z = C::Z, NB = C::NB, MB = C::MB, EB = C::EB, n = NB * Z;
    constexpr bool LOCAL = LY::LOCAL;  // edges of shift 0 never leave the lane
    extern __shared__ __align__(16) unsigned char smem[];

    const int tid = threadIdx.x;
    const bool lane_on = tid < F * Z;
    const int f = lane_on ? tid / Z : 0;
    const int z = lane_on ? tid - f * Z : 0;
    unsigned char *qv_b = smem + f * LY::SLOT_BYTES;  // [EB][Z] h2, variable order
    unsigned char *cs_b = qv_b + LY::QV_BYTES;        // [MB][Z] uint4
    const unsigned zb = (unsigned)z * 4, zc = (unsigned)z * 16;
    const unsigned long long slot_mask = (Z >= 64 ? ~0ull : ((1ull << (Z & 63)) - 1ull)) << ((f * Z) & 63);
    const h2 scale2 = pack_half2(scale_f, scale_f);

    const long long pairs = (frames + 1) / 2, groups = (pairs + F - 1) / F;
    constexpr bool PREFETCH = LY::PREFETCH && !FUSED;
    __shared__ float stage[PREFETCH ? 2 * NB * 64 : 1];  // [frame of the pair][NB][lane]
    __shared__ __align__(16) float fstage[FUSED ? 2 * F * n : 1];  // fused: channel LLRs by position, frame 2 f + h of the group
    __shared__ __align__(4) uint8_t fref[FUSED ? 2 * F * n : 4];   //        and the transmitted bits
    SweepAcc acc;
    auto prefetch = [&](long long g2) {
        if constexpr (PREFETCH) {
            const long long a = (g2 * F + f) * 2;
            llr_prefetch_frame<C>(stage, llr, a, lane_on && a < frames, z);
            llr_prefetch_frame<C>(stage + NB * 64, llr, a + 1, lane_on && a + 1 < frames, z);
        }
    };
    if (blockIdx.x < groups) prefetch(blockIdx.x);
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long fr0 = (grp * F + f) * 2, fr1 = fr0 + 1;
        const bool act0 = lane_on && fr0 < frames, act1 = lane_on && fr1 < frames;
        h2 L[NB], own[NB];
        h2 q[LY::QREG ? EB : 1];
        if constexpr (PREFETCH) __builtin_amdgcn_s_waitcnt(0);  // the staged LLRs have landed
        if constexpr (FUSED) {  // one wave: its LDS operations execute in order, no barrier needed
            sweep_stage_wave<n, 2 * F>(sp, fstage, fref, grp, frames, tid);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        float l0s[NB], l1s[NB];
        static_for<0, NB>([&](auto bc) {
            if constexpr (FUSED) {
                l0s[bc] = act0 ? fstage[(2 * f) * n + bc * Z + z] : 0.0f;
                l1s[bc] = act1 ? fstage[(2 * f + 1) * n + bc * Z + z] : 0.0f;
            } else if constexpr (PREFETCH) {
                l0s[bc] = act0 ? stage[bc * 64 + tid] : 0.0f;
                l1s[bc] = act1 ? stage[NB * 64 + bc * 64 + tid] : 0.0f;
            } else {
                l0s[bc] = act0 ? llr[fr0 * n + bc * Z + z] : 0.0f;
                l1s[bc] = act1 ? llr[fr1 * n + bc * Z + z] : 0.0f;
            }
        });
        if constexpr (PREFETCH) {
            __builtin_amdgcn_s_waitcnt(0);  // staging area read out before it is refilled
            if (grp + gridDim.x < groups) prefetch(grp + gridDim.x);
        }
        static_for<0, NB>([&](auto bc) {
            constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0;
            const float l0 = l0s[bc], l1 = l1s[bc];
            const h2 l = pack_half2(l0, l1);
            L[bc] = l;
            own[bc] = l;
            static_for<0, D>([&](auto k) {  // bpsk.py:170-177
                if constexpr (LY::QREG) q[k0 + k] = l;
                if constexpr (!(LOCAL && C::col_shift[k0 + k] == 0))
                    if (lane_on) *reinterpret_cast<h2 *>(qv_b + LY::row(k0 + k) * Z * 4 + zb) = l;  // idle lanes alias lane 0
            });
        });

        // syndrome of one frame of the pair from the signs of its posteriors
        auto frame_ok = [&](auto h) -> bool {
            unsigned long long hm[NB];
            static_for<0, NB>([&](auto bc) { hm[bc] = __ballot(!half_positive(own[bc], h)); });
            return r1_slot_ok<C, F>(hm, slot_mask);
        };
        auto write_half = [&](auto h, long long frame, int iters, int conv) {
            if constexpr (FUSED) {
                unsigned errs = 0;
                static_for<0, NB>([&](auto bc) {
                    errs += (unsigned)(half_positive(own[bc], h) ? 0 : 1) != (unsigned)fref[(2 * f + h) * n + bc * Z + z];
                });
                acc.book(errs, (__ballot(errs != 0) & slot_mask) != 0ull, z == 0, iters, conv);
            } else {
                static_for<0, NB>([&](auto bc) { hard_out[frame * n + bc * Z + z] = half_positive(own[bc], h) ? 0 : 1; });
                if (z == 0) { iters_out[frame] = iters; conv_out[frame] = (uint8_t)conv; }
            }
        };

        bool done0 = !act0, done1 = !act1;
        uint4 cl[LOCAL ? MB : 1];  // check states of this lane's own checks
        for (int t = 0; t < max_iters; ++t) {
            if (early_exit) {  // bpsk.py:205-209, per frame of the pair
                if (__ballot(!done0) != 0ull) {
                    const bool ok = frame_ok(IC<0>{});
                    if (!done0 && ok) { write_half(IC<0>{}, fr0, t, 1); done0 = true; }
                }
                if (__ballot(!done1) != 0ull) {
                    const bool ok = frame_ok(IC<1>{});
                    if (!done1 && ok) { write_half(IC<1>{}, fr1, t, 1); done1 = true; }
                }
                if (__ballot(!(done0 && done1)) == 0ull) break;
            }
            if (!(done0 && done1)) {
                // ---- check nodes (bpsk.py:213-233): smallest / second smallest magnitude and sign parity
                static_for<0, MB>([&](auto br) {
                    constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0;
                    h2 m1 = kHalfMax2, m2 = kHalfMax2, sx = 0;
                    static_for<0, D>([&](auto p) {
                        constexpr int kpos = C::row_kpos[e0 + p], s = C::row_shift[e0 + p];
                        h2 v;
                        if constexpr (LOCAL && s == 0) v = q[kpos];
                        else v = *reinterpret_cast<const h2 *>(qv_b + LY::row(kpos) * Z * 4 + rot_fwd<Z, 4, s>(zb));
                        const h2 a = v & kAbsMask;
                        m2 = pk_min(m2, pk_max(m1, a));
                        m1 = pk_min(m1, a);
                        sx ^= v;
                    });
                    uint4 st;
                    if constexpr (D < 2) st = make_uint4(0, 0, 0, 0);  // bpsk.py:216
                    else {
                        const h2 A = pk_min(pk_mul(m1, scale2), kHalfClamp2), B = pk_min(pk_mul(m2, scale2), kHalfClamp2);
                        st = make_uint4(A, B, pk_add(A, B), sx & kSignMask);
                    }
                    *reinterpret_cast<uint4 *>(cs_b + br * Z * 16 + zc) = st;
                    if constexpr (LOCAL) cl[br] = st;
                });
                // ---- variable nodes (bpsk.py:235-247)
                static_for<0, NB>([&](auto bc) {
                    constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0;
                    h2 r[D > 0 ? D : 1];
                    uint4 c[D > 0 ? D : 1];
                    h2 total = L[bc];
                    static_for<0, D>([&](auto k) {  // all check states of the variable are requested first
                        constexpr int br = C::col_row[k0 + k], s = C::col_shift[k0 + k];
                        if constexpr (LOCAL && s == 0) c[k] = cl[br];
                        else c[k] = *reinterpret_cast<const uint4 *>(cs_b + br * Z * 16 + rot_bwd<Z, 16, s>(zc));
                    });
                    static_for<0, D>([&](auto k) {
                        h2 qo;
                        if constexpr (LY::QREG) qo = q[k0 + k];
                        else qo = *reinterpret_cast<const h2 *>(qv_b + LY::row(k0 + k) * Z * 4 + zb);
                        const h2 mag = pk_min(c[k].y, pk_max(c[k].x, pk_fnma(qo & kAbsMask, scale2, c[k].z)));
                        r[k] = mag ^ ((qo ^ c[k].w) & kSignMask);  // sign = parity of the check without this input
                        total = pk_add(total, r[k]);
                    });
                    static_for<0, D>([&](auto k) {
                        const h2 qn = pk_sub(total, r[k]);
                        if constexpr (LY::QREG) q[k0 + k] = qn;
                        if constexpr (!(LOCAL && C::col_shift[k0 + k] == 0))
                            *reinterpret_cast<h2 *>(qv_b + LY::row(k0 + k) * Z * 4 + zb) = qn;
                    });
                    own[bc] = total;  // bpsk.py:237
                });
            }
        }
        if (!done0) {  // bpsk.py:251-260
            if (early_exit) write_half(IC<0>{}, fr0, max_iters - 1, 0);
            else write_half(IC<0>{}, fr0, max_iters, frame_ok(IC<0>{}) ? 1 : 0);
        }
        if (!done1) {
            if (early_exit) write_half(IC<1>{}, fr1, max_iters - 1, 0);
            else write_half(IC<1>{}, fr1, max_iters, frame_ok(IC<1>{}) ? 1 : 0);
        }
    }
    if constexpr (FUSED) acc.flush_wave(sp.counters);
}

template <class C, int F, bool FUSED = false>
static hipError_t launch_r1h(const ldpc_decode_config &cfg, const Launch &io, int num_cus, const char **name,
                             const SweepDev *sp = nullptr)
{
    typedef HalfLayout<C, F> LY;
    auto k = spec_r1h_kernel<C, F, FUSED>;
    int per_cu = 0;
    hipError_t e = kernel_setup(reinterpret_cast<const void *>(k), 64, LY::LDS_BYTES, &per_cu);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    const long long pairs = (io.frames + 1) / 2, groups = (pairs + F - 1) / F;
    long long grid = (long long)num_cus * per_cu * 2;
    if (grid > groups) grid = groups;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(64), LY::LDS_BYTES, io.stream, cfg.max_iters, cfg.early_exit,
                       (float)cfg.scale, reinterpret_cast<const float *>(io.llr), io.frames, io.hard, io.iters, io.converged,
                       sp ? *sp : SweepDev{});
    *name = FUSED ? "fused_spec_r1_f16x2" : "spec_r1_f16x2";
    return hipGetLastError();
}

// Rule R1 on the small compiled-in codes, float LLRs, no posteriors out.
hipError_t launch_spec_half(int spec_id, const ldpc_decode_config &cfg, const Launch &io, int num_cus,
                            const char **name, bool *supported)
{
    *supported = false;
    if (cfg.rule != LDPC_RULE_R1 || io.llr_dtype != LDPC_F32 || io.posterior) return hipSuccess;
    int id = 0;
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) {                                                                           \
        if constexpr (HalfLayout<CODE, SLOTS>::OK) {                                               \
            *supported = true;                                                                     \
            return launch_r1h<CODE, SLOTS>(cfg, io, num_cus, name);                                \
        }                                                                                          \
        return hipSuccess;                                                                         \
    }                                                                                              \
    ++id;
    LDPC_FOR_EACH_SPEC_CODE(LDPC_LAUNCH)
#undef LDPC_LAUNCH
    return hipSuccess;
}

// The same rule as one launch of channel + decode + error accounting (ldpc_sweep_batch), for the codes
// whose pair of staged frames fits beside the decoder's state.
hipError_t launch_spec_half_fused(int spec_id, const ldpc_decode_config &cfg, const Launch &io, const SweepDev &sp,
                                  int num_cus, const char **name, bool *supported)
{
    *supported = false;
    if (cfg.rule != LDPC_RULE_R1) return hipSuccess;
    int id = 0;
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) {                                                                           \
        if constexpr (HalfLayout<CODE, SLOTS>::OK && (CODE::NB * CODE::Z) % 4 == 0 &&              \
                      HalfLayout<CODE, SLOTS>::LDS_BYTES + 10 * SLOTS * CODE::NB * CODE::Z <= 64 * 1024) { \
            *supported = true;                                                                     \
            return launch_r1h<CODE, SLOTS, true>(cfg, io, num_cus, name, &sp);                     \
        }                                                                                          \
        return hipSuccess;                                                                         \
    }                                                                                              \
    ++id;
    LDPC_FOR_EACH_SPEC_CODE(LDPC_LAUNCH)
#undef LDPC_LAUNCH
    return hipSuccess;
}

}  // namespace ldpc
