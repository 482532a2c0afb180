// Code-specialised sum-product decoders (rules R2, R3, R4) for the small packaged codes.
//
// Same construction as the specialised min-sum kernels (decode_spec.hip): the base graph is a
// template parameter, one wave is one workgroup (F frames side by side when Z < 64), no barriers.
//   msg [EB][Z]  one message per edge, in check order (by-row edge index, check's z), updated in
//                place: check (br, z) reads its variable-to-check messages at [e][z] and overwrites
//                them with its check-to-variable messages; variable (bc, z') finds edge k of its
//                column at [col_slot[k]][(z' - shift) mod Z] and writes the next variable-to-check
//                message there.  LDS operations of one wave execute in order, which is all the
//                synchronisation this needs.
// Channel LLRs and posteriors live in registers; hard decisions are ballots and the syndrome is
// scalar code (see r1_slot_ok).
//
// R2  bpsk.py:263-387     tanh rule, inner clip 20, product clip 0.999999999, posterior clip 30
// R3  channel.py:402-513  tanh rule, 0.85 * 2 * atanh, product clip 0.99999999, posterior clip 50
// R4  channel.py:260-398  phi rule, sum clip [1e-12, 20], posterior clip 500
// The clips arrive in the configuration; prefix products and suffix products replace the
// reference's per-target loops (agreement to rounding, see the table-driven kernel).  The tanh rules
// (R2, R3) evaluate a whole check at once through spa_check_tanh (spec_common.h): numerators and
// denominators of the tanh values are multiplied separately and one logarithm of a quotient replaces
// tanh followed by atanh.
//
// FUSED: the channel in front and the error accounting behind run inside the kernel (sweep_fused.h),
// as in the single-wave min-sum kernels of decode_spec.hip: the lanes draw the LLRs of the group's
// frames into an LDS staging area, the frames' transmitted bits beside them, and a frame that ends
// books its bit errors and flags instead of writing results.  llr and the result arrays are unused.
#include "spec_common.h"
#include "sweep_fused.h"

namespace ldpc {

template <class C, typename T, int F>
struct SpaLayout {
    static constexpr int THREADS = ((F * C::Z + 63) / 64) * 64;
    // edges of shift 0 connect a check and a variable of the same lane: their message lives in a
    // register of that lane (ml[]) and has no place in LDS; row(e) = place of edge e (check order) in msg
    static constexpr bool LOCAL = spec_local_edges<C>();
    static constexpr bool local(int e) { return LOCAL && C::row_shift[e] == 0; }
    static constexpr int row(int e)
    {
        int k = 0;
        for (int i = 0; i < e; ++i) k += !local(i);
        return k;
    }
    static constexpr int reg(int e)  // index of a local edge among the local edges
    {
        int k = 0;
        for (int i = 0; i < e; ++i) k += local(i);
        return k;
    }
    static constexpr int NLOCAL = reg(C::EB);
    static constexpr int MSG_BYTES = row(C::EB) * C::Z * (int)sizeof(T);
    static constexpr int SLOT_BYTES = ((MSG_BYTES + 127) / 128) * 128 + (F > 1 ? 32 : 0);
    static constexpr int LDS_BYTES = F * SLOT_BYTES;
};

template <int RULE, class C, typename T, typename TL, int F, bool FUSED = false>
__global__ void __launch_bounds__(64)
spec_spa_kernel(Cfg<T> g, const TL *__restrict__ llr, long long frames, uint8_t *__restrict__ hard_out,
                int *__restrict__ iters_out, uint8_t *__restrict__ conv_out, T *__restrict__ post_out, SweepDev sp)
{
    typedef SpaLayout<C, T, F> LY;
    static_assert(LY::THREADS == 64, "single-wave workgroups only");
    constexpr int Z = C::Z, NB = C::NB, MB = C::MB, ST = sizeof(T);
    constexpr int n = NB * Z;
    extern __shared__ __align__(16) unsigned char smem[];

    const int tid = threadIdx.x;
    const bool lane_on = tid < F * Z;
    const int f = lane_on ? tid / Z : 0;
    const int z = lane_on ? tid - f * Z : 0;
    unsigned char *msg_b = smem + f * LY::SLOT_BYTES;  // [EB][Z] T
    const unsigned zb = (unsigned)z * ST;
    const unsigned long long slot_mask = (Z >= 64 ? ~0ull : ((1ull << (Z & 63)) - 1ull)) << ((f * Z) & 63);

    constexpr bool PREFETCH = !FUSED && llr_prefetch_pays<C, TL>(LY::LDS_BYTES);  // see llr_prefetch
    __shared__ float stage[PREFETCH ? NB * 64 : 1];  // [NB][lane]
    __shared__ __align__(16) float fstage[FUSED ? F * n : 1];  // fused: channel LLRs by position
    __shared__ __align__(4) uint8_t fref[FUSED ? F * n : 4];   //        and the transmitted bits
    SweepAcc acc;
    const long long groups = (frames + F - 1) / F;
    if constexpr (PREFETCH) {
        if (blockIdx.x < groups) llr_prefetch<C, F>(stage, llr, blockIdx.x, frames, lane_on, f, z);
    }
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long frame = grp * F + f;
        const bool active = lane_on && frame < frames;
        T L[NB], own[NB];
        T ml[LY::NLOCAL > 0 ? LY::NLOCAL : 1];  // messages of the lane-local edges
        unsigned long long hm[NB];
        if constexpr (PREFETCH) __builtin_amdgcn_s_waitcnt(0);  // the staged LLRs have landed
        if constexpr (FUSED) {  // one wave: its LDS operations execute in order, no barrier needed
            sweep_stage_wave<n, F>(sp, fstage, fref, grp, frames, tid);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        static_for<0, NB>([&](auto bc) {
            constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0;
            T l;
            if constexpr (FUSED) l = active ? (T)fstage[f * n + bc * Z + z] : (T)0;
            else if constexpr (PREFETCH) l = active ? (T)stage[bc * 64 + tid] : (T)0;
            else l = active ? (T)llr[frame * n + bc * Z + z] : (T)0;
            L[bc] = l;
            own[bc] = l;
            hm[bc] = __ballot(!(l > (T)0));
            static_for<0, D>([&](auto k) {  // bpsk.py:299-306: every edge starts with the channel LLR
                constexpr int e = C::col_slot[k0 + k], s = C::col_shift[k0 + k];
                if constexpr (LY::local(e)) ml[LY::reg(e)] = l;
                else if (lane_on) *reinterpret_cast<T *>(msg_b + LY::row(e) * Z * ST + rot_bwd<Z, ST, s>(zb)) = l;  // idle lanes alias lane 0
            });
        });
        if constexpr (PREFETCH) {
            __builtin_amdgcn_s_waitcnt(0);  // staging area read out before it is refilled
            if (grp + gridDim.x < groups) llr_prefetch<C, F>(stage, llr, grp + gridDim.x, frames, lane_on, f, z);
        }

        // results of a frame: to the arrays, or (fused) into the error counters
        auto write_out = [&](int iters, int conv) {
            unsigned errs = 0;
            static_for<0, NB>([&](auto bc) {
                const uint8_t hv = (uint8_t)hard_bit<RULE>(own[bc]);
                if constexpr (FUSED) errs += hv != fref[f * n + bc * Z + z];
                else {
                    hard_out[frame * n + bc * Z + z] = hv;
                    if (post_out) post_out[frame * n + bc * Z + z] = own[bc];
                }
            });
            if constexpr (FUSED) acc.book(errs, (__ballot(errs != 0) & slot_mask) != 0ull, z == 0, iters, conv);
            else if (z == 0) { iters_out[frame] = iters; conv_out[frame] = (uint8_t)conv; }
        };

        bool done = !active;
        for (int t = 0; t < g.max_iters; ++t) {
            if (g.early_exit) {  // bpsk.py:311-316
                const bool ok = r1_slot_ok<C, F>(hm, slot_mask);
                if (!done && ok) { write_out(t, 1); done = true; }
                if (__ballot(!done) == 0ull) break;
            }
            if (!done) {
                // ---- check nodes
                static_for<0, MB>([&](auto br) {
                    constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0;
                    T v[D > 0 ? D : 1];
                    static_for<0, D>([&](auto p) {
                        if constexpr (LY::local(e0 + p)) v[p] = ml[LY::reg(e0 + p)];
                        else v[p] = *reinterpret_cast<const T *>(msg_b + LY::row(e0 + p) * Z * ST + zb);
                    });
                    auto put = [&](auto p, T m) {
                        if constexpr (LY::local(e0 + p)) ml[LY::reg(e0 + p)] = m;
                        else *reinterpret_cast<T *>(msg_b + LY::row(e0 + p) * Z * ST + zb) = m;
                    };
                    if constexpr (RULE == 2 || RULE == 3) {  // bpsk.py:325-339, channel.py:471-474
                        T o[D > 0 ? D : 1];
                        spa_check_tanh<T, D>(v, o, g.clip_tanh_arg, g.clip_prod, g.scale, RULE == 3);
                        static_for<0, D>([&](auto p) { put(p, o[p]); });
                    } else {  // channel.py:332-352
                        T ph[D > 0 ? D : 1], suf[D > 0 ? D : 1];
                        unsigned neg = 0;
                        static_for<0, D>([&](auto p) {
                            neg |= (unsigned)(v[p] < (T)0) << p;  // 0 and 1e-9 count as +
                            ph[p] = phi(t_abs(v[p]));
                        });
                        T run = (T)0;
                        static_for<0, D>([&](auto pp) { constexpr int p = D - 1 - pp; suf[p] = run; run += ph[p]; });
                        T pre = (T)0;
                        static_for<0, D>([&](auto p) {
                            const bool negative = (__popc(neg & ~(1u << p)) & 1) != 0;
                            put(p, phi_message(pre + suf[p], negative));
                            pre += ph[p];
                        });
                    }
                });
                // ---- variable nodes (bpsk.py:346-375, channel.py:358-367, 480-491)
                static_for<0, NB>([&](auto bc) {
                    constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0;
                    T r[D > 0 ? D : 1], qn[D > 0 ? D : 1];
                    unsigned off[D > 0 ? D : 1];
                    static_for<0, D>([&](auto k) {
                        constexpr int e = C::col_slot[k0 + k], s = C::col_shift[k0 + k];
                        off[k] = LY::row(e) * Z * ST + rot_bwd<Z, ST, s>(zb);
                        if constexpr (LY::local(e)) r[k] = ml[LY::reg(e)];
                        else r[k] = *reinterpret_cast<const T *>(msg_b + off[k]);
                    });
                    bool sw[D > 0 ? D : 1];
                    static_for<0, (D > 0 ? D - 1 : 0)>([&](auto k) {
                        if constexpr (C::col_pair[k0 + k] != 0) {  // two circulants in one block: lower check index first
                            constexpr int s0 = C::col_shift[k0 + k], s1 = C::col_shift[k0 + k + 1];
                            sw[k] = rot_bwd<Z, ST, s0>(zb) > rot_bwd<Z, ST, s1>(zb);
                            const T a = r[k], b = r[k + 1];
                            r[k] = sw[k] ? b : a;
                            r[k + 1] = sw[k] ? a : b;
                        }
                    });
                    T run = (T)0;
                    static_for<0, D>([&](auto k) {
                        T ext = run;
                        static_for<k + 1, D>([&](auto kk) { ext += r[kk]; });
                        run += r[k];
                        qn[k] = L[bc] + ext;
                    });
                    static_for<0, (D > 0 ? D - 1 : 0)>([&](auto k) {
                        if constexpr (C::col_pair[k0 + k] != 0) {
                            const T a = qn[k], b = qn[k + 1];
                            qn[k] = sw[k] ? b : a;
                            qn[k + 1] = sw[k] ? a : b;
                        }
                    });
                    static_for<0, D>([&](auto k) {
                        constexpr int e = C::col_slot[k0 + k];
                        if constexpr (LY::local(e)) ml[LY::reg(e)] = qn[k];
                        else *reinterpret_cast<T *>(msg_b + off[k]) = qn[k];
                    });
                    T b = L[bc] + run;
                    if (g.clip_post > (T)0) b = clip(b, -g.clip_post, g.clip_post);
                    own[bc] = b;
                    hm[bc] = __ballot(!(b > (T)0));
                });
            }
        }
        if (!done) {
            if (g.early_exit) write_out(g.max_iters - 1, 0);
            else write_out(g.max_iters, r1_slot_ok<C, F>(hm, slot_mask) ? 1 : 0);
        }
    }
    if constexpr (FUSED) acc.flush_wave(sp.counters);
}

// ------------------------------------------------------------------------------------------ host
// One wave per workgroup: with fewer than eight resident workgroups per compute unit (LDS or
// registers) the long dependent chains of the transcendentals are not hidden and the table-driven
// kernel, which spreads a frame over several waves, is faster; *supported = false sends the call there.
// This file is compiled twice, like decode_spec.hip: as it is (the decoders behind ldpc_decode_batch) and
// with -DLDPC_SPEC_FUSED (the fused variants behind ldpc_sweep_batch).
#ifdef LDPC_SPEC_FUSED
constexpr bool kFused = true;
#else
constexpr bool kFused = false;
#endif

template <int RULE, class C, typename T, typename TL, int F>
static hipError_t launch_spa(const ldpc_decode_config &cfg, const Launch &io, int num_cus, const char **name,
                             bool *supported, const SweepDev *sp)
{
    typedef SpaLayout<C, T, F> LY;
    Cfg<T> g;
    g.rule = RULE; g.max_iters = cfg.max_iters; g.early_exit = cfg.early_exit;
    g.float_syndrome0 = 0; g.num_active_checks = -1; g.num_refresh_vars = -1;
    g.scale = (T)cfg.scale; g.clip_tanh_arg = (T)cfg.clip_tanh_arg; g.clip_prod = (T)cfg.clip_prod;
    g.clip_post = (T)cfg.clip_post; g.offset = (T)cfg.offset;
    auto k = spec_spa_kernel<RULE, C, T, TL, F, kFused>;
    int per_cu = 0;
    hipError_t e = kernel_setup(reinterpret_cast<const void *>(k), 64, LY::LDS_BYTES, &per_cu);
    if (e != hipSuccess) return e;
    // the plain kernel decides which codes this file serves, so that a code's fused sweep runs the kernel
    // family its three launches run (the fused variant holds F * n * 5 bytes of staging besides)
    if (kFused ? (160 * 1024) / LY::LDS_BYTES < 8 : per_cu < 8) { *supported = false; return hipSuccess; }
    if (per_cu < 1) per_cu = 1;
    const long long groups = (io.frames + F - 1) / F;
    long long grid = (long long)num_cus * per_cu * 2;
    if (grid > groups) grid = groups;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(64), LY::LDS_BYTES, io.stream, g,
                       reinterpret_cast<const TL *>(io.llr), io.frames, io.hard, io.iters, io.converged,
                       reinterpret_cast<T *>(io.posterior), sp ? *sp : SweepDev{});
    *name = kFused ? (sizeof(T) == 8 ? "fused_spec_spa_f64" : "fused_spec_spa_f32") : (sizeof(T) == 8 ? "spec_spa_f64" : "spec_spa_f32");
    return hipGetLastError();
}

// sp != NULL only in the fused build, where the LLRs are floats drawn in the kernel
template <int RULE, class C, int F>
static hipError_t launch_spa_rule(const ldpc_decode_config &cfg, int compute_dtype, const Launch &io, int num_cus,
                                  const char **name, bool *supported, const SweepDev *sp)
{
    *supported = true;
    if (compute_dtype == LDPC_F64) {
        if constexpr (!kFused) {
            if (io.llr_dtype == LDPC_F64) return launch_spa<RULE, C, double, double, F>(cfg, io, num_cus, name, supported, sp);
        }
        return launch_spa<RULE, C, double, float, F>(cfg, io, num_cus, name, supported, sp);
    }
    if (kFused || io.llr_dtype == LDPC_F32) return launch_spa<RULE, C, float, float, F>(cfg, io, num_cus, name, supported, sp);
    *supported = false;
    return hipSuccess;
}

template <class C, int F>
static hipError_t launch_spa_code(const ldpc_decode_config &cfg, int compute_dtype, const Launch &io, int num_cus,
                                  const char **name, bool *supported, const SweepDev *sp)
{
    *supported = false;
    // small base graphs only: the whole graph is unrolled
    if constexpr (C::EB <= 100 && SpaLayout<C, double, F>::THREADS == 64 && SpaLayout<C, double, F>::LDS_BYTES <= 160 * 1024 &&
                  (!kFused || (C::NB * C::Z) % 4 == 0)) {
        if (cfg.rule == 2) return launch_spa_rule<2, C, F>(cfg, compute_dtype, io, num_cus, name, supported, sp);
        if (cfg.rule == 3) return launch_spa_rule<3, C, F>(cfg, compute_dtype, io, num_cus, name, supported, sp);
        if (cfg.rule == 4) return launch_spa_rule<4, C, F>(cfg, compute_dtype, io, num_cus, name, supported, sp);
    }
    return hipSuccess;
}

#ifdef LDPC_SPEC_FUSED
// Fused sweep on the single-wave sum-product kernels (rules R2, R3, R4 on the small compiled-in codes)
hipError_t launch_spec_spa_fused(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                                 const SweepDev &sp, int num_cus, const char **name, bool *supported)
{
    int id = 0;
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) return launch_spa_code<CODE, SLOTS>(cfg, compute_dtype, io, num_cus, name, supported, &sp); \
    ++id;
    LDPC_FOR_EACH_SPEC_CODE(LDPC_LAUNCH)
#undef LDPC_LAUNCH
    *supported = false;
    return hipSuccess;
}
#else
hipError_t launch_spec_spa(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                           int num_cus, const char **name, bool *supported)
{
    int id = 0;
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) return launch_spa_code<CODE, SLOTS>(cfg, compute_dtype, io, num_cus, name, supported, nullptr); \
    ++id;
    LDPC_FOR_EACH_SPEC_CODE(LDPC_LAUNCH)
#undef LDPC_LAUNCH
    *supported = false;
    return hipSuccess;
}
#endif

}  // namespace ldpc
