// Sum-product rules R2, R3, R4 on the compile-time path with several waves per frame group, for
// the packaged codes the single-wave kernel of decode_spec_spa.hip does not serve: base graphs of
// more than 100 edges (the BG2-shaped 5G matrix) and codes whose messages leave room for fewer
// than eight single-wave workgroups per compute unit (802.11n in double).
//
// Division of work as in decode_spec_r1w.hip: W waves share the F frame slots, every wave owns a
// balanced share of the block rows (check phase) and of the block columns (variable phase).
// LDS per slot:  msg [EB][Z]  one message per edge in check order, updated in place by both phases
//                             (see decode_spec_spa.hip)
//                hb  [NB][Z]  hard decisions `not (b > 0)`, one byte each
// One iteration: parity of the hard decisions over the wave's rows (bpsk.py:311-316) | barrier |
// slots with a zero syndrome stop | check phase | barrier | variable phase | barrier.
//
// R2  bpsk.py:263-387     R3  channel.py:402-513     R4  channel.py:260-398
//
// Compiled a second time with -DLDPC_SPEC_FUSED for ldpc_sweep_batch: spec_spaw_kernel with the channel
// in front and the error accounting behind (FUSED, as in spec_r1w_kernel of decode_spec_r1w.hip).
#include "spec_r1w_plan.h"
#include "sweep_fused.h"

namespace ldpc {

template <class C, typename T, int F>
struct SpawLayout {
    static constexpr int LANES = F * C::Z;
    static constexpr int MSG_BYTES = C::EB * C::Z * (int)sizeof(T);
    static constexpr int HB_BYTES = ((C::NB * C::Z + 15) / 16) * 16;
    static constexpr int SLOT_BYTES = ((MSG_BYTES + HB_BYTES + 127) / 128) * 128 + (F > 1 ? 32 : 0);
    static constexpr int FLAG_OFF = F * SLOT_BYTES;
    static constexpr int LDS_BYTES = FLAG_OFF + 96;  // flags: two rows of eight (syndromes), one more for the fused sweep
};

// REF: variables from g.num_refresh_vars on keep sending their channel value (the reference's refresh
// loops run over range(self.n), the constructor's n: bpsk.py:356, channel.py:358,480).
// FUSED: the workgroup draws the LLRs of its F frames into its own slice of `scratch` ([F][n] floats +
// [F][n] reference bytes in device memory, reused group after group) and decodes from there; every lane
// counts the bit errors of the variables it owns against the transmitted word, the per-frame flags meet in
// LDS.  llr and the result arrays are unused.
template <int RULE, class C, typename T, typename TL, int F, int W, bool REF = false, bool FUSED = false>
__global__ void __launch_bounds__(64 * W * ((F * C::Z + 63) / 64))
spec_spaw_kernel(Cfg<T> g, const TL *__restrict__ llr, long long frames, uint8_t *__restrict__ hard_out,
                 int *__restrict__ iters_out, uint8_t *__restrict__ conv_out, T *__restrict__ post_out, SweepDev sp,
                 float *scratch)
{
    typedef SpawLayout<C, T, F> LY;
    typedef R1wPlan<C, W> PL;
    constexpr int G = (LY::LANES + 63) / 64;  // lane groups: G waves side by side cover the F slots
    constexpr int Z = C::Z, NB = C::NB, ST = sizeof(T);
    constexpr int n = NB * Z, NBW = PL::NBW, MBW = PL::MBW;
    extern __shared__ __align__(16) unsigned char smem[];
    static_assert(F <= 8, "the per-slot syndrome flags are two rows of eight");

    const int wave = ((int)threadIdx.x >> 6) / G, lane = (((int)threadIdx.x >> 6) % G) * 64 + ((int)threadIdx.x & 63);
    const bool lane_on = lane < F * Z;
    const int f = lane_on ? lane / Z : 0;
    const int z = lane_on ? lane - f * Z : 0;
    unsigned char *msg_b = smem + f * LY::SLOT_BYTES;  // [EB][Z] T
    unsigned char *hb_b = msg_b + LY::MSG_BYTES;       // [NB][Z] u8
    int *flags = reinterpret_cast<int *>(smem + LY::FLAG_OFF);  // [2][8]: failed-syndrome flag per slot, by iteration parity
    const unsigned zb = (unsigned)z * ST, z1 = (unsigned)z;
    SweepAcc acc;

    const long long groups = (frames + F - 1) / F;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long frame = grp * F + f;
        const bool active = lane_on && frame < frames;
        const float *mine = nullptr;
        if constexpr (FUSED) {
            mine = sweep_stage_group<n, F>(sp, scratch, grp, frames);
        }
        T L[NBW], own[NBW];
        static_for<0, W>([&](auto w) {
            if (wave == w) {
                static_for<0, NBW>([&](auto j) {
                    if constexpr (j < PL::COLS.cnt[w]) {
                        constexpr int bc = PL::COLS.item[w][j];
                        constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0;
                        T l;
                        if constexpr (FUSED) l = active ? (T)mine[f * n + bc * Z + z] : (T)0;
                        else l = active ? (T)llr[frame * n + bc * Z + z] : (T)0;
                        L[j] = l;
                        own[j] = l;
                        if (lane_on) hb_b[bc * Z + z1] = (uint8_t)!(l > (T)0);  // idle lanes alias lane 0
                        static_for<0, D>([&](auto k) {  // bpsk.py:299-306: every edge starts with the channel LLR
                            constexpr int e = C::col_slot[k0 + k], s = C::col_shift[k0 + k];
                            if (lane_on) *reinterpret_cast<T *>(msg_b + e * Z * ST + rot_bwd<Z, ST, s>(zb)) = l;
                        });
                    }
                });
            }
        });
        if (threadIdx.x < 24) flags[threadIdx.x] = 0;
        __syncthreads();

        bool done = !active;
        int res_iters = g.early_exit ? g.max_iters - 1 : g.max_iters, res_conv = 0;
        for (int t = 0;; ++t) {
            const bool last = t == g.max_iters;
            if (last && g.early_exit) break;  // no test after the last update
            if (g.early_exit || last) {
                // ---- syndrome of the current hard decisions, this wave's block rows
                unsigned bad = 0;
                if (!done) {
                    static_for<0, W>([&](auto w) {
                        if (wave == w) {
                            static_for<0, MBW>([&](auto i) {
                                if constexpr (i < PL::ROWS.cnt[w]) {
                                    constexpr int br = PL::ROWS.item[w][i];
                                    constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0;
                                    unsigned hv[D > 0 ? D : 1];
                                    static_for<0, D>([&](auto p) {
                                        constexpr int bc = C::row_col[e0 + p], s = C::row_shift[e0 + p];
                                        hv[p] = hb_b[bc * Z + rot_fwd<Z, 1, s>(z1)];
                                    });
                                    unsigned par = 0;
                                    static_for<0, D>([&](auto p) { par ^= hv[p]; });
                                    bad |= par;
                                }
                            });
                        }
                    });
                }
                int *slot_bad = flags + (t & 1) * 8;
                if (bad) slot_bad[f] = 1;
                __syncthreads();
                // the other buffer was last read one iteration ago; its next writes come after the
                // barriers of this iteration
                if (threadIdx.x < 8) flags[((t + 1) & 1) * 8 + threadIdx.x] = 0;
                const bool slot_ok = slot_bad[f] == 0;
                if (last) {
                    if (!done) res_conv = slot_ok ? 1 : 0;
                    break;
                }
                if (!done && slot_ok) { res_iters = t; res_conv = 1; done = true; }
                bool any_live = false;
                static_for<0, F>([&](auto ff) { any_live |= slot_bad[ff] != 0; });
                if (!any_live) break;
            }
            // ---- check nodes of this wave's block rows
            if (!done) {
                static_for<0, W>([&](auto w) {
                    if (wave == w) {
                        static_for<0, MBW>([&](auto i) {
                            if constexpr (i < PL::ROWS.cnt[w]) {
                                constexpr int br = PL::ROWS.item[w][i];
                                constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0;
                                T v[D > 0 ? D : 1];
                                static_for<0, D>([&](auto p) { v[p] = *reinterpret_cast<const T *>(msg_b + (e0 + p) * Z * ST + zb); });
                                if constexpr (RULE == 2 || RULE == 3) {  // bpsk.py:325-339, channel.py:471-474
                                    T o[D > 0 ? D : 1];
                                    spa_check_tanh<T, D>(v, o, g.clip_tanh_arg, g.clip_prod, g.scale, RULE == 3);
                                    static_for<0, D>([&](auto p) { *reinterpret_cast<T *>(msg_b + (e0 + p) * Z * ST + zb) = o[p]; });
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
                                        *reinterpret_cast<T *>(msg_b + (e0 + p) * Z * ST + zb) = phi_message(pre + suf[p], negative);
                                        pre += ph[p];
                                    });
                                }
                            }
                        });
                    }
                });
            }
            __syncthreads();
            // ---- variable nodes of this wave's block columns (bpsk.py:346-375, channel.py:358-367, 480-491)
            if (!done) {
                static_for<0, W>([&](auto w) {
                    if (wave == w) {
                        static_for<0, NBW>([&](auto j) {
                            if constexpr (j < PL::COLS.cnt[w]) {
                                constexpr int bc = PL::COLS.item[w][j];
                                constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0;
                                T r[D > 0 ? D : 1], qn[D > 0 ? D : 1];
                                unsigned off[D > 0 ? D : 1];
                                static_for<0, D>([&](auto k) {
                                    constexpr int e = C::col_slot[k0 + k], s = C::col_shift[k0 + k];
                                    off[k] = e * Z * ST + rot_bwd<Z, ST, s>(zb);
                                    r[k] = *reinterpret_cast<const T *>(msg_b + off[k]);
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
                                    qn[k] = L[j] + ext;
                                });
                                static_for<0, (D > 0 ? D - 1 : 0)>([&](auto k) {
                                    if constexpr (C::col_pair[k0 + k] != 0) {
                                        const T a = qn[k], b = qn[k + 1];
                                        qn[k] = sw[k] ? b : a;
                                        qn[k + 1] = sw[k] ? a : b;
                                    }
                                });
                                bool refresh = true;
                                if constexpr (REF) refresh = bc * Z + z < g.num_refresh_vars;
                                static_for<0, D>([&](auto k) { *reinterpret_cast<T *>(msg_b + off[k]) = refresh ? qn[k] : L[j]; });
                                T b = L[j] + run;
                                if (g.clip_post > (T)0) b = clip(b, -g.clip_post, g.clip_post);
                                own[j] = b;
                                hb_b[bc * Z + z1] = (uint8_t)!(b > (T)0);
                            }
                        });
                    }
                });
            }
            __syncthreads();
        }
        // ---- results: every wave writes its own columns (fused: counts their bit errors)
        if constexpr (FUSED) {
            unsigned errs = 0;
            if (active) {
                const uint8_t *rf = sweep_slice_ref<n, F>(mine) + f * n;  // the frame's transmitted bits, staged beside its LLRs
                static_for<0, W>([&](auto w) {
                    if (wave == w) {
                        static_for<0, NBW>([&](auto j) {
                            if constexpr (j < PL::COLS.cnt[w]) {
                                constexpr int bc = PL::COLS.item[w][j];
                                errs += hard_bit<RULE>(own[j]) != (unsigned)rf[bc * Z + z];
                            }
                        });
                    }
                });
            }
            if (errs) flags[16 + f] = 1;
            __syncthreads();
            if (active) acc.book(errs, flags[16 + f] != 0, wave == 0 && z == 0, res_iters, res_conv);
        } else if (active) {
            static_for<0, W>([&](auto w) {
                if (wave == w) {
                    static_for<0, NBW>([&](auto j) {
                        if constexpr (j < PL::COLS.cnt[w]) {
                            constexpr int bc = PL::COLS.item[w][j];
                            hard_out[frame * n + bc * Z + z] = (uint8_t)hard_bit<RULE>(own[j]);
                            if (post_out) post_out[frame * n + bc * Z + z] = own[j];
                        }
                    });
                }
            });
            if (wave == 0 && z == 0) { iters_out[frame] = res_iters; conv_out[frame] = (uint8_t)res_conv; }
        }
        __syncthreads();
    }
    if constexpr (FUSED) acc.flush_wave(sp.counters);
}

// ------------------------------------------------------------------------------ independent slots
// Early exit with several slots per workgroup: every slot runs its own queue of frames, as in
// spec_r1w_slots_kernel (decode_spec_r1w.hip) - slot s of workgroup b decodes frames
// (b + k * gridDim.x) * F + s and starts the next one when the current one ends; the slots share
// the three barriers of an iteration.  Same arithmetic as spec_spaw_kernel; early exit only,
// max_iters >= 1.
template <int RULE, class C, typename T, typename TL, int F, int W, bool REF = false>
__global__ void __launch_bounds__(64 * W * ((F * C::Z + 63) / 64))
spec_spaw_slots_kernel(Cfg<T> g, const TL *__restrict__ llr, long long frames, uint8_t *__restrict__ hard_out,
                       int *__restrict__ iters_out, uint8_t *__restrict__ conv_out, T *__restrict__ post_out, SweepDev,
                       float *)  // the last two: spec_spaw_kernel's signature, unused here
{
    typedef SpawLayout<C, T, F> LY;
    typedef R1wPlan<C, W> PL;
    constexpr int G = (LY::LANES + 63) / 64;
    constexpr int Z = C::Z, NB = C::NB, ST = sizeof(T);
    constexpr int n = NB * Z, NBW = PL::NBW, MBW = PL::MBW;
    extern __shared__ __align__(16) unsigned char smem[];
    static_assert(F <= 8, "the per-slot syndrome flags are two rows of eight");

    const int wave = ((int)threadIdx.x >> 6) / G, lane = (((int)threadIdx.x >> 6) % G) * 64 + ((int)threadIdx.x & 63);
    const bool lane_on = lane < F * Z;
    const int f = lane_on ? lane / Z : 0;
    const int z = lane_on ? lane - f * Z : 0;
    unsigned char *msg_b = smem + f * LY::SLOT_BYTES;
    unsigned char *hb_b = msg_b + LY::MSG_BYTES;
    int *flags = reinterpret_cast<int *>(smem + LY::FLAG_OFF);
    const unsigned zb = (unsigned)z * ST, z1 = (unsigned)z;

    const long long stride = (long long)gridDim.x * F;
    long long frame_s[F];  // uniform view of every slot: current frame and iteration
    int t_s[F];
    static_for<0, F>([&](auto ff) { frame_s[ff] = (long long)blockIdx.x * F + ff; t_s[ff] = 0; });
    long long frame = (long long)blockIdx.x * F + f;  // this lane's slot
    int t = 0;
    bool live = lane_on && frame < frames;

    T L[NBW], own[NBW];
    auto load_frame = [&]() {
        static_for<0, W>([&](auto w) {
            if (wave == w) {
                static_for<0, NBW>([&](auto j) {
                    if constexpr (j < PL::COLS.cnt[w]) {
                        constexpr int bc = PL::COLS.item[w][j];
                        constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0;
                        const T l = (T)llr[frame * n + bc * Z + z];
                        L[j] = l;
                        own[j] = l;
                        hb_b[bc * Z + z1] = (uint8_t)!(l > (T)0);
                        static_for<0, D>([&](auto k) {  // bpsk.py:299-306: every edge starts with the channel LLR
                            constexpr int e = C::col_slot[k0 + k], s = C::col_shift[k0 + k];
                            *reinterpret_cast<T *>(msg_b + e * Z * ST + rot_bwd<Z, ST, s>(zb)) = l;
                        });
                    }
                });
            }
        });
    };
    static_for<0, NBW>([&](auto j) { L[j] = (T)0; own[j] = (T)0; });
    if (live) load_frame();
    if (threadIdx.x < 16) flags[threadIdx.x] = 0;
    __syncthreads();

    for (int round = 0;; ++round) {
        bool any_live = false;
        static_for<0, F>([&](auto ff) { any_live |= frame_s[ff] < frames; });
        if (!any_live) break;  // uniform: every queue is empty
        // ---- syndrome of the current hard decisions, this wave's block rows
        unsigned bad = 0;
        if (live) {
            static_for<0, W>([&](auto w) {
                if (wave == w) {
                    static_for<0, MBW>([&](auto i) {
                        if constexpr (i < PL::ROWS.cnt[w]) {
                            constexpr int br = PL::ROWS.item[w][i];
                            constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0;
                            unsigned hv[D > 0 ? D : 1];
                            static_for<0, D>([&](auto p) {
                                constexpr int bc = C::row_col[e0 + p], s = C::row_shift[e0 + p];
                                hv[p] = hb_b[bc * Z + rot_fwd<Z, 1, s>(z1)];
                            });
                            unsigned par = 0;
                            static_for<0, D>([&](auto p) { par ^= hv[p]; });
                            bad |= par;
                        }
                    });
                }
            });
        }
        int *slot_bad = flags + (round & 1) * 8;
        if (bad) slot_bad[f] = 1;
        __syncthreads();
        if (threadIdx.x < 8) flags[((round + 1) & 1) * 8 + threadIdx.x] = 0;
        bool ok_s[F];  // outcome per slot, the same in every wave
        static_for<0, F>([&](auto ff) { ok_s[ff] = __builtin_amdgcn_readfirstlane(slot_bad[ff]) == 0; });
        bool slot_ok = ok_s[0];
        static_for<1, F>([&](auto ff) { if (f == ff) slot_ok = ok_s[ff]; });
        const bool converged = live && slot_ok;
        const bool update = live && !slot_ok;
        // ---- check nodes of this wave's block rows
        if (update) {
            static_for<0, W>([&](auto w) {
                if (wave == w) {
                    static_for<0, MBW>([&](auto i) {
                        if constexpr (i < PL::ROWS.cnt[w]) {
                            constexpr int br = PL::ROWS.item[w][i];
                            constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0;
                            T v[D > 0 ? D : 1];
                            static_for<0, D>([&](auto p) { v[p] = *reinterpret_cast<const T *>(msg_b + (e0 + p) * Z * ST + zb); });
                            if constexpr (RULE == 2 || RULE == 3) {  // bpsk.py:325-339, channel.py:471-474
                                T o[D > 0 ? D : 1];
                                spa_check_tanh<T, D>(v, o, g.clip_tanh_arg, g.clip_prod, g.scale, RULE == 3);
                                static_for<0, D>([&](auto p) { *reinterpret_cast<T *>(msg_b + (e0 + p) * Z * ST + zb) = o[p]; });
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
                                    *reinterpret_cast<T *>(msg_b + (e0 + p) * Z * ST + zb) = phi_message(pre + suf[p], negative);
                                    pre += ph[p];
                                });
                            }
                        }
                    });
                }
            });
        }
        __syncthreads();
        // ---- variable nodes of this wave's block columns (bpsk.py:346-375, channel.py:358-367, 480-491)
        if (update) {
            static_for<0, W>([&](auto w) {
                if (wave == w) {
                    static_for<0, NBW>([&](auto j) {
                        if constexpr (j < PL::COLS.cnt[w]) {
                            constexpr int bc = PL::COLS.item[w][j];
                            constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0;
                            T r[D > 0 ? D : 1], qn[D > 0 ? D : 1];
                            unsigned off[D > 0 ? D : 1];
                            static_for<0, D>([&](auto k) {
                                constexpr int e = C::col_slot[k0 + k], s = C::col_shift[k0 + k];
                                off[k] = e * Z * ST + rot_bwd<Z, ST, s>(zb);
                                r[k] = *reinterpret_cast<const T *>(msg_b + off[k]);
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
                                qn[k] = L[j] + ext;
                            });
                            static_for<0, (D > 0 ? D - 1 : 0)>([&](auto k) {
                                if constexpr (C::col_pair[k0 + k] != 0) {
                                    const T a = qn[k], b = qn[k + 1];
                                    qn[k] = sw[k] ? b : a;
                                    qn[k + 1] = sw[k] ? a : b;
                                }
                            });
                            bool refresh = true;
                                if constexpr (REF) refresh = bc * Z + z < g.num_refresh_vars;
                                static_for<0, D>([&](auto k) { *reinterpret_cast<T *>(msg_b + off[k]) = refresh ? qn[k] : L[j]; });
                            T b = L[j] + run;
                            if (g.clip_post > (T)0) b = clip(b, -g.clip_post, g.clip_post);
                            own[j] = b;
                            hb_b[bc * Z + z1] = (uint8_t)!(b > (T)0);
                        }
                    });
                }
            });
        }
        // ---- a frame ends with a zero syndrome, or after its last update without a further test
        const bool ends = converged || (update && t == g.max_iters - 1);
        if (ends) {
            static_for<0, W>([&](auto w) {
                if (wave == w) {
                    static_for<0, NBW>([&](auto j) {
                        if constexpr (j < PL::COLS.cnt[w]) {
                            constexpr int bc = PL::COLS.item[w][j];
                            hard_out[frame * n + bc * Z + z] = (uint8_t)hard_bit<RULE>(own[j]);
                            if (post_out) post_out[frame * n + bc * Z + z] = own[j];
                        }
                    });
                }
            });
            if (wave == 0 && z == 0) { iters_out[frame] = t; conv_out[frame] = (uint8_t)converged; }
            frame += stride;
            t = 0;
            live = frame < frames;
            if (live) load_frame();
        } else if (update) {
            ++t;
        }
        static_for<0, F>([&](auto ff) {  // the uniform copies take the same steps
            if (frame_s[ff] < frames) {
                if (ok_s[ff] || t_s[ff] == g.max_iters - 1) { frame_s[ff] += stride; t_s[ff] = 0; }
                else ++t_s[ff];
            }
        });
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ host
// built for the large base graphs (the reference's own call shape: its 5G matrix under a constructor n of 648,
// channel.py:646,842); elsewhere the table-driven kernels serve the bound
template <class C>
constexpr bool spaw_refresh_built() { return C::EB > 100; }

#ifdef LDPC_SPEC_FUSED
constexpr bool kFused = true;
#else
constexpr bool kFused = false;
#endif

template <int RULE, class C, typename T, typename TL, int F, int W>
static hipError_t launch_spaw(const ldpc_decode_config &cfg, const Launch &io, int num_cus, const char **name, bool *supported,
                              const SweepDev *sp = nullptr, Scratch *ws = nullptr)
{
    typedef SpawLayout<C, T, F> LY;
    constexpr int G = (LY::LANES + 63) / 64;
    const bool bounded = cfg.num_refresh_vars >= 0 && cfg.num_refresh_vars < C::NB * C::Z;
    Cfg<T> g;
    g.rule = RULE; g.max_iters = cfg.max_iters; g.early_exit = cfg.early_exit;
    g.float_syndrome0 = 0; g.num_active_checks = -1; g.num_refresh_vars = bounded ? cfg.num_refresh_vars : -1;
    g.scale = (T)cfg.scale; g.clip_tanh_arg = (T)cfg.clip_tanh_arg; g.clip_prod = (T)cfg.clip_prod;
    g.clip_post = (T)cfg.clip_post; g.offset = (T)cfg.offset;
    if constexpr (kFused) {  // fused sweep: float LLRs drawn in the kernel, no loop bound; the slots start and end together
        if constexpr (sizeof(TL) == 4 && (C::NB * C::Z) % 4 == 0) {
            if (bounded || !sp || !ws) { *supported = false; return hipSuccess; }
            auto kf = spec_spaw_kernel<RULE, C, T, TL, F, W, false, true>;
            int per_cu = 0;
            hipError_t e = kernel_setup(reinterpret_cast<const void *>(kf), 64 * W * G, LY::LDS_BYTES, &per_cu);
            if (e != hipSuccess) return e;
            if (per_cu < 1) per_cu = 1;
            const long long groups = (io.frames + F - 1) / F, full = (long long)num_cus * per_cu * 2;
            long long grid = full > groups ? groups : full;
            if (grid < 1) grid = 1;
            e = scratch_reserve(*ws, sweep_slice_bytes<C::NB * C::Z, F>(), full, grid);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(64 * W * G), LY::LDS_BYTES, io.stream, g, nullptr, io.frames, nullptr,
                               nullptr, nullptr, nullptr, *sp, reinterpret_cast<float *>(ws->ptr));
            *name = sizeof(T) == 8 ? "fused_spec_spaw_f64" : "fused_spec_spaw_f32";
            return hipGetLastError();
        } else {
            *supported = false;
            return hipSuccess;
        }
    } else {
        auto k = spec_spaw_kernel<RULE, C, T, TL, F, W>;
        if constexpr (F > 1) {  // early exit with several slots: every slot runs its own queue of frames
            if (cfg.early_exit && cfg.max_iters >= 1) k = spec_spaw_slots_kernel<RULE, C, T, TL, F, W>;
        }
        if (bounded) {
            if constexpr (spaw_refresh_built<C>()) {
                k = spec_spaw_kernel<RULE, C, T, TL, F, W, true>;
                if constexpr (F > 1) {
                    if (cfg.early_exit && cfg.max_iters >= 1) k = spec_spaw_slots_kernel<RULE, C, T, TL, F, W, true>;
                }
            } else {
                *supported = false;
                return hipSuccess;
            }
        }
        int per_cu = 0;
        hipError_t e = kernel_setup(reinterpret_cast<const void *>(k), 64 * W * G, LY::LDS_BYTES, &per_cu);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        const long long groups = (io.frames + F - 1) / F;
        long long grid = (long long)num_cus * per_cu * 2;
        if (grid > groups) grid = groups;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(64 * W * G), LY::LDS_BYTES, io.stream, g,
                           reinterpret_cast<const TL *>(io.llr), io.frames, io.hard, io.iters, io.converged,
                           reinterpret_cast<T *>(io.posterior), SweepDev{}, nullptr);
        *name = sizeof(T) == 8 ? "spec_spaw_f64" : "spec_spaw_f32";
        return hipGetLastError();
    }
}

// Codes and arithmetic types that come here: see the head of this file.
template <class C, typename T, int F>
constexpr bool spaw_wanted()
{
    constexpr int single = F * C::EB * C::Z * (int)sizeof(T);
    return C::Z <= 64 && F * C::Z <= 128 && SpawLayout<C, T, F>::LDS_BYTES <= 160 * 1024 && C::EB <= 200 &&
           (C::EB > 100 || (160 * 1024) / single < 8);
}

#ifndef LDPC_SPAW_WAVES
#define LDPC_SPAW_WAVES 4
#endif
#ifndef LDPC_SPAW_WAVES_PACKED
#define LDPC_SPAW_WAVES_PACKED 8  // per lane group, when a workgroup carries two lane groups
#endif

// Frame slots per workgroup: as in decode_spec_r1w.hip (three Z = 40 frames over two lane groups
// where LDS holds them).
template <class C, typename T, int SLOTS>
constexpr int spaw_slots()
{
    if (SLOTS * C::Z <= 64) return SLOTS;
    if (SLOTS * C::Z <= 128 && SpawLayout<C, T, SLOTS>::LDS_BYTES <= 160 * 1024) return SLOTS;
    return 1;
}

template <int RULE, class C, int SLOTS>
static hipError_t launch_spaw_rule(const ldpc_decode_config &cfg, int compute_dtype, const Launch &io, int num_cus,
                                   const char **name, bool *supported, const SweepDev *sp = nullptr, Scratch *ws = nullptr)
{
    constexpr int FD = spaw_slots<C, double, SLOTS>(), FF = spaw_slots<C, float, SLOTS>();
    *supported = true;
    if constexpr (spaw_wanted<C, double, FD>()) {
        constexpr int W = FD * C::Z > 64 ? LDPC_SPAW_WAVES_PACKED : LDPC_SPAW_WAVES;
        if (compute_dtype == LDPC_F64) {
            if constexpr (!kFused) {
                if (io.llr_dtype == LDPC_F64) return launch_spaw<RULE, C, double, double, FD, W>(cfg, io, num_cus, name, supported);
            }
            return launch_spaw<RULE, C, double, float, FD, W>(cfg, io, num_cus, name, supported, sp, ws);
        }
    }
    if constexpr (spaw_wanted<C, float, FF>()) {
        constexpr int W = FF * C::Z > 64 ? LDPC_SPAW_WAVES_PACKED : LDPC_SPAW_WAVES;
        if (compute_dtype == LDPC_F32 && (kFused || io.llr_dtype == LDPC_F32))
            return launch_spaw<RULE, C, float, float, FF, W>(cfg, io, num_cus, name, supported, sp, ws);
    }
    *supported = false;
    return hipSuccess;
}

// One translation unit per rule (-DLDPC_SPAW_RULE=2, 3, 4) so that the units compile side by side.
#ifndef LDPC_SPAW_RULE
#error "compile with -DLDPC_SPAW_RULE=2, 3 or 4"
#endif
#define LDPC_SPAW_CAT2(a, b) a##b
#define LDPC_SPAW_CAT(a, b) LDPC_SPAW_CAT2(a, b)

#ifdef LDPC_SPEC_FUSED
// the same kernels as one launch of channel + decode + error accounting (ldpc_sweep_batch)
hipError_t LDPC_SPAW_CAT(launch_spec_spaw_fused_rule_, LDPC_SPAW_RULE)(int spec_id, const ldpc_decode_config &cfg,
                                                                       int compute_dtype, const Launch &io, const SweepDev &sp,
                                                                       int num_cus, Scratch &workspace, const char **name,
                                                                       bool *supported)
{
    int id = 0;
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) {                                                                           \
        return launch_spaw_rule<LDPC_SPAW_RULE, CODE, SLOTS>(cfg, compute_dtype, io, num_cus, name, supported, &sp, &workspace); \
    }                                                                                              \
    ++id;
    LDPC_FOR_EACH_SPEC_CODE(LDPC_LAUNCH)
#undef LDPC_LAUNCH
    *supported = false;
    return hipSuccess;
}
#else
hipError_t LDPC_SPAW_CAT(launch_spec_spaw_rule_, LDPC_SPAW_RULE)(int spec_id, const ldpc_decode_config &cfg,
                                                                 int compute_dtype, const Launch &io, int num_cus,
                                                                 const char **name, bool *supported)
{
    int id = 0;
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) {                                                                           \
        return launch_spaw_rule<LDPC_SPAW_RULE, CODE, SLOTS>(cfg, compute_dtype, io, num_cus, name, supported); \
    }                                                                                              \
    ++id;
    LDPC_FOR_EACH_SPEC_CODE(LDPC_LAUNCH)
#undef LDPC_LAUNCH
    *supported = false;
    return hipSuccess;
}
#endif

}  // namespace ldpc
