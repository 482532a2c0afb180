// Rule R1 (bpsk.py:154-260, extrinsic normalised min-sum) on the compile-time path for codes whose
// per-frame state is too large for the one-wave-per-workgroup kernel of decode_spec.hip to keep a
// compute unit busy (802.11n in double: three waves per CU; the BG2-shaped 5G matrix: one or two).
//
// W waves share the F frame slots of a workgroup.  Every wave maps its lanes to (slot, z) in the
// same way; every wave owns a share of the block rows in the check phase and a share of the block
// columns in the variable phase (balanced by edge count at compile time), each share as its own
// unrolled code path.  Per-variable state (channel LLR, posterior, own variable-to-check messages)
// sits in registers indexed by the wave's
// LOCAL column number, so a wave only pays registers for its share of the graph.
//
// LDS per slot:  qv [EB][Z]  variable-to-check messages in variable order (written linearly by the
//                            variable, read rotated by the check)
//                cs [MB][Z]  per check (m1, m2) with the sign parity in both sign bits
//                hb [NB][Z]  hard decisions, one byte each (`0 if b > 0 else 1`, bpsk.py:240-243)
// One iteration: check phase (which also forms the parity of the current hard decisions - the
// syndrome test at the top of the iteration, bpsk.py:205-209) | barrier | slots whose syndrome is
// zero stop here, before their update | variable phase | barrier.
#include "spec_r1w_plan.h"
#include "sweep_fused.h"

namespace ldpc {

template <class C, typename T, int F>
struct R1wLayout {
    static constexpr int LANES = F * C::Z;  // must be <= 64: a wave covers all slots
    static constexpr int QV_BYTES = C::EB * C::Z * (int)sizeof(T);
    static constexpr int CS_BYTES = C::MB * C::Z * 2 * (int)sizeof(T);
    static constexpr int HB_BYTES = ((C::NB * C::Z + 15) / 16) * 16;
    static constexpr int SLOT_BYTES = ((QV_BYTES + CS_BYTES + HB_BYTES + 127) / 128) * 128 + (F > 1 ? 32 : 0);
    static constexpr int FLAG_OFF = F * SLOT_BYTES;
    static constexpr int LDS_BYTES = FLAG_OFF + 96;  // flags: two rows of eight (syndromes), one more for the fused sweep
};

// REF: the refresh loop of the reference runs over range(self.n), the CONSTRUCTOR's n (bpsk.py:244,
// channel.py:241): variables from g.num_refresh_vars on keep sending their channel value.  A switch of
// its own so that the plain kernel does not pay the comparison and the selects.
// FUSED: channel in front and error accounting behind inside the kernel (sweep_fused.h).  The workgroup
// draws the LLRs of its F frames into its own slice of `scratch` ([F][n] floats + [F][n] reference bytes in device memory,
// reused group after group, hence cache-resident: beside three frames' messages LDS has no room for
// them) and decodes from there; every lane counts the bit errors of the variables it owns against the
// transmitted word, the per-frame flags meet in LDS.  llr and the result arrays are unused.
template <class C, typename T, typename TL, int F, int W, bool OMS, bool REF = false, bool FUSED = false>  // OMS: the offset rule (LDPC_RULE_OMS) instead of R1's
__global__ void __launch_bounds__(64 * W * ((F * C::Z + 63) / 64))
spec_r1w_kernel(Cfg<T> g, const TL *__restrict__ llr, long long frames, uint8_t *__restrict__ hard_out,
                int *__restrict__ iters_out, uint8_t *__restrict__ conv_out, T *__restrict__ post_out, SweepDev sp,
                float *scratch)
{
    typedef R1wLayout<C, T, F> LY;
    typedef R1wPlan<C, W> PL;
    typedef typename SNum<T>::Pair Pair;
    constexpr int G = (LY::LANES + 63) / 64;  // lane groups: G waves side by side cover the F slots
    constexpr int Z = C::Z, NB = C::NB, MB = C::MB, ST = sizeof(T), SP = 2 * sizeof(T);
    constexpr int n = NB * Z, NBW = PL::NBW, QW = PL::QW, MBW = PL::MBW;
    extern __shared__ __align__(16) unsigned char smem[];
    static_assert(F <= 8, "the per-slot syndrome flags are two rows of eight");

    const int wave = ((int)threadIdx.x >> 6) / G, lane = (((int)threadIdx.x >> 6) % G) * 64 + ((int)threadIdx.x & 63);
    const bool lane_on = lane < F * Z;
    const int f = lane_on ? lane / Z : 0;
    const int z = lane_on ? lane - f * Z : 0;
    unsigned char *qv_b = smem + f * LY::SLOT_BYTES;          // [EB][Z] T
    unsigned char *cs_b = qv_b + LY::QV_BYTES;                // [MB][Z] Pair
    unsigned char *hb_b = cs_b + LY::CS_BYTES;                // [NB][Z] u8
    int *flags = reinterpret_cast<int *>(smem + LY::FLAG_OFF);  // [2][8]: failed-syndrome flag per slot, double-buffered by iteration parity
    const unsigned zb = (unsigned)z * ST, zp = (unsigned)z * SP, z1 = (unsigned)z;
    SweepAcc acc;

    const long long groups = (frames + F - 1) / F;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long frame = grp * F + f;
        const bool active = lane_on && frame < frames;
        const float *mine = nullptr;
        if constexpr (FUSED) {
            mine = sweep_stage_group<n, F>(sp, scratch, grp, frames);
        }
        T L[NBW], own[NBW], q[QW];
        // ---- start: every edge carries the channel LLR (bpsk.py:170-177)
        static_for<0, W>([&](auto w) {
            if (wave == w) {
                static_for<0, NBW>([&](auto j) {
                    if constexpr (j < PL::COLS.cnt[w]) {
                        constexpr int bc = PL::COLS.item[w][j];
                        constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0, q0 = PL::COLS.edge0[w][j];
                        T l;
                        if constexpr (FUSED) l = (active ? (T)mine[f * n + bc * Z + z] : (T)0) + (T)0;
                        else l = (active ? (T)llr[frame * n + bc * Z + z] : (T)0) + (T)0;  // -0.0 -> +0.0: signs are sign bits from here on
                        L[j] = l;
                        own[j] = l;
                        if (lane_on) hb_b[bc * Z + z1] = (uint8_t)hard_bit<1>(l);  // idle lanes alias lane 0
                        static_for<0, D>([&](auto k) {
                            q[q0 + k] = l;
                            if (lane_on) *reinterpret_cast<T *>(qv_b + (k0 + k) * Z * ST + zb) = l;
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
            if (last && g.early_exit) break;  // no test after the last update (bpsk.py:200-248)
            const bool want_syn = g.early_exit || last;
            // ---- check nodes of this wave's block rows (bpsk.py:213-233) and the parity of the
            //      current hard decisions
            unsigned bad = 0;
            if (!done) {
                static_for<0, W>([&](auto w) {
                    if (wave == w) {
                        static_for<0, MBW>([&](auto i) {
                            if constexpr (i < PL::ROWS.cnt[w]) {
                                constexpr int br = PL::ROWS.item[w][i];
                                constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0;
                                T m1 = Lim<T>::inf(), m2 = Lim<T>::inf();
                                unsigned sx = 0, par = 0;
                                static_for<0, D>([&](auto p) {
                                    constexpr int kpos = C::row_kpos[e0 + p], s = C::row_shift[e0 + p], bc = C::row_col[e0 + p];
                                    const T v = *reinterpret_cast<const T *>(qv_b + kpos * Z * ST + rot_fwd<Z, ST, s>(zb));
                                    if (want_syn) par ^= hb_b[bc * Z + rot_fwd<Z, 1, s>(z1)];
                                    SNum<T>::template book<p>(m1, m2, v);
                                    sx ^= SNum<T>::sign_word(v);
                                });
                                if constexpr (D < 2) { m1 = (T)0; m2 = (T)0; }  // bpsk.py:216
                                if (!last)
                                    *reinterpret_cast<Pair *>(cs_b + br * Z * SP + zp) =
                                        SNum<T>::make(SNum<T>::with_sign(m1, sx), SNum<T>::with_sign(m2, sx));
                                bad |= par;
                            }
                        });
                    }
                });
            }
            int *slot_bad = flags + (t & 1) * 8;
            if (bad) slot_bad[f] = 1;
            __syncthreads();
            // the other buffer was last read one iteration ago; the next iteration's writes to it
            // come after the barrier that closes this iteration
            if (threadIdx.x < 8) flags[((t + 1) & 1) * 8 + threadIdx.x] = 0;
            if (want_syn) {
                const bool slot_ok = slot_bad[f] == 0;
                if (last) {
                    if (!done) res_conv = slot_ok ? 1 : 0;
                    break;
                }
                if (!done && slot_ok) { res_iters = t; res_conv = 1; done = true; }  // bpsk.py:205-209
                bool any_live = false;
                static_for<0, F>([&](auto ff) { any_live |= slot_bad[ff] != 0; });
                if (!any_live) break;  // every slot that was still running has just converged
            }
            // ---- variable nodes of this wave's block columns (bpsk.py:235-247)
            if (!done) {
                static_for<0, W>([&](auto w) {
                    if (wave == w) {
                        static_for<0, NBW>([&](auto j) {
                            if constexpr (j < PL::COLS.cnt[w]) {
                                constexpr int bc = PL::COLS.item[w][j];
                                constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0, q0 = PL::COLS.edge0[w][j];
                                T r[D > 0 ? D : 1], qn[D > 0 ? D : 1];
                                unsigned off[D > 0 ? D : 1];
                                static_for<0, D>([&](auto k) {
                                    constexpr int br = C::col_row[k0 + k], s = C::col_shift[k0 + k];
                                    const T qo = q[q0 + k];
                                    off[k] = rot_bwd<Z, SP, s>(zp);
                                    const Pair c = *reinterpret_cast<const Pair *>(cs_b + br * Z * SP + off[k]);
                                    const T sel = t_abs(qo) == t_abs(c.x) ? c.y : c.x;
                                    r[k] = SNum<T>::flip_sign(OMS ? SNum<T>::flip_sign(oms_magnitude(t_abs(sel), g.scale, g.offset), SNum<T>::sign_word(sel) & 0x80000000u)
                                                                               : g.scale * sel,
                                                                           SNum<T>::sign_word(qo) & 0x80000000u);
                                });
                                bool sw[D > 0 ? D : 1];
                                static_for<0, (D > 0 ? D - 1 : 0)>([&](auto k) {
                                    if constexpr (C::col_pair[k0 + k] != 0) {  // lower check index first
                                        sw[k] = off[k] > off[k + 1];
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
                                static_for<0, D>([&](auto k) {
                                    const T qw = refresh ? qn[k] : L[j];
                                    q[q0 + k] = qw;
                                    *reinterpret_cast<T *>(qv_b + (k0 + k) * Z * ST + zb) = qw;
                                });
                                own[j] = L[j] + run;  // bpsk.py:237
                                hb_b[bc * Z + z1] = (uint8_t)hard_bit<1>(own[j]);
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
                                errs += hard_bit<1>(own[j]) != (unsigned)rf[bc * Z + z];
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
                            hard_out[frame * n + bc * Z + z] = (uint8_t)hard_bit<1>(own[j]);
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

// ------------------------------------------------------------------------------ R1, independent slots
// Early-exit decoding with several slots per workgroup: frames need very different numbers of
// iterations (one to thirty), and a workgroup that starts its slots together waits for the slowest.
// Here every slot runs its own queue of frames - slot s of workgroup b decodes frames
// (b + k * gridDim.x) * F + s, k = 0, 1, ... - and moves on to its next frame the moment the current
// one ends, while the other slots carry on; the slots only share the two barriers per iteration.
// Every wave tracks all slots' iteration counters and queue positions as uniform values (they
// follow from the syndrome flags every wave reads), so the loop ends for all waves together when
// the last queue is empty.  Same arithmetic as spec_r1w_kernel; g.early_exit != 0, max_iters >= 1.
template <class C, typename T, typename TL, int F, int W, bool OMS, bool REF = false>  // OMS, REF: as above
__global__ void __launch_bounds__(64 * W * ((F * C::Z + 63) / 64))
spec_r1w_slots_kernel(Cfg<T> g, const TL *__restrict__ llr, long long frames, uint8_t *__restrict__ hard_out,
                      int *__restrict__ iters_out, uint8_t *__restrict__ conv_out, T *__restrict__ post_out, SweepDev, float *)
{
    typedef R1wLayout<C, T, F> LY;
    typedef R1wPlan<C, W> PL;
    typedef typename SNum<T>::Pair Pair;
    constexpr int G = (LY::LANES + 63) / 64;
    constexpr int Z = C::Z, NB = C::NB, ST = sizeof(T), SP = 2 * sizeof(T);
    constexpr int n = NB * Z, NBW = PL::NBW, QW = PL::QW, MBW = PL::MBW;
    extern __shared__ __align__(16) unsigned char smem[];
    static_assert(F <= 8, "the per-slot syndrome flags are two rows of eight");

    const int wave = ((int)threadIdx.x >> 6) / G, lane = (((int)threadIdx.x >> 6) % G) * 64 + ((int)threadIdx.x & 63);
    const bool lane_on = lane < F * Z;
    const int f = lane_on ? lane / Z : 0;
    const int z = lane_on ? lane - f * Z : 0;
    unsigned char *qv_b = smem + f * LY::SLOT_BYTES;
    unsigned char *cs_b = qv_b + LY::QV_BYTES;
    unsigned char *hb_b = cs_b + LY::CS_BYTES;
    int *flags = reinterpret_cast<int *>(smem + LY::FLAG_OFF);
    const unsigned zb = (unsigned)z * ST, zp = (unsigned)z * SP, z1 = (unsigned)z;

    const long long stride = (long long)gridDim.x * F;
    // uniform view of every slot: current frame and iteration
    long long frame_s[F];
    int t_s[F];
    static_for<0, F>([&](auto ff) { frame_s[ff] = (long long)blockIdx.x * F + ff; t_s[ff] = 0; });
    // this lane's slot
    long long frame = (long long)blockIdx.x * F + f;
    int t = 0;
    bool live = lane_on && frame < frames;

    T L[NBW], own[NBW], q[QW];
    // every edge of the slot's new frame carries the channel LLR (bpsk.py:170-177)
    auto load_frame = [&]() {
        static_for<0, W>([&](auto w) {
            if (wave == w) {
                static_for<0, NBW>([&](auto j) {
                    if constexpr (j < PL::COLS.cnt[w]) {
                        constexpr int bc = PL::COLS.item[w][j];
                        constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0, q0 = PL::COLS.edge0[w][j];
                        const T l = (T)llr[frame * n + bc * Z + z] + (T)0;  // -0.0 -> +0.0
                        L[j] = l;
                        own[j] = l;
                        hb_b[bc * Z + z1] = (uint8_t)hard_bit<1>(l);
                        static_for<0, D>([&](auto k) {
                            q[q0 + k] = l;
                            *reinterpret_cast<T *>(qv_b + (k0 + k) * Z * ST + zb) = l;
                        });
                    }
                });
            }
        });
    };
    static_for<0, NBW>([&](auto j) { L[j] = (T)0; own[j] = (T)0; });
    static_for<0, QW>([&](auto k) { q[k] = (T)0; });
    if (live) load_frame();
    if (threadIdx.x < 16) flags[threadIdx.x] = 0;
    __syncthreads();

    for (int round = 0;; ++round) {
        bool any_live = false;
        static_for<0, F>([&](auto ff) { any_live |= frame_s[ff] < frames; });
        if (!any_live) break;  // uniform: every queue is empty
        // ---- check nodes of this wave's block rows and the parity of the current hard decisions
        unsigned bad = 0;
        if (live) {
            static_for<0, W>([&](auto w) {
                if (wave == w) {
                    static_for<0, MBW>([&](auto i) {
                        if constexpr (i < PL::ROWS.cnt[w]) {
                            constexpr int br = PL::ROWS.item[w][i];
                            constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0;
                            T m1 = Lim<T>::inf(), m2 = Lim<T>::inf();
                            unsigned sx = 0, par = 0;
                            static_for<0, D>([&](auto p) {
                                constexpr int kpos = C::row_kpos[e0 + p], s = C::row_shift[e0 + p], bc = C::row_col[e0 + p];
                                const T v = *reinterpret_cast<const T *>(qv_b + kpos * Z * ST + rot_fwd<Z, ST, s>(zb));
                                par ^= hb_b[bc * Z + rot_fwd<Z, 1, s>(z1)];
                                SNum<T>::template book<p>(m1, m2, v);
                                sx ^= SNum<T>::sign_word(v);
                            });
                            if constexpr (D < 2) { m1 = (T)0; m2 = (T)0; }  // bpsk.py:216
                            *reinterpret_cast<Pair *>(cs_b + br * Z * SP + zp) =
                                SNum<T>::make(SNum<T>::with_sign(m1, sx), SNum<T>::with_sign(m2, sx));
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
        // ---- outcome per slot, the same in every wave
        bool ok_s[F];
        static_for<0, F>([&](auto ff) { ok_s[ff] = __builtin_amdgcn_readfirstlane(slot_bad[ff]) == 0; });
        bool slot_ok = ok_s[0];
        static_for<1, F>([&](auto ff) { if (f == ff) slot_ok = ok_s[ff]; });
        const bool converged = live && slot_ok;   // bpsk.py:205-209
        const bool update = live && !slot_ok;
        // ---- variable nodes of this wave's block columns (bpsk.py:235-247)
        if (update) {
            static_for<0, W>([&](auto w) {
                if (wave == w) {
                    static_for<0, NBW>([&](auto j) {
                        if constexpr (j < PL::COLS.cnt[w]) {
                            constexpr int bc = PL::COLS.item[w][j];
                            constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0, q0 = PL::COLS.edge0[w][j];
                            T r[D > 0 ? D : 1], qn[D > 0 ? D : 1];
                            unsigned off[D > 0 ? D : 1];
                            static_for<0, D>([&](auto k) {
                                constexpr int br = C::col_row[k0 + k], s = C::col_shift[k0 + k];
                                const T qo = q[q0 + k];
                                off[k] = rot_bwd<Z, SP, s>(zp);
                                const Pair c = *reinterpret_cast<const Pair *>(cs_b + br * Z * SP + off[k]);
                                const T sel = t_abs(qo) == t_abs(c.x) ? c.y : c.x;
                                r[k] = SNum<T>::flip_sign(OMS ? SNum<T>::flip_sign(oms_magnitude(t_abs(sel), g.scale, g.offset), SNum<T>::sign_word(sel) & 0x80000000u)
                                                                               : g.scale * sel,
                                                                           SNum<T>::sign_word(qo) & 0x80000000u);
                            });
                            bool sw[D > 0 ? D : 1];
                            static_for<0, (D > 0 ? D - 1 : 0)>([&](auto k) {
                                if constexpr (C::col_pair[k0 + k] != 0) {  // lower check index first
                                    sw[k] = off[k] > off[k + 1];
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
                            static_for<0, D>([&](auto k) {
                                const T qw = refresh ? qn[k] : L[j];
                                q[q0 + k] = qw;
                                *reinterpret_cast<T *>(qv_b + (k0 + k) * Z * ST + zb) = qw;
                            });
                            own[j] = L[j] + run;  // bpsk.py:237
                            hb_b[bc * Z + z1] = (uint8_t)hard_bit<1>(own[j]);
                        }
                    });
                }
            });
        }
        // ---- a frame ends with a zero syndrome, or after its last update without a further test
        //      (bpsk.py:200-248); both ways numIterations is the current counter
        const bool ends = converged || (update && t == g.max_iters - 1);
        if (ends) {
            static_for<0, W>([&](auto w) {
                if (wave == w) {
                    static_for<0, NBW>([&](auto j) {
                        if constexpr (j < PL::COLS.cnt[w]) {
                            constexpr int bc = PL::COLS.item[w][j];
                            hard_out[frame * n + bc * Z + z] = (uint8_t)hard_bit<1>(own[j]);
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
        // the uniform copies take the same steps
        static_for<0, F>([&](auto ff) {
            if (frame_s[ff] < frames) {
                if (ok_s[ff] || t_s[ff] == g.max_iters - 1) { frame_s[ff] += stride; t_s[ff] = 0; }
                else ++t_s[ff];
            }
        });
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ R0
// Rule R0 (ldpc.py:137-227) with the same division of work.  Its variable-to-check message is the
// posterior itself (ldpc.py:180-187 reads bitNodes), so LDS per slot only holds
//     post [NB][Z]  posteriors      cs [MB][Z]  per check (m1, m2) with the sign parity
// and the parity of the hard decisions `1 - signbit` is a by-product of the sign word of the check.
template <class C, typename T, int F>
struct R0wLayout {
    static constexpr int LANES = F * C::Z;
    static constexpr int POST_BYTES = C::NB * C::Z * (int)sizeof(T);
    static constexpr int CS_BYTES = C::MB * C::Z * 2 * (int)sizeof(T);
    static constexpr int SLOT_BYTES = ((POST_BYTES + CS_BYTES + 127) / 128) * 128 + (F > 1 ? 32 : 0);
    static constexpr int FLAG_OFF = F * SLOT_BYTES;
    static constexpr int LDS_BYTES = FLAG_OFF + 96;  // flags: two rows of eight (syndromes), one more for the fused sweep
};

template <class C, typename T, typename TL, int F, int W, bool FUSED = false>  // FUSED: as in spec_r1w_kernel
__global__ void __launch_bounds__(64 * W * ((F * C::Z + 63) / 64))
spec_r0w_kernel(Cfg<T> g, const TL *__restrict__ llr, long long frames, uint8_t *__restrict__ hard_out,
                int *__restrict__ iters_out, uint8_t *__restrict__ conv_out, T *__restrict__ post_out, SweepDev sp,
                float *scratch)
{
    typedef R0wLayout<C, T, F> LY;
    typedef R1wPlan<C, W> PL;
    typedef typename SNum<T>::Pair Pair;
    constexpr int G = (LY::LANES + 63) / 64;  // lane groups: G waves side by side cover the F slots
    constexpr int Z = C::Z, NB = C::NB, ST = sizeof(T), SP = 2 * sizeof(T);
    constexpr int n = NB * Z, NBW = PL::NBW, MBW = PL::MBW;
    extern __shared__ __align__(16) unsigned char smem[];
    static_assert(F <= 8, "the per-slot syndrome flags are two rows of eight");

    const int wave = ((int)threadIdx.x >> 6) / G, lane = (((int)threadIdx.x >> 6) % G) * 64 + ((int)threadIdx.x & 63);
    const bool lane_on = lane < F * Z;
    const int f = lane_on ? lane / Z : 0;
    const int z = lane_on ? lane - f * Z : 0;
    unsigned char *post_b = smem + f * LY::SLOT_BYTES;  // [NB][Z] T
    unsigned char *cs_b = post_b + LY::POST_BYTES;      // [MB][Z] Pair
    int *flags = reinterpret_cast<int *>(smem + LY::FLAG_OFF);  // [2][8], as in the R1 kernel
    const unsigned zb = (unsigned)z * ST, zp = (unsigned)z * SP;
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
                        // + 0.0 turns -0.0 into +0.0: from here on "negative" is exactly "sign bit set"
                        T l;
                        if constexpr (FUSED) l = (active ? (T)mine[f * n + bc * Z + z] : (T)0) + (T)0;
                        else l = (active ? (T)llr[frame * n + bc * Z + z] : (T)0) + (T)0;
                        L[j] = l;
                        own[j] = l;  // ldpc.py:154
                        if (lane_on) *reinterpret_cast<T *>(post_b + bc * Z * ST + zb) = l;  // idle lanes alias lane 0
                    }
                });
            }
        });
        if (threadIdx.x < 24) flags[threadIdx.x] = 0;
        __syncthreads();

        bool done = !active, float_hard = false;
        int res_iters = g.early_exit ? g.max_iters - 1 : g.max_iters, res_conv = 0;
        for (int t = 0;; ++t) {
            const bool last = t == g.max_iters;
            if (last && g.early_exit) break;
            const bool want_syn = g.early_exit || last;
            const bool float_check = t == 0 && g.float_syndrome0 && g.early_exit;  // ldpc.py:158,165
            // ---- check nodes of this wave's block rows (ldpc.py:175-187) and the syndrome
            unsigned bad = 0;
            auto check_phase = [&](auto float_chk) {  // the float test only exists at t == 0: its sums stay out of the loop's code
                constexpr bool FC = decltype(float_chk)::value != 0;
                static_for<0, W>([&](auto w) {
                    if (wave == w) {
                        static_for<0, MBW>([&](auto i) {
                            if constexpr (i < PL::ROWS.cnt[w]) {
                                constexpr int br = PL::ROWS.item[w][i];
                                constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0;
                                T m1 = Lim<T>::inf(), m2 = Lim<T>::inf(), fsum = (T)0;
                                unsigned sx = 0;
                                static_for<0, D>([&](auto p) {
                                    constexpr int bc = C::row_col[e0 + p], s = C::row_shift[e0 + p];
                                    const T b = *reinterpret_cast<const T *>(post_b + bc * Z * ST + rot_fwd<Z, ST, s>(zb));
                                    SNum<T>::template book<p>(m1, m2, b);
                                    sx ^= SNum<T>::sign_word(b);
                                    if constexpr (FC) fsum += b;
                                });
                                if constexpr (D < 2) { m1 = (T)0; m2 = (T)0; }  // ldpc.py:178
                                if (!last)
                                    *reinterpret_cast<Pair *>(cs_b + br * Z * SP + zp) =
                                        SNum<T>::make(SNum<T>::with_sign(m1, sx), SNum<T>::with_sign(m2, sx));
                                if constexpr (FC) bad |= !even_integer(fsum);
                                else bad |= ((unsigned)D ^ (sx >> 31)) & 1u;  // parity of the hard bits 1 - signbit
                            }
                        });
                    }
                });
            };
            if (!done) {
                if (float_check) check_phase(IC<1>{});
                else check_phase(IC<0>{});
            }
            int *slot_bad = flags + (t & 1) * 8;
            if (bad && want_syn) slot_bad[f] = 1;
            __syncthreads();
            if (threadIdx.x < 8) flags[((t + 1) & 1) * 8 + threadIdx.x] = 0;
            if (want_syn) {
                const bool slot_ok = slot_bad[f] == 0;
                if (last) {
                    if (!done) res_conv = slot_ok ? 1 : 0;
                    break;
                }
                if (!done && slot_ok) { res_iters = t; res_conv = 1; done = true; float_hard = float_check; }  // ldpc.py:165-170
                bool any_live = false;
                static_for<0, F>([&](auto ff) { any_live |= slot_bad[ff] != 0; });
                if (!any_live) break;
            }
            // ---- variable nodes of this wave's block columns (ldpc.py:173,188-196)
            if (!done) {
                static_for<0, W>([&](auto w) {
                    if (wave == w) {
                        static_for<0, NBW>([&](auto j) {
                            if constexpr (j < PL::COLS.cnt[w]) {
                                constexpr int bc = PL::COLS.item[w][j];
                                constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0;
                                const unsigned so = SNum<T>::sign_word(own[j]) & 0x80000000u;
                                const T ao = t_abs(own[j]);
                                T r[D > 0 ? D : 1];
                                unsigned off[D > 0 ? D : 1];
                                static_for<0, D>([&](auto k) {
                                    constexpr int br = C::col_row[k0 + k], s = C::col_shift[k0 + k];
                                    off[k] = rot_bwd<Z, SP, s>(zp);
                                    const Pair c = *reinterpret_cast<const Pair *>(cs_b + br * Z * SP + off[k]);
                                    r[k] = ao == t_abs(c.x) ? c.y : c.x;  // the variable's own sign is applied to the sum
                                });
                                static_for<0, (D > 0 ? D - 1 : 0)>([&](auto k) {
                                    if constexpr (C::col_pair[k0 + k] != 0) {  // lower check index first
                                        const bool sw = off[k] > off[k + 1];
                                        const T a = r[k], b = r[k + 1];
                                        r[k] = sw ? b : a;
                                        r[k + 1] = sw ? a : b;
                                    }
                                });
                                T sum = (T)0;
                                static_for<0, D>([&](auto k) { sum += r[k]; });
                                own[j] = g.scale * (SNum<T>::flip_sign(sum, so) + L[j]);  // ldpc.py:196
                                *reinterpret_cast<T *>(post_b + bc * Z * ST + zb) = own[j];
                            }
                        });
                    }
                });
            }
            __syncthreads();
        }
        if constexpr (FUSED) {
            unsigned errs = 0;
            if (active) {
                const uint8_t *rf = sweep_slice_ref<n, F>(mine) + f * n;  // the frame's transmitted bits, staged beside its LLRs
                static_for<0, W>([&](auto w) {
                    if (wave == w) {
                        static_for<0, NBW>([&](auto j) {
                            if constexpr (j < PL::COLS.cnt[w]) {
                                constexpr int bc = PL::COLS.item[w][j];
                                const T v = own[j];
                                const unsigned hv = float_hard ? (v == (T)0 ? 0u : (v == (T)1 ? 1u : 2u)) : hard_bit<0>(v);
                                errs += hv != (unsigned)rf[bc * Z + z];
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
                            const T v = own[j];
                            hard_out[frame * n + bc * Z + z] =
                                float_hard ? (v == (T)0 ? 0 : (v == (T)1 ? 1 : 2)) : (uint8_t)hard_bit<0>(v);
                            if (post_out) post_out[frame * n + bc * Z + z] = v;
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

// Frame slots per workgroup: the code's slot count when those frames fit one wave; when they need
// two waves side by side (Z = 40: three frames in 120 of 128 lanes instead of one in 40 of 64) only
// if LDS holds them; else one frame.
#ifndef LDPC_PACK_LDS
#define LDPC_PACK_LDS (160 * 1024)
#endif
template <class C, typename T, int SLOTS, template <class, typename, int> class LAYOUT>
constexpr int packed_slots()
{
    if (SLOTS * C::Z <= 64) return SLOTS;
    if (SLOTS * C::Z <= 128 && LAYOUT<C, T, SLOTS>::LDS_BYTES <= LDPC_PACK_LDS) return SLOTS;
    return 1;
}

// ------------------------------------------------------------------------------------------ host
#ifndef LDPC_R1W_RULE
#error "compile with -DLDPC_R1W_RULE=0, 1 or 6"
#endif
constexpr bool kOms = LDPC_R1W_RULE == 6;  // this translation unit builds the offset rule instead of R1

// The refresh bound is built for rule R1 on the large base graphs (the reference's own call shape is its
// 5G matrix under a constructor n of 648, channel.py:646); elsewhere the table-driven kernels serve it.
template <class C>
constexpr bool r1w_refresh_built() { return !kOms && C::EB > 100; }

template <class C, typename T, typename TL, int F, int W>
static hipError_t launch_r1w(const ldpc_decode_config &cfg, const Launch &io, int num_cus, const char **name, bool *supported,
                             const SweepDev *sp = nullptr, Scratch *ws = nullptr)
{
    typedef R1wLayout<C, T, F> LY;
    constexpr int G = (LY::LANES + 63) / 64;
    const bool bounded = cfg.num_refresh_vars >= 0 && cfg.num_refresh_vars < C::NB * C::Z;
    Cfg<T> g;
    g.rule = cfg.rule; g.max_iters = cfg.max_iters; g.early_exit = cfg.early_exit;
    g.float_syndrome0 = 0; g.num_active_checks = -1; g.num_refresh_vars = bounded ? cfg.num_refresh_vars : -1;
    g.scale = (T)cfg.scale; g.clip_tanh_arg = (T)0; g.clip_prod = (T)0; g.clip_post = (T)0; g.offset = (T)cfg.offset;
    if (sp) {  // fused sweep: float LLRs drawn in the kernel, no loop bound; the slots start and end together
        if constexpr (sizeof(TL) == 4 && (C::NB * C::Z) % 4 == 0) {
            if (bounded) { *supported = false; return hipSuccess; }
            auto kf = spec_r1w_kernel<C, T, TL, F, W, kOms, false, true>;
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
            *name = kOms ? (sizeof(T) == 8 ? "fused_spec_omsw_f64" : "fused_spec_omsw_f32") : (sizeof(T) == 8 ? "fused_spec_r1w_f64" : "fused_spec_r1w_f32");
            return hipGetLastError();
        } else {
            *supported = false;
            return hipSuccess;
        }
    }
    auto k = spec_r1w_kernel<C, T, TL, F, W, kOms>;
    if constexpr (F > 1) {  // early exit with several slots: every slot runs its own queue of frames
        if (cfg.early_exit && cfg.max_iters >= 1) k = spec_r1w_slots_kernel<C, T, TL, F, W, kOms>;
    }
    if (bounded) {
        if constexpr (r1w_refresh_built<C>()) {
            k = spec_r1w_kernel<C, T, TL, F, W, kOms, true>;
            if constexpr (F > 1) {
                if (cfg.early_exit && cfg.max_iters >= 1) k = spec_r1w_slots_kernel<C, T, TL, F, W, kOms, true>;
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
    *name = kOms ? (sizeof(T) == 8 ? "spec_omsw_f64" : "spec_omsw_f32") : (sizeof(T) == 8 ? "spec_r1w_f64" : "spec_r1w_f32");
    return hipGetLastError();
}

// Which codes and arithmetic types take this kernel instead of the single-wave one: those whose
// single-wave footprint leaves fewer than eight workgroups per compute unit, or whose base graph
// is too large for it altogether; the multi-wave layout must itself fit LDS.
template <class C, typename T, int F>
constexpr bool r1w_wanted()
{
    constexpr int single = F * (C::EB * C::Z * (int)sizeof(T) + C::MB * C::Z * 2 * (int)sizeof(T));
    return C::Z <= 64 && F * C::Z <= 128 && R1wLayout<C, T, F>::LDS_BYTES <= 160 * 1024 && C::EB <= 200 &&
           (C::EB > 100 || (160 * 1024) / single < 8);
}

// Four waves per workgroup; eight when LDS holds a single workgroup per compute unit anyway.
template <class C, typename T, int F>
constexpr int r1w_waves() { return R1wLayout<C, T, F>::LDS_BYTES > 80 * 1024 ? 8 : 4; }

template <class C, int SLOTS>
static hipError_t launch_r1w_code(const ldpc_decode_config &cfg, int compute_dtype, const Launch &io, int num_cus,
                                  const char **name, bool *supported, const SweepDev *sp = nullptr, Scratch *ws = nullptr)
{
    *supported = true;
    constexpr int FD = packed_slots<C, double, SLOTS, R1wLayout>(), FF = packed_slots<C, float, SLOTS, R1wLayout>();
    if constexpr (r1w_wanted<C, double, FD>()) {
        constexpr int W = r1w_waves<C, double, FD>();
        if (compute_dtype == LDPC_F64) {
            if (!sp && io.llr_dtype == LDPC_F64) return launch_r1w<C, double, double, FD, W>(cfg, io, num_cus, name, supported);
            return launch_r1w<C, double, float, FD, W>(cfg, io, num_cus, name, supported, sp, ws);
        }
    }
    if constexpr (r1w_wanted<C, float, FF>()) {
        constexpr int W = r1w_waves<C, float, FF>();
        if (compute_dtype == LDPC_F32 && (sp || io.llr_dtype == LDPC_F32))
            return launch_r1w<C, float, float, FF, W>(cfg, io, num_cus, name, supported, sp, ws);
    }
    *supported = false;
    return hipSuccess;
}

template <class C, typename T, typename TL, int F, int W>
static hipError_t launch_r0w(const ldpc_decode_config &cfg, const Launch &io, int num_cus, const char **name, bool *supported,
                             const SweepDev *sp = nullptr, Scratch *ws = nullptr)
{
    typedef R0wLayout<C, T, F> LY;
    constexpr int G = (LY::LANES + 63) / 64;
    Cfg<T> g;
    g.rule = 0; g.max_iters = cfg.max_iters; g.early_exit = cfg.early_exit;
    g.float_syndrome0 = cfg.float_syndrome0; g.num_active_checks = -1; g.num_refresh_vars = -1;
    g.scale = (T)cfg.scale; g.clip_tanh_arg = (T)0; g.clip_prod = (T)0; g.clip_post = (T)0; g.offset = (T)0;
    if (sp) {  // fused sweep
        if constexpr (sizeof(TL) == 4 && (C::NB * C::Z) % 4 == 0) {
            auto kf = spec_r0w_kernel<C, T, TL, F, W, true>;
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
            *name = sizeof(T) == 8 ? "fused_spec_r0w_f64" : "fused_spec_r0w_f32";
            return hipGetLastError();
        } else {
            *supported = false;
            return hipSuccess;
        }
    }
    auto k = spec_r0w_kernel<C, T, TL, F, W>;
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
    *name = sizeof(T) == 8 ? "spec_r0w_f64" : "spec_r0w_f32";
    return hipGetLastError();
}

// Rule R0 comes here for base graphs the single-wave kernel does not unroll (more than 100 edges)
// and for codes whose single-wave footprint leaves fewer than eight workgroups per compute unit.
template <class C, typename T, int F>
constexpr bool r0w_wanted()
{
    constexpr int single = F * (C::NB * C::Z * (int)sizeof(T) + C::MB * C::Z * 2 * (int)sizeof(T));
    return C::Z <= 64 && F * C::Z <= 128 && R0wLayout<C, T, F>::LDS_BYTES <= 160 * 1024 && C::EB <= 200 &&
           (C::EB > 100 || (160 * 1024) / single < 8);
}

// Waves per frame group where LDS holds a single workgroup per compute unit anyway (0: four everywhere):
// sixteen waves instead of eight hide the barriers of the BG2-shaped matrix in double (6.0 -> 7.1 Gbit/s)
#ifndef LDPC_R0W_WAVES_BIG
#define LDPC_R0W_WAVES_BIG 8
#endif

template <class C, int SLOTS>
static hipError_t launch_r0w_code(const ldpc_decode_config &cfg, int compute_dtype, const Launch &io, int num_cus,
                                  const char **name, bool *supported, const SweepDev *sp = nullptr, Scratch *ws = nullptr)
{
    *supported = true;
    constexpr int FD = packed_slots<C, double, SLOTS, R0wLayout>(), FF = packed_slots<C, float, SLOTS, R0wLayout>();
    if constexpr (r0w_wanted<C, double, FD>()) {
        constexpr int W = LDPC_R0W_WAVES_BIG > 0 && R0wLayout<C, double, FD>::LDS_BYTES > 80 * 1024 ? LDPC_R0W_WAVES_BIG : 4;
        if (compute_dtype == LDPC_F64) {
            if (!sp && io.llr_dtype == LDPC_F64) return launch_r0w<C, double, double, FD, W>(cfg, io, num_cus, name, supported);
            return launch_r0w<C, double, float, FD, W>(cfg, io, num_cus, name, supported, sp, ws);
        }
    }
    if constexpr (r0w_wanted<C, float, FF>()) {
        constexpr int W = LDPC_R0W_WAVES_BIG > 0 && R0wLayout<C, float, FF>::LDS_BYTES > 80 * 1024 ? LDPC_R0W_WAVES_BIG : 4;
        if (compute_dtype == LDPC_F32 && (sp || io.llr_dtype == LDPC_F32))
            return launch_r0w<C, float, float, FF, W>(cfg, io, num_cus, name, supported, sp, ws);
    }
    *supported = false;
    return hipSuccess;
}

// decode_spec_r1m.hip: rules R1 and OMS in double with one number per edge, for codes whose frame fits LDS
// twice that way and only once in the layout of this file; asked first
hipError_t launch_spec_r1m(int spec_id, const ldpc_decode_config &cfg, const Launch &io, int num_cus, const char **name,
                           bool *supported, const SweepDev *sp, Scratch *ws);

// One translation unit per rule (-DLDPC_R1W_RULE=0, 1 or 6) so that they compile side by side;
// templates the unit does not launch are never instantiated.
#if LDPC_R1W_RULE == 0
hipError_t launch_spec_r0w(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                           int num_cus, const char **name, bool *supported)
#elif LDPC_R1W_RULE == 6
hipError_t launch_spec_omsw(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                            int num_cus, const char **name, bool *supported)
#else
hipError_t launch_spec_r1w(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                           int num_cus, const char **name, bool *supported)
#endif
{
#if LDPC_R1W_RULE != 0
    if (compute_dtype == LDPC_F64) {
        hipError_t e1 = launch_spec_r1m(spec_id, cfg, io, num_cus, name, supported, nullptr, nullptr);
        if (e1 != hipSuccess || *supported) return e1;
    }
#endif
    int id = 0;
#if LDPC_R1W_RULE == 0
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) return launch_r0w_code<CODE, SLOTS>(cfg, compute_dtype, io, num_cus, name, supported); \
    ++id;
#else
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) return launch_r1w_code<CODE, SLOTS>(cfg, compute_dtype, io, num_cus, name, supported); \
    ++id;
#endif
    LDPC_FOR_EACH_SPEC_CODE(LDPC_LAUNCH)
#undef LDPC_LAUNCH
    *supported = false;
    return hipSuccess;
}

// the same kernels as one launch of channel + decode + error accounting (ldpc_sweep_batch)
#if LDPC_R1W_RULE == 0
hipError_t launch_spec_r0w_fused(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io, const SweepDev &sp,
                                 int num_cus, Scratch &workspace, const char **name, bool *supported)
#elif LDPC_R1W_RULE == 6
hipError_t launch_spec_omsw_fused(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io, const SweepDev &sp,
                                  int num_cus, Scratch &workspace, const char **name, bool *supported)
#else
hipError_t launch_spec_r1w_fused(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io, const SweepDev &sp,
                                 int num_cus, Scratch &workspace, const char **name, bool *supported)
#endif
{
#if LDPC_R1W_RULE != 0
    if (compute_dtype == LDPC_F64) {
        hipError_t e1 = launch_spec_r1m(spec_id, cfg, io, num_cus, name, supported, &sp, &workspace);
        if (e1 != hipSuccess || *supported) return e1;
    }
#endif
    int id = 0;
#if LDPC_R1W_RULE == 0
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) return launch_r0w_code<CODE, SLOTS>(cfg, compute_dtype, io, num_cus, name, supported, &sp, &workspace); \
    ++id;
#else
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) return launch_r1w_code<CODE, SLOTS>(cfg, compute_dtype, io, num_cus, name, supported, &sp, &workspace); \
    ++id;
#endif
    LDPC_FOR_EACH_SPEC_CODE(LDPC_LAUNCH)
#undef LDPC_LAUNCH
    *supported = false;
    return hipSuccess;
}

}  // namespace ldpc
