// Rule R1 (bpsk.py:154-260, extrinsic normalised min-sum) and the offset rule in double precision for
// the compiled-in codes that need several waves per frame group: the BG2-shaped 5G matrix of the reference
// and its 1400-row cut (channel.py:35), and 802.11n (648,324).  In the layout of decode_spec_r1w.hip the
// full 5G matrix takes 63 KB of messages + 27 KB of check states per frame: one eight-wave workgroup per
// compute unit, half of its time spent waiting at the two barriers of an iteration; 802.11n gets two
// four-wave workgroups because of its registers.
//
// Here a frame keeps ONE number per edge, updated in place: the variable writes its message to the
// check, the check overwrites it with its answer (the same lane reads and writes an edge's word in each
// phase, so the phases need nothing but the barrier between them).  No check-state array, no copy of
// the variable's own messages in registers: 65 KB of LDS and at most 128 registers per lane, so that
// TWO workgroups share a compute unit and one computes while the other waits (802.11n, two frames side by
// side in a wave: 39 KB, four workgroups).  Arithmetic and order of operations are those of
// spec_r1w_kernel: results are identical bit for bit.  Frame slots of a workgroup start together; with
// early exit a slot that has converged sits out until the last one has (no per-slot frame queues here).
//
// Block columns are dealt to the waves by what they cost in the variable phase - a variable of degree
// d forms d sums of d - 1 messages in the reference's order (bpsk.py:244-247), d (d - 1) / 2 additions
// with the common prefix shared - not by their edge count: the two columns of degree 22 and 23 of this
// graph are a wave's whole share.  (Halving such a column between two waves, each forming every other
// message from a separate array of the checks' answers, evened the shares out and changed nothing, and
// neither did dealing the block rows on top of the waves' column shares: with two workgroups per compute
// unit the kernel runs at what LDS takes in - eight bytes written per edge and phase, the LDS arrays busy
// two thirds of the time.)
#include "spec_r1w_plan.h"
#include "sweep_fused.h"

namespace ldpc {

// What a block column of degree d costs a wave in the variable phase: loads, running sum, stores, and the
// tails of the d sums.
constexpr int r1m_col_cost(int d) { return d * (d - 1) / 2 + 3 * d + 4; }

// Block columns dealt to the waves by that cost: longest first, always to the least loaded wave.
template <int W, int N>
struct R1mAssign {
    int cnt[W];
    int col[W][N];
    int max_cnt;
};

template <int W, int N>
constexpr R1mAssign<W, N> r1m_balance_cols(const int *ptr)
{
    R1mAssign<W, N> a{};
    int load[W] = {};
    bool used[N] = {};
    for (int step = 0; step < N; ++step) {
        int best = -1, bw = -1;
        for (int i = 0; i < N; ++i) {
            const int wgt = r1m_col_cost(ptr[i + 1] - ptr[i]);
            if (!used[i] && wgt > bw) { best = i; bw = wgt; }
        }
        used[best] = true;
        int w = 0;
        for (int v = 1; v < W; ++v)
            if (load[v] < load[w]) w = v;
        load[w] += bw;
        a.col[w][a.cnt[w]++] = best;
    }
    for (int w = 0; w < W; ++w) {
        for (int i = 1; i < a.cnt[w]; ++i)  // ascending inside a wave
            for (int j = i; j > 0 && a.col[w][j - 1] > a.col[w][j]; --j) {
                const int t = a.col[w][j]; a.col[w][j] = a.col[w][j - 1]; a.col[w][j - 1] = t;
            }
        a.max_cnt = a.cnt[w] > a.max_cnt ? a.cnt[w] : a.max_cnt;
    }
    return a;
}

template <class C, int W>
struct R1mPlan {
    static constexpr R1mAssign<W, C::NB> COLS = r1m_balance_cols<W, C::NB>(C::col_ptr);
    static constexpr R1wAssign<W, C::MB> ROWS = r1w_balance<W, C::MB>(C::row_ptr);
    static constexpr int NBW = COLS.max_cnt;
    static constexpr int MBW = ROWS.max_cnt;
};

// F frame slots side by side in the lanes of a wave (F Z <= 64: two 802.11n frames; one frame of the 5G
// matrices), each with its own messages and hard decisions.
template <class C, typename T, int F>
struct R1mLayout {
    static constexpr int QV_BYTES = C::EB * C::Z * (int)sizeof(T);   // [EB][Z]: one number per edge, in variable order
    static constexpr int HB_BYTES = ((C::NB * C::Z + 15) / 16) * 16;  // [NB][Z]: hard decisions, one byte each
    static constexpr int HB_OFF = QV_BYTES;
    static constexpr int SLOT_BYTES = ((QV_BYTES + HB_BYTES + 127) / 128) * 128 + (F > 1 ? 32 : 0);
    static constexpr int FLAG_OFF = F * SLOT_BYTES;
    // flags: two rows of eight per-slot "the syndrome failed" (by iteration parity), one row "a bit error" (fused)
    static constexpr int LDS_BYTES = FLAG_OFF + 128;
};

#ifndef LDPC_R1M_WAVES
#define LDPC_R1M_WAVES 8
#endif

template <class C>
constexpr bool r1m_no_double_blocks()
{
    for (int k = 0; k < C::EB; ++k)
        if (C::col_pair[k] != 0) return false;
    return true;
}

// REF, FUSED: as in spec_r1w_kernel (the reference's refresh bound num_refresh_vars, channel.py:241;
// channel in front and error accounting behind inside the kernel).
template <class C, typename T, typename TL, int F, int W, bool OMS, bool REF, bool FUSED>
__global__ void __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(4, 4)))
spec_r1m_kernel(Cfg<T> g, const TL *__restrict__ llr, long long frames, uint8_t *__restrict__ hard_out,
                int *__restrict__ iters_out, uint8_t *__restrict__ conv_out, T *__restrict__ post_out, SweepDev sp,
                float *scratch)
{
    typedef R1mLayout<C, T, F> LY;
    typedef R1mPlan<C, W> PL;
    constexpr int Z = C::Z, NB = C::NB, ST = sizeof(T);
    constexpr int n = NB * Z, NBW = PL::NBW, MBW = PL::MBW;
    static_assert(F * Z <= 64 && F <= 8, "a wave covers all frame slots; the per-slot flags are rows of eight");
    static_assert(r1m_no_double_blocks<C>(), "blocks that are the sum of two circulants need the ordering of spec_r1w_kernel");
    extern __shared__ __align__(16) unsigned char smem[];

    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const bool lane_on = lane < F * Z;  // idle lanes sit out every phase; they only meet the others at the barriers
    const int f = lane_on ? lane / Z : 0;
    const int z = lane_on ? lane - f * Z : 0;
    unsigned char *qv_b = smem + f * LY::SLOT_BYTES;
    unsigned char *hb_b = qv_b + LY::HB_OFF;
    int *flags = reinterpret_cast<int *>(smem + LY::FLAG_OFF);
    const unsigned zb = (unsigned)z * ST, z1 = (unsigned)z;
    SweepAcc acc;

    const long long groups = (frames + F - 1) / F;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long frame = grp * F + f;
        const bool active = lane_on && frame < frames;
        const float *mine = nullptr;
        if constexpr (FUSED) mine = sweep_stage_group<n, F>(sp, scratch, grp, frames);
        T L[NBW], own[NBW];
        // ---- start: every edge carries the channel LLR (bpsk.py:170-177)
        static_for<0, W>([&](auto w) {
            if (wave == w && lane_on) {
                static_for<0, NBW>([&](auto j) {
                    if constexpr (j < PL::COLS.cnt[w]) {
                        constexpr int bc = PL::COLS.col[w][j];
                        constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0;
                        T l;
                        if constexpr (FUSED) l = (T)mine[f * n + bc * Z + z] + (T)0;
                        else l = ((F == 1 || active) ? (T)llr[frame * n + bc * Z + z] : (T)0) + (T)0;  // -0.0 -> +0.0: signs are sign bits from here on
                        L[j] = l;
                        own[j] = l;
                        hb_b[bc * Z + z1] = (uint8_t)hard_bit<1>(l);
                        static_for<0, D>([&](auto k) { *reinterpret_cast<T *>(qv_b + (k0 + k) * Z * ST + zb) = l; });
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
            // ---- check nodes of this wave's block rows (bpsk.py:213-233): the messages of a check's
            //      variables in, its answers out over the same words; and the parity of the current
            //      hard decisions
            unsigned bad = 0;
            static_for<0, W>([&](auto w) {
                if (wave == w && (F == 1 ? lane_on : !done)) {  // one slot: the loop ends for all lanes when it is done
                    static_for<0, MBW>([&](auto i) {
                        if constexpr (i < PL::ROWS.cnt[w]) {
                            constexpr int br = PL::ROWS.item[w][i];
                            constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0;
                            T v[D > 0 ? D : 1];
                            unsigned off[D > 0 ? D : 1];
                            T m1 = Lim<T>::inf(), m2 = Lim<T>::inf();
                            unsigned sx = 0, par = 0;
                            static_for<0, D>([&](auto p) {
                                constexpr int kpos = C::row_kpos[e0 + p], s = C::row_shift[e0 + p], bc = C::row_col[e0 + p];
                                off[p] = kpos * Z * ST + rot_fwd<Z, ST, s>(zb);
                                v[p] = *reinterpret_cast<const T *>(qv_b + off[p]);
                                if (want_syn) par ^= hb_b[bc * Z + rot_fwd<Z, 1, s>(z1)];
                            });
                            static_for<0, D>([&](auto p) {
                                SNum<T>::template book<p>(m1, m2, v[p]);
                                sx ^= SNum<T>::sign_word(v[p]);
                            });
                            if constexpr (D < 2) { m1 = (T)0; m2 = (T)0; }  // bpsk.py:216
                            if (!last) {
                                // the two magnitudes a check can answer with, formed once (the same products
                                // as scaling the selected minimum edge by edge)
                                const T a1 = OMS ? oms_magnitude(m1, g.scale, g.offset) : g.scale * m1;
                                const T a2 = OMS ? oms_magnitude(m2, g.scale, g.offset) : g.scale * m2;
                                static_for<0, D>([&](auto p) {
                                    const T mag = t_abs(v[p]) == m1 ? a2 : a1;
                                    *reinterpret_cast<T *>(qv_b + off[p]) =
                                        SNum<T>::flip_sign(mag, (sx ^ SNum<T>::sign_word(v[p])) & 0x80000000u);
                                });
                            }
                            bad |= par;
                        }
                    });
                }
            });
            int *slot_bad = flags + (t & 1) * 8;
            if (bad) slot_bad[f] = 1;
            __syncthreads();
            // the other row was last read one iteration ago; the next iteration's writes to it come
            // after the barrier that closes this iteration
            if (threadIdx.x < F) flags[((t + 1) & 1) * 8 + threadIdx.x] = 0;
            if constexpr (F == 1) {
                if (want_syn) {
                    const bool ok = slot_bad[0] == 0;
                    if (last) { res_conv = ok ? 1 : 0; break; }
                    if (ok) { res_iters = t; res_conv = 1; break; }  // bpsk.py:205-209; the same in every lane
                }
            } else if (want_syn) {
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
            static_for<0, W>([&](auto w) {
                if (wave == w && (F == 1 ? lane_on : !done)) {
                    static_for<0, NBW>([&](auto j) {
                        if constexpr (j < PL::COLS.cnt[w]) {
                            constexpr int bc = PL::COLS.col[w][j];
                            constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0;
                            T r[D > 0 ? D : 1];
                            static_for<0, D>([&](auto k) { r[k] = *reinterpret_cast<const T *>(qv_b + (k0 + k) * Z * ST + zb); });
                            bool refresh = true;
                            if constexpr (REF) refresh = bc * Z + z < g.num_refresh_vars;
                            T run = (T)0;
                            static_for<0, D>([&](auto k) {
                                T ext = run;
                                static_for<k + 1, D>([&](auto kk) { ext += r[kk]; });
                                const T qn = L[j] + ext;
                                *reinterpret_cast<T *>(qv_b + (k0 + k) * Z * ST + zb) = refresh ? qn : L[j];
                                run += r[k];
                            });
                            own[j] = L[j] + run;  // bpsk.py:237
                            hb_b[bc * Z + z1] = (uint8_t)hard_bit<1>(own[j]);
                        }
                    });
                }
            });
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
                                constexpr int bc = PL::COLS.col[w][j];
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
                            constexpr int bc = PL::COLS.col[w][j];
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

// ------------------------------------------------------------------------------------------ host
// Served: compiled-in codes without double-circulant blocks that the multi-wave kernels of
// decode_spec_r1w.hip would serve in double (base graphs of 100 to 200 edges, or a single-wave footprint of
// fewer than eight workgroups per compute unit) and whose workgroup takes at most half of LDS in this
// layout.  In the layout of spec_r1w_kernel the full 5G matrix takes more than half and its 1400-row cut
// 77 KB (two workgroups of four waves against two of eight here, 6.3 -> 7.9 Gbit/s); 802.11n fits three
// times but its registers allow two workgroups of four waves, here four.
template <class C, int SLOTS>
constexpr int r1m_slots() { return SLOTS * C::Z <= 64 && SLOTS <= 8 ? SLOTS : 1; }
template <class C>
constexpr int r1m_waves() { return C::EB > 100 ? LDPC_R1M_WAVES : 4; }

template <class C, int SLOTS>
constexpr bool r1m_wanted()
{
    constexpr int F = r1m_slots<C, SLOTS>();
    constexpr int single = F * (C::EB * C::Z * 8 + C::MB * C::Z * 16);
    return C::Z <= 64 && C::EB <= 200 && (C::EB > 100 || (160 * 1024) / single < 8) &&
           R1mLayout<C, double, F>::LDS_BYTES <= 80 * 1024 && r1m_no_double_blocks<C>();
}

template <class C, typename TL, int F, bool OMS>
static hipError_t launch_r1m(const ldpc_decode_config &cfg, const Launch &io, int num_cus, const char **name, bool *supported,
                             const SweepDev *sp, Scratch *ws)
{
    typedef double T;
    constexpr int W = r1m_waves<C>();
    typedef R1mLayout<C, T, F> LY;
    const bool bounded = cfg.num_refresh_vars >= 0 && cfg.num_refresh_vars < C::NB * C::Z;
    Cfg<T> g;
    g.rule = cfg.rule; g.max_iters = cfg.max_iters; g.early_exit = cfg.early_exit;
    g.float_syndrome0 = 0; g.num_active_checks = -1; g.num_refresh_vars = bounded ? cfg.num_refresh_vars : -1;
    g.scale = (T)cfg.scale; g.clip_tanh_arg = (T)0; g.clip_prod = (T)0; g.clip_post = (T)0; g.offset = (T)cfg.offset;
    *supported = true;
    auto run = [&](auto k, bool fused) -> hipError_t {
        int per_cu = 0;
        hipError_t e = kernel_setup(reinterpret_cast<const void *>(k), 64 * W, LY::LDS_BYTES, &per_cu);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        const long long groups = (io.frames + F - 1) / F, full = (long long)num_cus * per_cu * 2;
        long long grid = full > groups ? groups : full;
        if (grid < 1) grid = 1;
        float *scratch = nullptr;
        if (fused) {
            e = scratch_reserve(*ws, sweep_slice_bytes<C::NB * C::Z, F>(), full, grid);
            if (e != hipSuccess) return e;
            scratch = reinterpret_cast<float *>(ws->ptr);
        }
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(64 * W), LY::LDS_BYTES, io.stream, g,
                           reinterpret_cast<const TL *>(io.llr), io.frames, io.hard, io.iters, io.converged,
                           reinterpret_cast<T *>(io.posterior), sp ? *sp : SweepDev{}, scratch);
        return hipGetLastError();
    };
    if (sp) {  // fused sweep: float LLRs drawn in the kernel, no loop bound
        if constexpr (sizeof(TL) == 4 && (C::NB * C::Z) % 4 == 0) {
            if (bounded) { *supported = false; return hipSuccess; }
            *name = OMS ? "fused_spec_omsm_f64" : "fused_spec_r1m_f64";
            return run(spec_r1m_kernel<C, T, TL, F, W, OMS, false, true>, true);
        } else {
            *supported = false;
            return hipSuccess;
        }
    }
    *name = OMS ? "spec_omsm_f64" : "spec_r1m_f64";
    if (bounded) {
        if constexpr (!OMS && C::EB > 100) return run(spec_r1m_kernel<C, T, TL, F, W, OMS, true, false>, false);
        *supported = false;
        return hipSuccess;
    }
    return run(spec_r1m_kernel<C, T, TL, F, W, OMS, false, false>, false);
}

// Rules R1 and OMS in double on the codes above; *supported = false sends the caller on to the kernels of
// decode_spec_r1w.hip.  sp != NULL: the fused sweep (io.llr and the result arrays unused).
hipError_t launch_spec_r1m(int spec_id, const ldpc_decode_config &cfg, const Launch &io, int num_cus, const char **name,
                           bool *supported, const SweepDev *sp, Scratch *ws)
{
    *supported = false;
    if (cfg.rule != 1 && cfg.rule != LDPC_RULE_OMS) return hipSuccess;
    int id = 0;
#define LDPC_LAUNCH(CODE, SLOTS)                                                                              \
    if (spec_id == id) {                                                                                      \
        if constexpr (r1m_wanted<CODE, SLOTS>()) {                                                            \
            constexpr int FS = r1m_slots<CODE, SLOTS>();                                                      \
            if (cfg.rule == 1) {                                                                              \
                if (!sp && io.llr_dtype == LDPC_F64) return launch_r1m<CODE, double, FS, false>(cfg, io, num_cus, name, supported, sp, ws); \
                return launch_r1m<CODE, float, FS, false>(cfg, io, num_cus, name, supported, sp, ws);         \
            }                                                                                                 \
            if (!sp && io.llr_dtype == LDPC_F64) return launch_r1m<CODE, double, FS, true>(cfg, io, num_cus, name, supported, sp, ws); \
            return launch_r1m<CODE, float, FS, true>(cfg, io, num_cus, name, supported, sp, ws);              \
        }                                                                                                     \
        return hipSuccess;                                                                                    \
    }                                                                                                         \
    ++id;
    LDPC_FOR_EACH_SPEC_CODE(LDPC_LAUNCH)
#undef LDPC_LAUNCH
    return hipSuccess;
}

}  // namespace ldpc
