// Row-layered normalised min-sum (build extension, see LDPC_RULE_LNMS in ldpc_amd.h) for the 5G NR
// base graphs at ANY lifting size, with ANY shift table and on ANY prefix of the block rows.
//
// decode_lnms.hip compiles a whole matrix into its kernel: connectivity, lifting size and every
// shift are constant expressions.  That serves the packaged tables only.  What a 5G link actually
// decodes changes from transport block to transport block: 51 lifting sizes, eight shift tables per
// base graph, and rate matching cuts the matrix to the block rows whose parity bits were sent (what
// /root/reference/channel.py:35 fakes with H[:1400, :] and /root/reference/bpsk.py:163 with a
// hard-coded prefix).  Only the CONNECTIVITY of the two base graphs never changes, so this kernel
// is compiled per base graph (block rows and their block columns unrolled, state registers
// assigned per block row, barriers placed at compile time) and takes everything else at run time:
//   * Z, the frame slots per workgroup and the LDS geometry as kernel arguments (scalar registers);
//   * per base edge two words {Z - s, byte offset of element s of its column}, read by scalar loads
//     one block row at a time.  The rotated address of an edge is then
//         v_cmp_le_u32 (z against Z - s)  v_cndmask (wrapped / unwrapped lane base)  v_add (offset)
//     i.e. ONE vector instruction more than with the shift as an immediate (15 instead of 14 per
//     edge and sweep);
//   * the number of active block rows: the rows beyond it are skipped by a scalar branch each.
// Arithmetic, schedule and results are those of decode_lnms.hip and of the CPU restatement
// (oracle/bp_oracle.c: decode_lnms_*), bit for bit.
//
// Thread (f, z) owns check (br, z) of every block row of frame slot f; F = MAXT / Z slots (at most
// 16) share a workgroup.  LDS: [state of the block rows beyond the RL kept in registers: A, B, W per
// thread][posteriors of the columns that sit in two or more checks: slot, column, Z][flags].
// Variables of degree one (the extension parity bits) are never stored: their channel LLR is
// re-read one layer ahead, their hard decision rides in bit 30 of the check's state word.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "spec_common.h"
#include "lnms_plan.h"

namespace ldpc {

typedef const unsigned __attribute__((address_space(4))) *bg_u32;

struct BgGeom {
    int Z, F, FZ;         // lifting size, frame slots per workgroup, F * Z
    int mb_act;           // active block rows (a prefix)
    int npl_act;          // ... of which have a variable of degree one
    int nb_idle0, nb_idle1;  // block columns [nb_idle0, nb_idle1) sit in no active check: the channel decides
    int n;                // row length of the LLR and result arrays
    unsigned vec;         // Z * sizeof(T)
    unsigned slot_bytes;  // posteriors of one slot: shared columns * vec
    const unsigned *tab;  // [2 EB] {Z - s, offset} per base edge by block row, then [2 NPL] the same
                          // for the degree-one edge of the j-th block row that has one
};

// smallest prefix of block rows in which every column that is kept in LDS sits in two checks (so
// that "degree one" means the same in the prefix as in the whole graph)
template <class C>
constexpr int bg_min_prefix()
{
    int need = 0;
    for (int bc = 0; bc < C::NB; ++bc) {
        if (!LnPlan<C>::shared(bc)) continue;
        const int second = C::col_row[C::col_ptr[bc] + 1] + 1;  // rows needed to see two of its checks
        if (second > need) need = second;
    }
    return need;
}

// rows the kernel can serve without asking: every base graph row is a layer of its own
template <class C>
constexpr bool bg_graph_ok()
{
    if (!LnPlan<C>::supported()) return false;
    for (int br = 0; br < C::MB; ++br)
        if (LnPlan<C>::rdeg(br) < 2) return false;
    // every prefix of the block rows must use a prefix of the block columns (the extension parity columns
    // follow their rows): "the columns beyond the last one used" are then the ones in no active check
    int top = -1;
    bool seen[C::NB] = {};
    for (int br = 0; br < C::MB; ++br) {
        for (int e = C::row_ptr[br]; e < C::row_ptr[br + 1]; ++e) {
            seen[C::row_col[e]] = true;
            if (C::row_col[e] > top) top = C::row_col[e];
        }
        if (br + 1 >= bg_min_prefix<C>())
            for (int bc = 0; bc <= top; ++bc)
                if (!seen[bc]) return false;
    }
    return LnPlan<C>::NS > 0;
}
// byte address of element ((z + s) mod Z): b = base + z * size, bW = b - Z * size, thr = Z - s,
// off = (column start + s) * size
__device__ __forceinline__ unsigned bg_rot(unsigned z, unsigned b, unsigned bW, unsigned thr, unsigned off)
{
    unsigned a;
    asm("v_cmp_le_u32_e32 vcc, %3, %4\n\tv_cndmask_b32_e32 %0, %1, %2, vcc\n\tv_add_u32_e32 %0, %5, %0"
        : "=&v"(a) : "v"(b), "v"(bW), "s"(thr), "v"(z), "s"(off) : "vcc");
    return a;
}

template <class C, typename T, int MAXT, int RL>
struct BgLayout {
    typedef LnPlan<C> PL;
    static constexpr int ST = sizeof(T);
    static constexpr int SPILLED = C::MB > RL ? C::MB - RL : 0;
    static constexpr unsigned STATE_ROW = MAXT * (2 * ST + 4);  // A[MAXT], B[MAXT], W[MAXT] of one block row
    static constexpr unsigned P_BASE = SPILLED * STATE_ROW;
    static constexpr unsigned FLAG_OFF = P_BASE + PL::NS * MAXT * ST;
    static constexpr unsigned LDS_BYTES = FLAG_OFF + 128;
    static constexpr unsigned a_off(int k) { return k * STATE_ROW; }
    static constexpr unsigned b_off(int k) { return k * STATE_ROW + MAXT * ST; }
    static constexpr unsigned w_off(int k) { return k * STATE_ROW + 2 * MAXT * ST; }
};

template <class C, typename T, typename TL, int MAXT, int RL, int WPE>
__global__ void __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(WPE, 8)))
bg_lnms_kernel(Cfg<T> g, BgGeom q, const TL *__restrict__ llr, long long frames, uint8_t *__restrict__ hard_out,
               int *__restrict__ iters_out, uint8_t *__restrict__ conv_out, T *__restrict__ post_out)
{
    typedef LnPlan<C> PL;
    typedef BgLayout<C, T, MAXT, RL> LY;
    static_assert(bg_graph_ok<C>(), "base graph not supported by the layered kernel");
    constexpr int NB = C::NB, MB = C::MB, ST = sizeof(T), TLS = sizeof(TL);
    constexpr int MINP = bg_min_prefix<C>();
    constexpr int NPL = PL::n_priv_layers();
    extern __shared__ __align__(16) unsigned char smem[];

    // the table pointer is re-read through an empty asm wherever a block row's entries are fetched: the
    // entries are loop invariants, and hoisted out of the sweep all 2 EB of them would be spilled
    unsigned long long tab_bits = (unsigned long long)q.tab;
    auto table = [&]() -> bg_u32 {
        asm volatile("" : "+s"(tab_bits));
        return (bg_u32)tab_bits;
    };
    const int Z = q.Z, n = q.n;
    // number of active block rows / of those with a private variable, re-read through an empty asm where
    // a block row is gated: the 46 comparisons are loop invariants too, and the compiler would keep
    // their outcomes in (spilled) scalar register pairs
    int mba_bits = q.mb_act, npl_bits = q.npl_act;
    auto row_on = [&](int br) -> bool {
        asm volatile("" : "+s"(mba_bits));
        return br < mba_bits;
    };
    const int tid = threadIdx.x;
    const bool lane_on = tid < q.FZ;
    int f = 0, z = 0;
    if (lane_on) { f = tid / Z; z = tid - f * Z; }
    const unsigned zu = (unsigned)z;
    const unsigned pz = LY::P_BASE + (unsigned)f * q.slot_bytes + zu * ST;  // absolute LDS addresses from here on
    const unsigned pzW = pz - q.vec;
    const unsigned stA = (unsigned)tid * ST, stW = (unsigned)tid * 4;       // spilled check state of this thread
    int *slot_bad = reinterpret_cast<int *>(smem + LY::FLAG_OFF);           // [0, 16): per slot; [16]: any slot
    volatile int *slot_bad_now = slot_bad;  // polled while other waves may still be writing
    // lanes of this wave that work on the same slot as this lane
    unsigned long long wave_slot_mask = 0ull;
    for (int ff = 0; ff < q.F; ++ff) {
        const unsigned long long m = __ballot(lane_on && f == ff);
        if (f == ff) wave_slot_mask = m;
    }

    const long long groups = (frames + q.F - 1) / q.F;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long frame = grp * q.F + f;
        const bool active = lane_on && frame < frames;
        // loads never depend on `active`: idle lanes read a frame that exists.  The group's first
        // frame is the same for every lane (a scalar base address), the rest is a 32-bit offset.
        const TL *lf = llr + grp * q.F * n;
        const unsigned fo = (frame < frames ? (unsigned)f : 0u) * (unsigned)n + zu;
        const unsigned hz = fo * TLS, hzW = hz - (unsigned)Z * TLS;

        T A[RL > 0 ? RL : 1], B[RL > 0 ? RL : 1];
        unsigned W[RL > 0 ? RL : 1];
        static_for<0, MB>([&](auto br) {
            if constexpr (br < RL) { A[br] = (T)0; B[br] = (T)0; W[br] = 0u; }
            else if (lane_on) {
                lds_put<T>(stA + LY::a_off(br - RL), (T)0);
                lds_put<T>(stA + LY::b_off(br - RL), (T)0);
                lds_put<unsigned>(stW + LY::w_off(br - RL), 0u);
            }
        });
        static_for<0, NB>([&](auto bc) {
            if constexpr (PL::shared(bc)) {
                const T l = (T)lf[fo + (unsigned)(bc * Z)] + (T)0;  // -0.0 -> +0.0: "negative" is "sign bit set" from here on
                if (lane_on) lds_put<T>(pz + (unsigned)PL::lds_index(bc) * q.vec, l);
            }
        });
        if (tid <= 16) slot_bad[tid] = 0;
        __syncthreads();

        auto load_state = [&](auto br, T &a, T &b, unsigned &w) {
            if constexpr (br < RL) { a = A[br]; b = B[br]; w = W[br]; }
            else {
                a = lds_get<T>(stA + LY::a_off(br - RL));
                b = lds_get<T>(stA + LY::b_off(br - RL));
                w = lds_get<unsigned>(stW + LY::w_off(br - RL));
            }
        };
        // channel LLR of the variable of degree one behind table entry k (two words)
        auto load_priv_at = [&](unsigned thr, unsigned off) -> T {
            unsigned zz = zu;
            asm volatile("" : "+v"(zz));  // one address per use, computed where it is used
            const unsigned o = bg_rot(zz, hz, hzW, thr, off);
            return (T)*reinterpret_cast<const TL *>(reinterpret_cast<const char *>(lf) + o) + (T)0;
        };
        auto load_priv = [&](auto br, bg_u32 tab) -> T {
            constexpr int e = C::row_ptr[br] + PL::priv_pos(br);
            return load_priv_at(tab[2 * e], tab[2 * e + 1]);
        };
        // index inside the frame of that variable
        auto priv_index = [&](auto br, bg_u32 tab) -> unsigned {
            constexpr int e = C::row_ptr[br] + PL::priv_pos(br);
            unsigned zz = zu;
            asm volatile("" : "+v"(zz));
            return bg_rot(zz, zu * TLS, (zu - (unsigned)Z) * TLS, tab[2 * e], tab[2 * e + 1]) / TLS;
        };
        // Bit 30 of W holds the hard decision of the layer's private variable whenever a syndrome
        // test may follow (early exit: always; else after the last sweep), so that the test needs
        // no channel LLRs.  Before the first sweep that decision is the channel's.
        if (g.early_exit) {
            static_for<0, MB>([&](auto br) {
                if constexpr (PL::is_priv_layer(br)) {
                    if (br < MINP || row_on(br)) {
                        const unsigned hb = hard_bit<1>(load_priv(br, table())) << 30;
                        if constexpr (br < RL) W[br] = hb;
                        else if (lane_on) lds_put<unsigned>(stW + LY::w_off(br - RL), hb);
                    }
                }
            });
        }
        // posterior L + r of that variable, r rebuilt from the check's state
        auto priv_value = [&](auto br, T l) -> T {
            constexpr int D = PL::rdeg(br), p = PL::priv_pos(br);
            T a, b;
            unsigned w;
            load_state(br, a, b, w);
            const T mag = (int)((w >> 24) & 31u) == p ? b : a;
            return l + SNum<T>::flip_sign(mag, ((w << (31 - (D - 1 - p))) ^ w) & 0x80000000u);
        };
        auto write_out = [&](int iters, int conv, bg_u32 tab) {
            uint8_t *ho = hard_out + frame * n;
            T *po = post_out ? post_out + frame * n : nullptr;
            static_for<0, NB>([&](auto bc) {
                if constexpr (PL::shared(bc)) {
                    const T v = lds_get<T>(pz + (unsigned)PL::lds_index(bc) * q.vec);
                    ho[bc * Z + z] = (uint8_t)hard_bit<1>(v);
                    if (po) po[bc * Z + z] = v;
                }
            });
            static_for<0, MB>([&](auto br) {
                if constexpr (PL::priv_pos(br) >= 0) {
                    if (br < MINP || br < q.mb_act) {  // (lanes have parted ways here: no scalar tricks)
                        const unsigned i = priv_index(br, tab);
                        if (po) {  // posteriors wanted: L + r needs the channel LLR again
                            const T v = priv_value(br, load_priv(br, tab));
                            ho[i] = (uint8_t)hard_bit<1>(v);
                            po[i] = v;
                        } else {   // hard decisions only: they sit in bit 30 of the state words
                            T a, b;
                            unsigned w;
                            load_state(br, a, b, w);
                            ho[i] = (uint8_t)((w >> 30) & 1u);
                        }
                    }
                }
            });
            for (int bc = q.nb_idle0; bc < q.nb_idle1; ++bc) {  // in no active check: the channel decides
                const T v = (T)lf[fo + (unsigned)(bc * Z)] + (T)0;
                ho[bc * Z + z] = (uint8_t)hard_bit<1>(v);
                if (po) po[bc * Z + z] = v;
            }
            if (z == 0) { iters_out[frame] = iters; conv_out[frame] = (uint8_t)conv; }
        };

        // channel LLR of the private variable the sweep needs next, requested one layer ahead
        // (cyclically over the active layers that have one)
        constexpr int SYN_FIRST = 2, SYN_EVERY = 6;  // block rows between looks at the outcome of the syndrome test
        T lp_next = (T)0;
        if constexpr (NPL > 0) {
            const bg_u32 tab = table();
            lp_next = load_priv_at(tab[2 * C::EB], tab[2 * C::EB + 1]);
        }

        // A slot that has converged is left alone (its posteriors and check state stay as they were
        // at the successful test) and everything is written once, after the loop.
        bool done = !active;
        int res_iters = g.early_exit ? g.max_iters - 1 : g.max_iters, res_conv = 0;
        for (int t = 0;; ++t) {
            const bool last = t == g.max_iters;
            if (g.early_exit || last) {
                // ---- syndrome of the hard decisions `0 if P > 0 else 1` of the current posteriors
                __syncthreads();
                if (last && g.early_exit) break;  // no test after the last sweep
                // A failed test needs no more rows: every few block rows a wave stops if each of its
                // slots is known to have failed, from its own lanes or from the flag another wave set.
                unsigned bad = 0;
                bool skip = false;  // the same in all lanes of a wave
                static_for<0, MB>([&](auto br) {
                    if constexpr (br == SYN_FIRST || (br > SYN_FIRST && (br - SYN_FIRST) % SYN_EVERY == 0)) {
                        bool known = done || (__ballot(bad != 0) & wave_slot_mask) != 0ull;
                        if (bad) slot_bad_now[f] = 1;
                        known = known || slot_bad_now[f] != 0;
                        skip = __ballot(!known) == 0ull;
                    }
                    if (br >= MINP && !row_on(br)) return;
                    const bg_u32 tab = table();  // before the lanes part ways: a scalar value
                    if (skip || done) return;
                    constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0, PP = PL::priv_pos(br);
                    unsigned zr = zu;
                    asm volatile("" : "+v"(zr));  // keep the rotated addresses out of loop-invariant hoisting
                    unsigned te[2 * D];
                    static_for<0, 2 * D>([&](auto i) { te[i] = tab[2 * e0 + i]; });
                    __builtin_amdgcn_sched_barrier(0);  // the row's entries in one go, ahead of the first address
                    T v[D];
                    static_for<0, D>([&](auto p) {  // all reads of the row first, then their signs
                        if constexpr (p != PP) {
                            v[p] = lds_get<T>(bg_rot(zr, pz, pzW, te[2 * p], te[2 * p + 1]));
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                    unsigned par = 0;
                    static_for<0, D>([&](auto p) {
                        if constexpr (p != PP) par ^= hard_bit<1>(v[p]);
                        else {
                            T a, b;
                            unsigned w;
                            load_state(br, a, b, w);
                            par ^= (w >> 30) & 1u;
                        }
                    });
                    bad |= par;
                });
                if (bad) { slot_bad[f] = 1; slot_bad[16] = 1; }
                __syncthreads();
                const bool slot_ok = slot_bad[f] == 0;
                const bool any_bad = slot_bad[16] != 0;
                if (last) {
                    if (!done) res_conv = slot_ok ? 1 : 0;
                    break;
                }
                if (!done && slot_ok) { res_iters = t; res_conv = 1; done = true; }
                if (!any_bad) break;  // every slot that was still running has just converged
                __syncthreads();
                if (tid <= 16) slot_bad[tid] = 0;
            }
            // ---- one sweep over the active layers
            const bool want_bits = g.early_exit || t == g.max_iters - 1;
            static_for<0, MB>([&](auto br) {
                constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0, PP = PL::priv_pos(br);
                if (br >= MINP && !row_on(br)) return;  // scalar branch: the same for the whole workgroup
                if constexpr (PL::barrier_before(br)) __syncthreads();
                const bg_u32 tab = table();
                T lp = (T)0;
                if constexpr (PP >= 0) {
                    constexpr int j = PL::priv_layer_ord(br);
                    lp = lp_next;
                    asm volatile("" : "+s"(npl_bits));
                    const int jn = j + 1 >= npl_bits ? 0 : j + 1;
                    lp_next = load_priv_at(tab[2 * C::EB + 2 * jn], tab[2 * C::EB + 2 * jn + 1]);
                }
                if (!done) {
                    T qv[D];
                    unsigned off[D];
                    T Ao, Bo;
                    unsigned w;
                    load_state(br, Ao, Bo, w);
                    const int idx_old = (int)((w >> 24) & 31u);
                    const unsigned wp = w ^ (unsigned)((int)w >> 31);  // bit D-1-p: sign of the old message on edge p
                    T m1, m2;
                    unsigned zr = zu;
                    asm volatile("" : "+v"(zr));  // as above: hundreds of hoisted addresses would spill
                    unsigned te[2 * D];
                    static_for<0, 2 * D>([&](auto i) { te[i] = tab[2 * e0 + i]; });
                    __builtin_amdgcn_sched_barrier(0);  // the row's entries in one go, ahead of the first address
                    // all posteriors of the row are requested first, so that the row pays one LDS round
                    // trip instead of one per edge
                    static_for<0, D>([&](auto p) {
                        if constexpr (p == PP) {
                            off[p] = 0;
                        } else {
                            off[p] = bg_rot(zr, pz, pzW, te[2 * p], te[2 * p + 1]);
                            qv[p] = lds_get<T>(off[p]);
                            if constexpr (ST == 4) __builtin_amdgcn_sched_barrier(0);  // address, read, address, read
                        }
                    });
                    // bit 31 of ws: sign of the old message on the current edge; every edge shifts it out
                    // and the sign of its input in at the other end, so the word ends as the new signs
                    unsigned ws = wp << (32 - D);
                    static_for<0, D>([&](auto p) {
                        if constexpr (p == PP) {
                            qv[p] = lp;  // a variable of degree one always sends its channel LLR
                            SNum<T>::template edge_in_given<p == 0>(qv[p], ws, m1, m2, Lim<T>::inf());
                        } else {
                            SNum<T>::template edge_in<p, p == 0>(qv[p], (unsigned)idx_old, Ao, Bo, ws, m1, m2, Lim<T>::inf());
                        }
                    });
                    const unsigned sq = ws;
                    const unsigned px = (unsigned)__popc(sq) << 31;  // sign parity of the inputs
                    const T An = g.scale * m1, Bn = g.scale * m2;
                    const T Anp = SNum<T>::flip_sign(An, px), Bnp = SNum<T>::flip_sign(Bn, px);  // parity folded in
                    unsigned minbits = 0;  // bit p: edge p holds the smallest magnitude
                    unsigned privbit = 0;
                    static_for<0, D>([&](auto pp) {
                        constexpr int p = D - 1 - pp;
                        const T pn = SNum<T>::pick_add(qv[p], m1, Anp, Bnp, minbits);  // the variable's new posterior
                        if constexpr (p != PP) {
                            lds_put<T>(off[p], pn);
                            if constexpr (ST == 4) __builtin_amdgcn_sched_barrier(0);  // block, write, block, write
                        }
                        else if (want_bits) privbit = hard_bit<1>(pn) << 30;
                    });
                    const int idx = __builtin_ctz(minbits);  // the first minimum; some bit is always set
                    const unsigned wn = sq | ((unsigned)idx << 24) | px | privbit;
                    if constexpr (br < RL) { A[br] = An; B[br] = Bn; W[br] = wn; }
                    else {
                        lds_put<T>(stA + LY::a_off(br - RL), An);
                        lds_put<T>(stA + LY::b_off(br - RL), Bn);
                        lds_put<unsigned>(stW + LY::w_off(br - RL), wn);
                    }
                }
            });
        }
        const bg_u32 tab_out = table();
        if (active) write_out(res_iters, res_conv, tab_out);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ host
// Register budget as in decode_lnms.hip: the workgroup is planned as the only one on its compute
// unit, its waves per SIMD fix the registers, about 57 of them are left for everything that is not
// check state.
template <class C, typename T>
struct BgConfig {
    static constexpr int MAXT = sizeof(T) == 4 ? 768 : 384;
    static constexpr int RPL = 2 * (int)sizeof(T) / 4 + 1;
    static constexpr int WAVES = MAXT / 64;
    static constexpr int WPE = (WAVES + 3) / 4;
    static constexpr int VGPRS = WPE <= 2 ? 256 : (512 / WPE) / 8 * 8;
    static constexpr int STATE_REGS = VGPRS - 57 > 180 ? 180 : VGPRS - 57;
    static constexpr int RL = C::MB < STATE_REGS / RPL ? C::MB : STATE_REGS / RPL;
    static constexpr bool OK = BgLayout<C, T, MAXT, RL>::LDS_BYTES <= 160 * 1024;
};

// the base graphs this file is compiled for: connectivity of the packaged NR tables
#ifdef LDPC_BG_DEV_ONLY  // development aid: BG1 in float only
#define LDPC_FOR_EACH_BASE_GRAPH(X) X(0, Code_nr_bg1_z384)
constexpr bool kBgDouble = false;
#else
#define LDPC_FOR_EACH_BASE_GRAPH(X) X(0, Code_nr_bg1_z384) X(1, Code_nr_bg2_z40)
constexpr bool kBgDouble = true;
#endif
constexpr int kBaseGraphs = 2;

template <class C>
static bool bg_match_one(int Z, int mb, int nb, int Eb, const int *er, const int *ec)
{
    if constexpr (!bg_graph_ok<C>()) return false;
    else {
        if (mb < bg_min_prefix<C>() || mb > C::MB || nb > C::NB || Z < 2 || Z > 768) return false;
        if (Eb != C::row_ptr[mb]) return false;
        int top = 0;
        for (int e = 0; e < Eb; ++e) {
            // edges arrive sorted by (row, column): the same order as the generated tables
            if (ec[e] != C::row_col[e]) return false;
            if (e < C::row_ptr[er[e]] || e >= C::row_ptr[er[e] + 1]) return false;
            if (ec[e] > top) top = ec[e];
        }
        return nb > top;
    }
}

int bg_graph_match(int Z, int mb, int nb, int Eb, const int *er, const int *ec)
{
#define LDPC_MATCH(ID, CODE) if (bg_match_one<CODE>(Z, mb, nb, Eb, er, ec)) return ID;
    LDPC_FOR_EACH_BASE_GRAPH(LDPC_MATCH)
#undef LDPC_MATCH
    return -1;
}

template <class C>
static void bg_fill(int Z, int mb, const int *es, std::vector<unsigned> &host)
{
    typedef LnPlan<C> PL;
    constexpr int SLOT = 2 * C::EB + 2 * (PL::n_priv_layers() > 0 ? PL::n_priv_layers() : 1);
    host.assign((size_t)3 * SLOT, 0u);
    const int sizes[3][2] = {{4, 4}, {8, 4}, {8, 8}};  // (arithmetic, LLR) item sizes
    for (int v = 0; v < 3; ++v) {
        unsigned *t = host.data() + (size_t)v * SLOT;
        // entries of rows the handle does not have: {Z, 0} = "element z of the first column", an address
        // that exists, whatever reads through such an entry (the first private LLR is requested
        // before the sweep even when no active row has one)
        for (int i = 0; i < SLOT; i += 2) { t[i] = (unsigned)Z; t[i + 1] = 0u; }
        const unsigned st = sizes[v][0], tl = sizes[v][1];
        for (int br = 0; br < mb; ++br)
            for (int e = C::row_ptr[br]; e < C::row_ptr[br + 1]; ++e) {
                const int bc = C::row_col[e], s = es[e];
                t[2 * e] = (unsigned)(Z - s);
                t[2 * e + 1] = PL::shared(bc) ? ((unsigned)PL::lds_index(bc) * Z + s) * st : ((unsigned)bc * Z + s) * tl;
                if (PL::is_priv_layer(br) && e == C::row_ptr[br] + PL::priv_pos(br)) {
                    const int j = PL::priv_layer_ord(br);
                    t[2 * C::EB + 2 * j] = t[2 * e];
                    t[2 * C::EB + 2 * j + 1] = t[2 * e + 1];
                }
            }
    }
}

// Run-time tables of a handle whose description matched base graph `graph`: three variants
// (float / float LLRs, double / float LLRs, double / double LLRs) in one device allocation.
hipError_t bg_tables_create(int graph, int Z, int mb, const int *es, unsigned **d_tab)
{
    std::vector<unsigned> host;
#define LDPC_FILL(ID, CODE) if (graph == ID) { if constexpr (bg_graph_ok<CODE>()) bg_fill<CODE>(Z, mb, es, host); }
    LDPC_FOR_EACH_BASE_GRAPH(LDPC_FILL)
#undef LDPC_FILL
    if (host.empty()) return hipErrorInvalidValue;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(d_tab), host.size() * sizeof(unsigned));
    if (e != hipSuccess) return e;
    e = hipMemcpy(*d_tab, host.data(), host.size() * sizeof(unsigned), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(*d_tab); *d_tab = nullptr; }
    return e;
}

template <class C, typename T, typename TL>
static hipError_t launch_bg(const BgHandle &h, int mb_act, const ldpc_decode_config &cfg, const Launch &io, int num_cus,
                            const char **name)
{
    typedef LnPlan<C> PL;
    typedef BgConfig<C, T> CF;
    typedef BgLayout<C, T, CF::MAXT, CF::RL> LY;
    constexpr int SLOT = 2 * C::EB + 2 * (PL::n_priv_layers() > 0 ? PL::n_priv_layers() : 1);
    constexpr int variant = sizeof(T) == 4 ? 0 : (sizeof(TL) == 4 ? 1 : 2);
    Cfg<T> g;
    g.rule = LDPC_RULE_LNMS; g.max_iters = cfg.max_iters; g.early_exit = cfg.early_exit;
    g.float_syndrome0 = 0; g.num_active_checks = -1; g.num_refresh_vars = -1;
    g.scale = (T)cfg.scale; g.clip_tanh_arg = (T)0; g.clip_prod = (T)0; g.clip_post = (T)0; g.offset = (T)0;
    BgGeom q;
    q.Z = h.Z;
    q.F = CF::MAXT / h.Z > 16 ? 16 : CF::MAXT / h.Z;
    q.FZ = q.F * h.Z;
    q.mb_act = mb_act;
    q.npl_act = 0;
    int top = 0;  // last block column that sits in an active check
    for (int br = 0; br < mb_act; ++br) {
        q.npl_act += PL::is_priv_layer(br);
        for (int e = C::row_ptr[br]; e < C::row_ptr[br + 1]; ++e) top = C::row_col[e] > top ? C::row_col[e] : top;
    }
    // every column up to `top` sits in an active check (the extension parity columns follow the rows),
    // the handle's further columns in none
    q.nb_idle0 = top + 1;
    q.nb_idle1 = h.nb;
    q.n = h.nb * h.Z;
    q.vec = (unsigned)h.Z * sizeof(T);
    q.slot_bytes = (unsigned)PL::NS * q.vec;
    q.tab = h.d_tab + (size_t)variant * SLOT;
    auto k = bg_lnms_kernel<C, T, TL, CF::MAXT, CF::RL, CF::WPE>;
    static int static_lds = -1;  // the kernel addresses LDS from 0: it must own the whole segment
    if (static_lds < 0) {
        hipFuncAttributes attr;
        hipError_t ea = hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(k));
        if (ea != hipSuccess) return ea;
        static_lds = (int)attr.sharedSizeBytes;
    }
    if (static_lds != 0) return hipErrorInvalidConfiguration;
    const int threads = (q.FZ + 63) / 64 * 64;
    int per_cu = 0;
    hipError_t e = kernel_setup(reinterpret_cast<const void *>(k), threads, LY::LDS_BYTES, &per_cu);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    if (getenv("LDPC_DEBUG_OCCUPANCY"))
        fprintf(stderr, "bg_lnms_kernel<%s,%d-byte>: Z = %d, %d frames per workgroup of %d threads, %d of %d block rows active, "
                "%d in registers, %d workgroups per CU, %u B LDS\n",
                C::name, (int)sizeof(T), h.Z, q.F, threads, mb_act, C::MB, CF::RL, per_cu, LY::LDS_BYTES);
    const long long groups = (io.frames + q.F - 1) / q.F;
    long long grid = (long long)num_cus * per_cu * 2;
    if (grid > groups) grid = groups;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(threads), LY::LDS_BYTES, io.stream, g, q,
                       reinterpret_cast<const TL *>(io.llr), io.frames, io.hard, io.iters, io.converged,
                       reinterpret_cast<T *>(io.posterior));
    *name = sizeof(T) == 8 ? "bg_lnms_f64" : "bg_lnms_f32";
    return hipGetLastError();
}

template <class C>
static hipError_t launch_bg_code(const BgHandle &h, int mb_act, const ldpc_decode_config &cfg, int compute_dtype,
                                 const Launch &io, int num_cus, const char **name, bool *supported)
{
    *supported = false;
    if constexpr (bg_graph_ok<C>()) {
        if (mb_act < bg_min_prefix<C>() || mb_act > h.mb) return hipSuccess;
        if constexpr (kBgDouble && BgConfig<C, double>::OK) {
            if (compute_dtype == LDPC_F64 && h.Z <= BgConfig<C, double>::MAXT) {
                *supported = true;
                return io.llr_dtype == LDPC_F64 ? launch_bg<C, double, double>(h, mb_act, cfg, io, num_cus, name)
                                                : launch_bg<C, double, float>(h, mb_act, cfg, io, num_cus, name);
            }
        }
        if constexpr (BgConfig<C, float>::OK) {
            if (compute_dtype == LDPC_F32 && io.llr_dtype == LDPC_F32 && h.Z <= BgConfig<C, float>::MAXT) {
                *supported = true;
                return launch_bg<C, float, float>(h, mb_act, cfg, io, num_cus, name);
            }
        }
    }
    return hipSuccess;
}

hipError_t launch_bg_lnms(const BgHandle &h, int mb_act, const ldpc_decode_config &cfg, int compute_dtype,
                          const Launch &io, int num_cus, const char **name, bool *supported)
{
#define LDPC_LAUNCH(ID, CODE) if (h.graph == ID) return launch_bg_code<CODE>(h, mb_act, cfg, compute_dtype, io, num_cus, name, supported);
    LDPC_FOR_EACH_BASE_GRAPH(LDPC_LAUNCH)
#undef LDPC_LAUNCH
    *supported = false;
    return hipSuccess;
}

}  // namespace ldpc
