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
    for (int bc = 0; bc < C::NB; ++bc)  // the columns kept in LDS come first (the frame start and the write-out loop over them)
        if (LnPlan<C>::shared(bc) != (bc < LnPlan<C>::NS)) return false;
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

// Table entries of one block row, fetched a layer ahead: a scalar load that is waited for at once
// costs a wave its whole latency in every layer (the waves of a workgroup run in lockstep between
// barriers, nothing else hides it).  The loads are volatile: the entries are invariants, and the
// compiler would otherwise sink a load to its first use, one layer down.  It still tracks them as
// its own outstanding scalar loads and waits before the first read of a result.
typedef unsigned bg_v8 __attribute__((ext_vector_type(8)));
typedef unsigned bg_v4 __attribute__((ext_vector_type(4)));
typedef unsigned bg_v2 __attribute__((ext_vector_type(2)));
struct BgRow {
    bg_v8 lo, hi;  // words 0..7 and 8..15 of the row's 2 D words, where the row has that many
    bg_v4 m4;      // the four-word and the two-word piece of the remainder
    bg_v2 m2;
    bg_v2 p;       // entry of the private edge whose channel LLR this layer requests for a later one
};
#ifndef LDPC_BG_PREFETCH
#define LDPC_BG_PREFETCH 1
#endif
constexpr int kBgKeepAddr = 12;  // rows of up to this many edges keep their rotated addresses from the reads to the write-back
constexpr int kBgRowEdges = LDPC_BG_PREFETCH ? 10 : 0;  // rows of up to this many edges travel in a BgRow
// A row's 2 D words are fetched as pieces of 8, 8, 4 and 2 words that add up to exactly 2 D: registers
// of a wider load that nobody reads would be handed out again at once, and whatever writes them next
// (a scalar load returns out of order) must wait for the load still on its way.
constexpr int bg_full8(int D) { return 2 * D / 8 > 2 ? 2 : 2 * D / 8; }           // pieces of eight words
constexpr int bg_has4(int D) { return ((2 * D - 8 * bg_full8(D)) & 4) != 0; }   // then a piece of four
constexpr int bg_has2(int D) { return ((2 * D - 8 * bg_full8(D)) & 2) != 0; }   // then a piece of two
template <int D>
__device__ __forceinline__ void bg_row_issue(BgRow &r, unsigned long long tab, unsigned row_byte, unsigned priv_byte)
{
    static_assert(D >= 2 && D <= 10, "two to ten edges");
    typedef const volatile __attribute__((address_space(4))) unsigned char *cbytes;
    const cbytes t = (cbytes)tab;
    constexpr int W4 = 8 * bg_full8(D), W2 = W4 + 4 * bg_has4(D);
    r.p = *(const volatile __attribute__((address_space(4))) bg_v2 *)(t + priv_byte);
    if constexpr (bg_full8(D) >= 1) r.lo = *(const volatile __attribute__((address_space(4))) bg_v8 *)(t + row_byte);
    if constexpr (bg_full8(D) >= 2) r.hi = *(const volatile __attribute__((address_space(4))) bg_v8 *)(t + row_byte + 32u);
    if constexpr (bg_has4(D)) r.m4 = *(const volatile __attribute__((address_space(4))) bg_v4 *)(t + row_byte + 4u * W4);
    if constexpr (bg_has2(D)) r.m2 = *(const volatile __attribute__((address_space(4))) bg_v2 *)(t + row_byte + 4u * W2);
}
template <int I, int D>
__device__ __forceinline__ unsigned bg_row_word(const BgRow &r)
{
    constexpr int W4 = 8 * bg_full8(D), W2 = W4 + 4 * bg_has4(D);
    if constexpr (I < 8 && bg_full8(D) >= 1) return r.lo[I];
    else if constexpr (I < 16 && bg_full8(D) >= 2) return r.hi[I - 8];
    else if constexpr (bg_has4(D) && I >= W4 && I < W4 + 4) return r.m4[I - W4];
    else return r.m2[I - W2];
}

// fn(IC<B>), fn(IC<B + 1>), ... until one returns false.  Unlike a loop body that is skipped, the
// rows after the last active one are left by ONE branch out of a straight chain: no control flow
// joins between consecutive layers, hence no copies of values that are still on their way from memory
// (a copy of the prefetched channel LLR made the wave wait for it at the end of every layer).
template <int B, int E, class Fn>
__device__ __forceinline__ void static_chain(Fn &&fn)
{
    if constexpr (B < E) {
        if (fn(IC<B>{})) static_chain<B + 1, E>(fn);
    }
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
    constexpr int NSU = PL::NS % 13 == 0 ? 13 : PL::NS % 7 == 0 ? 7 : PL::NS % 4 == 0 ? 4 : PL::NS % 2 == 0 ? 2 : 1;  // channel LLRs in flight at the frame start
    extern __shared__ __align__(16) unsigned char smem[];

    // the table pointer is re-read through an empty asm wherever a block row's entries are fetched: the
    // entries are loop invariants, and hoisted out of the sweep all 2 EB of them would be spilled
    unsigned long long tab_bits = (unsigned long long)q.tab;
    auto table_bits = [&]() -> unsigned long long {
        asm volatile("" : "+s"(tab_bits));
        return tab_bits;
    };
    auto table = [&]() -> bg_u32 { return (bg_u32)table_bits(); };
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
    // flags at LDS address FLAG_OFF: words [0, 16) per slot, word 16 "any slot".  Accessed by absolute
    // LDS address (a generic pointer to them is two registers per lane for the whole kernel).
    typedef volatile __attribute__((address_space(3))) int *lds_flag;
    auto flag = [&](unsigned word) -> lds_flag { return (lds_flag)(unsigned long long)(LY::FLAG_OFF + 4u * word); };
    const long long groups = (frames + q.F - 1) / q.F;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long left64 = frames - grp * q.F;
        const int left = left64 > q.F ? q.F : (int)left64;  // frames of this group
        const bool active = lane_on && f < left;
        // loads never depend on `active`: idle lanes read a frame that exists.  The group's first
        // frame is the same for every lane (a scalar base address), the rest is a 32-bit offset.
        const TL *lf = llr + grp * q.F * n;
        const unsigned fo = (f < left ? (unsigned)f : 0u) * (unsigned)n + zu;
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
        // (real loops over the columns here and at the write-out: unrolled with constant column numbers,
        // their 2 NS addresses are invariants of the frame loop, and kept in registers across the sweeps
        // they pushed check state out into scratch memory)
        {
            unsigned pa = pz, ha = fo;
            unsigned vec_i = q.vec, Z_i = (unsigned)Z;
            asm volatile("" : "+s"(vec_i), "+s"(Z_i));  // (or the multiples of the two strides are computed at kernel start and spilled)
#pragma unroll 1
            for (int c0 = 0; c0 < PL::NS; c0 += NSU) {
                asm volatile("" : "+v"(pa), "+v"(ha));  // (or base + u * stride is split into a hoisted part and the loop's)
                T l[NSU];
#pragma unroll
                for (int u = 0; u < NSU; ++u) l[u] = (T)lf[ha + (unsigned)u * Z_i] + (T)0;  // -0.0 -> +0.0: "negative" is "sign bit set" from here on
#pragma unroll
                for (int u = 0; u < NSU; ++u)
                    if (lane_on) lds_put<T>(pa + (unsigned)u * vec_i, l[u]);
                pa += NSU * vec_i;
                ha += NSU * Z_i;
            }
        }
        if (tid <= 16) *flag((unsigned)tid) = 0;
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
        // (raw: the value is converted, and -0.0 turned into +0.0, where it is used - done at the load the
        // addition would make the wave wait for the data a layer early)
        auto load_priv_at = [&](unsigned thr, unsigned off) -> TL {
            unsigned zz = zu;
            asm volatile("" : "+v"(zz));  // one address per use, computed where it is used
            const unsigned o = bg_rot(zz, hz, hzW, thr, off);
            return *reinterpret_cast<const TL *>(reinterpret_cast<const char *>(lf) + o);
        };
        auto load_priv = [&](auto br, bg_u32 tab) -> T {
            constexpr int e = C::row_ptr[br] + PL::priv_pos(br);
            return (T)load_priv_at(tab[2 * e], tab[2 * e + 1]) + (T)0;
        };
        // index inside the frame of that variable
        auto priv_index = [&](auto br, bg_u32 tab) -> unsigned {
            constexpr int e = C::row_ptr[br] + PL::priv_pos(br);
            unsigned zz = zu;
            asm volatile("" : "+v"(zz));
            return bg_rot(zz, zz * TLS, (zz - (unsigned)Z) * TLS, tab[2 * e], tab[2 * e + 1]) / TLS;
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
        auto write_out = [&](int iters, int conv, bg_u32 tab, unsigned vec_o, unsigned Z_o) {
            // results through the group's scalar base and the lane's 32-bit offset, like the LLRs (an active
            // lane's offset into the group is fo): no 64-bit frame number is carried through the sweeps
            unsigned fo = hz;
            asm volatile("" : "+v"(fo));  // (derived again from the byte offset the sweeps keep, not kept beside it)
            fo /= TLS;
            uint8_t *ho = hard_out + grp * q.F * n + (fo - zu);
            T *po = post_out ? post_out + grp * q.F * n + (fo - zu) : nullptr;
            {
                unsigned pa = pz, ha = zu;
#pragma unroll 2
                for (int c = 0; c < PL::NS; ++c) {
                    asm volatile("" : "+v"(pa), "+v"(ha));
                    const T v = lds_get<T>(pa);
                    ho[ha] = (uint8_t)hard_bit<1>(v);
                    if (po) po[ha] = v;
                    pa += vec_o;
                    ha += Z_o;
                }
            }
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
            for (unsigned bc = (unsigned)q.nb_idle0; bc < (unsigned)q.nb_idle1; ++bc) {  // in no active check: the channel decides
                const T v = (T)lf[fo + bc * Z_o] + (T)0;
                ho[bc * Z_o + zu] = (uint8_t)hard_bit<1>(v);
                if (po) po[bc * Z_o + zu] = v;
            }
            if (z == 0) { iters_out[grp * q.F + f] = iters; conv_out[grp * q.F + f] = (uint8_t)conv; }
        };

        // channel LLR of the private variable the sweep needs next, requested one layer ahead
        // (cyclically over the active layers that have one)
        constexpr int SYN_FIRST = 2, SYN_EVERY = 6;  // block rows between looks at the outcome of the syndrome test
        TL lp_next = (TL)0;
        if constexpr (NPL > 0) {
            const bg_u32 tab = table();
            lp_next = load_priv_at(tab[2 * C::EB], tab[2 * C::EB + 1]);
        }
        BgRow pf;  // entries of the next block row, on their way

        // A slot that has converged is left alone (its posteriors and check state stay as they were
        // at the successful test) and everything is written once, after the loop.
        bool done = !active;
        unsigned res = (unsigned)(g.early_exit ? g.max_iters - 1 : g.max_iters);  // iterations, bit 31: converged
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
                static_chain<0, MB>([&](auto br) -> bool {
                    if constexpr (br == SYN_FIRST || (br > SYN_FIRST && (br - SYN_FIRST) % SYN_EVERY == 0)) {
                        // (a wave's LDS operations complete in order: the read sees what its own lanes just wrote)
                        if (bad) *flag((unsigned)f) = 1;
                        const bool known = done || *flag((unsigned)f) != 0;
                        skip = __ballot(!known) == 0ull;
                    }
                    if (br >= MINP && !row_on(br)) return false;
                    if (skip) return false;      // wave-uniform: the rest of the rows is not needed
                    const bg_u32 tab = table();  // before the lanes part ways: a scalar value
                    if (done) return true;
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
                    return true;
                });
                if (bad) { *flag((unsigned)f) = 1; *flag(16u) = 1; }
                __syncthreads();
                const bool slot_ok = *flag((unsigned)f) == 0;
                const bool any_bad = *flag(16u) != 0;
                if (last) {
                    if (!done && slot_ok) res |= 0x80000000u;
                    break;
                }
                if (!done && slot_ok) { res = (unsigned)t | 0x80000000u; done = true; }
                if (!any_bad) break;  // every slot that was still running has just converged
                __syncthreads();
                if (tid <= 16) *flag((unsigned)tid) = 0;
            }
            // ---- one sweep over the active layers
            const bool want_bits = g.early_exit || t == g.max_iters - 1;
            static_chain<0, MB>([&](auto br) -> bool {
                constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0, PP = PL::priv_pos(br);
                constexpr bool HAVE = br > 0 && D <= kBgRowEdges;  // the previous layer requested this row's entries
                if (br >= MINP && !row_on(br)) return false;  // scalar branch out of the sweep: the same for the whole workgroup
                if constexpr (PL::barrier_before(br)) __syncthreads();
                BgRow cur;
                if constexpr (HAVE) cur = pf;
                if constexpr (br + 1 < MB) {
                    if constexpr (PL::rdeg(br + 1) <= kBgRowEdges) {
                        // with them the entry of the private edge whose LLR layer br + 1 will request: that of
                        // the next block row with a private variable, or of the first one when the sweep ends before
                        constexpr int e1 = C::row_ptr[br + 1];
                        constexpr int nx = PL::next_priv_layer(br + 1), fp = PL::first_priv_layer();
                        unsigned pb = 0;
                        if constexpr (NPL > 0) {
                            constexpr unsigned eb_nx = 8u * (C::row_ptr[nx] + PL::priv_pos(nx)), eb_fp = 8u * (C::row_ptr[fp] + PL::priv_pos(fp));
                            pb = eb_fp;
                            if constexpr (nx > br + 1) pb = row_on(nx) ? eb_nx : eb_fp;
                        }
                        bg_row_issue<PL::rdeg(br + 1)>(pf, table_bits(), 8u * e1, pb);
                    }
                }
                TL lp_raw = (TL)0;
                const bg_u32 tab = table();
                if constexpr (PP >= 0) {
                    lp_raw = lp_next;
                    if constexpr (HAVE) {
                        lp_next = load_priv_at(cur.p[0], cur.p[1]);
                    } else {
                        constexpr int j = PL::priv_layer_ord(br);
                        asm volatile("" : "+s"(npl_bits));
                        const int jn = j + 1 >= npl_bits ? 0 : j + 1;
                        lp_next = load_priv_at(tab[2 * C::EB + 2 * jn], tab[2 * C::EB + 2 * jn + 1]);
                    }
                }
                if (!done) {
                    T qv[D];
                    unsigned off[D <= kBgKeepAddr ? D : 1];  // rotated addresses, kept from the reads to the write-back
                    T Ao, Bo;
                    unsigned w;
                    load_state(br, Ao, Bo, w);
                    const int idx_old = (int)((w >> 24) & 31u);
                    const unsigned wp = w ^ (unsigned)((int)w >> 31);  // bit D-1-p: sign of the old message on edge p
                    T m1, m2;
                    unsigned zr = zu;
                    asm volatile("" : "+v"(zr));  // as above: hundreds of hoisted addresses would spill
                    // A long row fetches its entries in two halves, once for the reads and once more for the
                    // write-back, and forms its addresses twice: nineteen posteriors, nineteen addresses and
                    // thirty-eight entries live at once pushed check state out into scratch memory and kernel
                    // arguments out of the scalar registers.
                    constexpr bool KEEP = D <= kBgKeepAddr;
                    constexpr int H = (D + 1) / 2;
                    auto half = [&](auto h, auto &&fn) {  // fn(p, Z - s, offset) for the edges of half h, descending if h says so
                        constexpr int p0 = (h & 1) * H, p1 = (h & 1) ? D : H;
                        unsigned long long tb = (unsigned long long)tab;
                        asm volatile("" : "+s"(tb));  // a fetch of its own: not the values of the other pass kept alive
                        const bg_u32 t2 = (bg_u32)tb;
                        unsigned th[2 * (p1 - p0)];
                        static_for<0, 2 * (p1 - p0)>([&](auto i) { th[i] = t2[2 * (e0 + p0) + i]; });
                        __builtin_amdgcn_sched_barrier(0);
                        static_for<0, p1 - p0>([&](auto k) {
                            constexpr int p = (h & 2) ? p1 - 1 - k : p0 + k;
                            fn(IC<p>{}, th[2 * (p - p0)], th[2 * (p - p0) + 1]);
                        });
                    };
                    unsigned te[KEEP ? 2 * D : 1];
                    if constexpr (!KEEP) {
                        auto rd = [&](auto p, unsigned thr, unsigned o) {
                            if constexpr (p != PP) {
                                qv[p] = lds_get<T>(bg_rot(zr, pz, pzW, thr, o));
                                if constexpr (ST == 4) __builtin_amdgcn_sched_barrier(0);
                            }
                        };
                        half(IC<0>{}, rd);
                        half(IC<1>{}, rd);
                    } else {
                        if constexpr (HAVE) {
                            static_for<0, 2 * D>([&](auto i) { te[i] = bg_row_word<i, D>(cur); });
                        } else {  // the first row of the sweep (or no prefetch): fetched here
                            static_for<0, 2 * D>([&](auto i) { te[i] = tab[2 * e0 + i]; });
                            __builtin_amdgcn_sched_barrier(0);  // the row's entries in one go, ahead of the first address
                        }
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
                    }
                    // bit 31 of ws: sign of the old message on the current edge; every edge shifts it out
                    // and the sign of its input in at the other end, so the word ends as the new signs
                    unsigned ws = wp << (32 - D);
                    static_for<0, D>([&](auto p) {
                        if constexpr (p == PP) {
                            qv[p] = (T)lp_raw + (T)0;  // a variable of degree one always sends its channel LLR
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
                    auto wr = [&](auto p, unsigned addr) {
                        const T pn = SNum<T>::pick_add(qv[p], m1, Anp, Bnp, minbits);  // the variable's new posterior
                        if constexpr (p != PP) {
                            lds_put<T>(addr, pn);
                            if constexpr (ST == 4) __builtin_amdgcn_sched_barrier(0);  // block, write, block, write
                        }
                        else if (want_bits) privbit = hard_bit<1>(pn) << 30;
                    };
                    if constexpr (KEEP) {
                        static_for<0, D>([&](auto pp) { wr(IC<D - 1 - pp>{}, off[D - 1 - pp]); });
                    } else {  // last edge first, as above: the comparison outcomes are shifted in from the right
                        auto wb = [&](auto p, unsigned thr, unsigned o) {
                            unsigned a = 0;
                            if constexpr (p != PP) a = bg_rot(zr, pz, pzW, thr, o);
                            wr(p, a);
                        };
                        half(IC<3>{}, wb);
                        half(IC<2>{}, wb);
                    }
                    const int idx = __builtin_ctz(minbits);  // the first minimum; some bit is always set
                    const unsigned wn = sq | ((unsigned)idx << 24) | px | privbit;
                    if constexpr (br < RL) { A[br] = An; B[br] = Bn; W[br] = wn; }
                    else {
                        lds_put<T>(stA + LY::a_off(br - RL), An);
                        lds_put<T>(stA + LY::b_off(br - RL), Bn);
                        lds_put<unsigned>(stW + LY::w_off(br - RL), wn);
                    }
                }
                return true;
            });
        }
        const bg_u32 tab_out = table();
        unsigned vec_o = q.vec, Z_o = (unsigned)Z;
        asm volatile("" : "+s"(vec_o), "+s"(Z_o));
        if (active) write_out((int)(res & 0x7fffffffu), (int)(res >> 31), tab_out, vec_o, Z_o);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ host
// Two sizes of workgroup per base graph and arithmetic.  LARGE: one workgroup of up to 768 threads (384
// in double) that fills the compute unit alone - what Z = 384 needs.  SMALL: up to 256 threads (128 in
// double), three (four) workgroups per compute unit: for small lifting sizes the frames of a workgroup
// wait for each other at every barrier and at every syndrome test, and several small workgroups that
// run independently of each other hide that better than sixteen frames in lockstep.
// Register budget as in decode_lnms.hip: three waves per SIMD in float (168 registers), two in double
// (256); about 57 registers are left for everything that is not check state.
template <class C, typename T, bool SMALL>
struct BgConfig {
    static constexpr int MAXT = sizeof(T) == 4 ? (SMALL ? 256 : 768) : (SMALL ? 128 : 384);
    static constexpr int RPL = 2 * (int)sizeof(T) / 4 + 1;
    static constexpr int WPE = sizeof(T) == 4 ? 3 : 2;
    static constexpr int VGPRS = WPE <= 2 ? 256 : (512 / WPE) / 8 * 8;
    static constexpr int STATE_REGS = VGPRS - 57 > 180 ? 180 : VGPRS - 57;
    static constexpr int RL = C::MB < STATE_REGS / RPL ? C::MB : STATE_REGS / RPL;
    static constexpr bool OK = BgLayout<C, T, MAXT, RL>::LDS_BYTES <= 160 * 1024;
    static int slots(int Z) { return MAXT / Z > 16 ? 16 : MAXT / Z; }  // frame slots per workgroup (sixteen flags)
};

// One translation unit per base graph (-DLDPC_BG_GRAPH=0: BG1, 1: BG2): connectivity of the packaged NR
// tables.  The entry points that pick the graph are compiled with graph 0.
#ifndef LDPC_BG_GRAPH
#define LDPC_BG_GRAPH 0
#endif
#if LDPC_BG_GRAPH == 0
typedef Code_nr_bg1_z384 BgCode;
#define LDPC_BG_NAME(f) f##_0
#else
typedef Code_nr_bg2_z40 BgCode;
#define LDPC_BG_NAME(f) f##_1
#endif
#ifdef LDPC_BG_DEV_ONLY  // development aid: float only
constexpr bool kBgDouble = false;
#else
constexpr bool kBgDouble = true;
#endif
bool bg_match_0(int Z, int mb, int nb, int Eb, const int *er, const int *ec);
bool bg_match_1(int Z, int mb, int nb, int Eb, const int *er, const int *ec);
void bg_fill_0(int Z, int mb, const int *es, std::vector<unsigned> &host);
void bg_fill_1(int Z, int mb, const int *es, std::vector<unsigned> &host);
hipError_t launch_bg_lnms_0(const BgHandle &h, int mb_act, const ldpc_decode_config &cfg, int compute_dtype,
                            const Launch &io, int num_cus, const char **name, bool *supported);
hipError_t launch_bg_lnms_1(const BgHandle &h, int mb_act, const ldpc_decode_config &cfg, int compute_dtype,
                            const Launch &io, int num_cus, const char **name, bool *supported);

bool LDPC_BG_NAME(bg_match)(int Z, int mb, int nb, int Eb, const int *er, const int *ec)
{
    typedef BgCode C;
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

constexpr int kBgTabWords = 2 * BgCode::EB + 2 * (LnPlan<BgCode>::n_priv_layers() > 0 ? LnPlan<BgCode>::n_priv_layers() : 1);

// Run-time tables of a handle: three variants (float / float LLRs, double / float LLRs, double / double LLRs).
void LDPC_BG_NAME(bg_fill)(int Z, int mb, const int *es, std::vector<unsigned> &host)
{
    typedef BgCode C;
    typedef LnPlan<C> PL;
    constexpr int SLOT = kBgTabWords;
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

#if LDPC_BG_GRAPH == 0
int bg_graph_match(int Z, int mb, int nb, int Eb, const int *er, const int *ec)
{
    if (bg_match_0(Z, mb, nb, Eb, er, ec)) return 0;
#ifndef LDPC_BG_DEV_ONLY
    if (bg_match_1(Z, mb, nb, Eb, er, ec)) return 1;
#endif
    return -1;
}

// Tables of a handle whose description matched base graph `graph`, in one device allocation.
hipError_t bg_tables_create(int graph, int Z, int mb, const int *es, unsigned **d_tab)
{
    std::vector<unsigned> host;
    if (graph == 0) bg_fill_0(Z, mb, es, host);
#ifndef LDPC_BG_DEV_ONLY
    if (graph == 1) bg_fill_1(Z, mb, es, host);
#endif
    if (host.empty()) return hipErrorInvalidValue;
    host.resize(host.size() + 32, 0u);  // scalar loads of whole pieces may reach past the last entry
    hipError_t e = hipMalloc(reinterpret_cast<void **>(d_tab), host.size() * sizeof(unsigned));
    if (e != hipSuccess) return e;
    e = hipMemcpy(*d_tab, host.data(), host.size() * sizeof(unsigned), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(*d_tab); *d_tab = nullptr; }
    return e;
}

hipError_t launch_bg_lnms(const BgHandle &h, int mb_act, const ldpc_decode_config &cfg, int compute_dtype,
                          const Launch &io, int num_cus, const char **name, bool *supported)
{
    if (h.graph == 0) return launch_bg_lnms_0(h, mb_act, cfg, compute_dtype, io, num_cus, name, supported);
#ifndef LDPC_BG_DEV_ONLY
    if (h.graph == 1) return launch_bg_lnms_1(h, mb_act, cfg, compute_dtype, io, num_cus, name, supported);
#endif
    *supported = false;
    return hipSuccess;
}
#endif

template <class C, typename T, typename TL, bool SMALL>
static hipError_t launch_bg(const BgHandle &h, int mb_act, const ldpc_decode_config &cfg, const Launch &io, int num_cus,
                            const char **name)
{
    typedef LnPlan<C> PL;
    typedef BgConfig<C, T, SMALL> CF;
    typedef BgLayout<C, T, CF::MAXT, CF::RL> LY;
    constexpr int variant = sizeof(T) == 4 ? 0 : (sizeof(TL) == 4 ? 1 : 2);
    Cfg<T> g;
    g.rule = LDPC_RULE_LNMS; g.max_iters = cfg.max_iters; g.early_exit = cfg.early_exit;
    g.float_syndrome0 = 0; g.num_active_checks = -1; g.num_refresh_vars = -1;
    g.scale = (T)cfg.scale; g.clip_tanh_arg = (T)0; g.clip_prod = (T)0; g.clip_post = (T)0; g.offset = (T)0;
    BgGeom q;
    q.Z = h.Z;
    q.F = CF::slots(h.Z);
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
    q.tab = h.d_tab + (size_t)variant * kBgTabWords;
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
        fprintf(stderr, "bg_lnms_kernel<%s,%d-byte,%d>: Z = %d, %d frames per workgroup of %d threads, %d of %d block rows active, "
                "%d in registers, %d workgroups per CU, %u B LDS\n",
                C::name, (int)sizeof(T), CF::MAXT, h.Z, q.F, threads, mb_act, C::MB, CF::RL, per_cu, LY::LDS_BYTES);
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

// Small workgroups when they put at least nine tenths as many lanes to work on a compute unit as the
// large one does (Z = 40: 3 x 6 frames against 16; Z = 128: 3 x 2 against 6; Z = 96: 3 x 2 against 8 - large).
template <class C, typename T>
static bool bg_use_small(int Z)
{
    typedef BgConfig<C, T, true> S;
    typedef BgConfig<C, T, false> L;
    if constexpr (!S::OK) return false;
    else {
        if (Z > S::MAXT) return false;
        const int per_cu = sizeof(T) == 4 ? 3 : 4;
        return 10 * per_cu * S::slots(Z) >= 9 * L::slots(Z);
    }
}

hipError_t LDPC_BG_NAME(launch_bg_lnms)(const BgHandle &h, int mb_act, const ldpc_decode_config &cfg, int compute_dtype,
                                        const Launch &io, int num_cus, const char **name, bool *supported)
{
    typedef BgCode C;
    *supported = false;
    if constexpr (bg_graph_ok<C>()) {
        if (mb_act < bg_min_prefix<C>() || mb_act > h.mb) return hipSuccess;
        if constexpr (kBgDouble && BgConfig<C, double, false>::OK) {
            if (compute_dtype == LDPC_F64 && h.Z <= BgConfig<C, double, false>::MAXT) {
                *supported = true;
                const bool small = bg_use_small<C, double>(h.Z);
                if (io.llr_dtype == LDPC_F64)
                    return small ? launch_bg<C, double, double, true>(h, mb_act, cfg, io, num_cus, name)
                                 : launch_bg<C, double, double, false>(h, mb_act, cfg, io, num_cus, name);
                return small ? launch_bg<C, double, float, true>(h, mb_act, cfg, io, num_cus, name)
                             : launch_bg<C, double, float, false>(h, mb_act, cfg, io, num_cus, name);
            }
        }
        if constexpr (BgConfig<C, float, false>::OK) {
            if (compute_dtype == LDPC_F32 && io.llr_dtype == LDPC_F32 && h.Z <= BgConfig<C, float, false>::MAXT) {
                *supported = true;
                return bg_use_small<C, float>(h.Z) ? launch_bg<C, float, float, true>(h, mb_act, cfg, io, num_cus, name)
                                                   : launch_bg<C, float, float, false>(h, mb_act, cfg, io, num_cus, name);
            }
        }
    }
    return hipSuccess;
}

}  // namespace ldpc
