// Compile-time analysis of a circulant base graph for the row-layered kernels (decode_lnms.hip,
// decode_lnms_h.hip): which block columns are shared between checks and kept in LDS, which are
// "private" to one check, where barriers are needed between layers, and LDS access by absolute
// address.
#pragma once
#include "spec_common.h"

namespace ldpc {

template <class C>
struct LnPlan {
    static constexpr int cdeg(int bc) { return C::col_ptr[bc + 1] - C::col_ptr[bc]; }
    static constexpr int rdeg(int br) { return C::row_ptr[br + 1] - C::row_ptr[br]; }
    static constexpr bool has_pairs()
    {
        for (int k = 0; k < C::EB; ++k) if (C::col_pair[k]) return true;
        return false;
    }
    // position inside block row br of its degree-one column, -1 if it has none
    static constexpr int priv_pos(int br)
    {
        int pos = -1;
        for (int e = C::row_ptr[br]; e < C::row_ptr[br + 1]; ++e)
            if (cdeg(C::row_col[e]) == 1) pos = e - C::row_ptr[br];
        return pos;
    }
    static constexpr int priv_count(int br)
    {
        int c = 0;
        for (int e = C::row_ptr[br]; e < C::row_ptr[br + 1]; ++e) c += cdeg(C::row_col[e]) == 1;
        return c;
    }
    static constexpr bool supported()
    {
        if (has_pairs() || C::MAX_DC > 24) return false;  // sign bits and argmin position share one word
        int layers = 0;
        for (int br = 0; br < C::MB; ++br) {
            if (priv_count(br) > 1) return false;
            layers += rdeg(br) >= 2;
        }
        return layers > 0;
    }
    static constexpr bool shared(int bc) { return cdeg(bc) >= 2; }
    static constexpr int lds_index(int bc)
    {
        int k = 0;
        for (int c = 0; c < bc; ++c) k += shared(c);
        return k;
    }
    static constexpr int NS = lds_index(C::NB);
    // does layer br touch a shared column that an earlier layer since the last barrier touched?
    // (the first layer of a sweep always synchronises with whatever ran before the sweep)
    static constexpr bool barrier_before(int br)
    {
        bool used[C::NB] = {};
        bool first = true;
        for (int r = 0; r <= br; ++r) {
            if (rdeg(r) < 2) continue;
            bool clash = first;
            for (int e = C::row_ptr[r]; e < C::row_ptr[r + 1]; ++e)
                if (shared(C::row_col[e]) && used[C::row_col[e]]) clash = true;
            if (clash) for (int c = 0; c < C::NB; ++c) used[c] = false;
            for (int e = C::row_ptr[r]; e < C::row_ptr[r + 1]; ++e) used[C::row_col[e]] = true;
            first = false;
            if (r == br) return clash;
        }
        return false;
    }
    static constexpr bool is_priv_layer(int br) { return rdeg(br) >= 2 && priv_pos(br) >= 0; }
    // the next layer after br (cyclically) that has a private variable; -1 if there is none
    static constexpr int next_priv_layer(int br)
    {
        for (int k = 1; k <= C::MB; ++k) {
            const int r = (br + k) % C::MB;
            if (is_priv_layer(r)) return r;
        }
        return -1;
    }
    static constexpr int first_priv_layer() { return next_priv_layer(C::MB - 1); }
    // j-th layer with a private variable, their number, and the inverse map
    static constexpr int n_priv_layers()
    {
        int c = 0;
        for (int br = 0; br < C::MB; ++br) c += is_priv_layer(br);
        return c;
    }
    static constexpr int priv_layer(int j)
    {
        for (int br = 0; br < C::MB; ++br)
            if (is_priv_layer(br) && j-- == 0) return br;
        return 0;
    }
    static constexpr int priv_layer_ord(int br)
    {
        int c = 0;
        for (int r = 0; r < br; ++r) c += is_priv_layer(r);
        return c;
    }
    // the same over all block rows that have a private variable (layers or not)
    static constexpr int n_priv_rows()
    {
        int c = 0;
        for (int br = 0; br < C::MB; ++br) c += priv_pos(br) >= 0;
        return c;
    }
    static constexpr int priv_row(int j)
    {
        for (int br = 0; br < C::MB; ++br)
            if (priv_pos(br) >= 0 && j-- == 0) return br;
        return 0;
    }
    static constexpr int priv_row_ord(int br)
    {
        int c = 0;
        for (int r = 0; r < br; ++r) c += priv_pos(r) >= 0;
        return c;
    }
    static constexpr int barriers_per_sweep()
    {
        int c = 0;
        for (int br = 0; br < C::MB; ++br) c += rdeg(br) >= 2 && barrier_before(br);
        return c;
    }
};

// LDS of one frame slot: [state of the first spilled block row, or padding][P][state of the other
// spilled block rows].  P starts at least one Z-vector into the slot so that "one vector back"
// never leaves the slot (see ln_addr).
template <class C, typename T, int F, int RL>
struct LnLayout {
    typedef LnPlan<C> PL;
    static constexpr int ST = sizeof(T);
    static constexpr int VEC = C::Z * ST;                    // bytes of one Z-vector of T
    static constexpr int THREADS = ((F * C::Z + 63) / 64) * 64;
    static constexpr bool SINGLE = THREADS == 64;
    static constexpr int P_BYTES = ((PL::NS * VEC + 15) / 16) * 16;
    static constexpr int LAYER_BYTES = C::Z * (2 * ST + 4);  // A[Z], B[Z], W[Z] of one spilled block row
    static constexpr int SPILLED = C::MB - RL;
    static constexpr int P_OFF = SPILLED > 0 ? LAYER_BYTES : ((VEC + 15) / 16) * 16;
    static constexpr int REST_OFF = P_OFF + P_BYTES;
    static constexpr int SLOT_BYTES = ((REST_OFF + (SPILLED > 1 ? SPILLED - 1 : 0) * LAYER_BYTES + 127) / 128) * 128 + (F > 1 ? 32 : 0);
    static constexpr int FLAG_OFF = F * SLOT_BYTES;
    static constexpr int LDS_BYTES = FLAG_OFF + 64;
    // slot offset of the state of spilled block row k = br - RL
    static constexpr int state_off(int k) { return k == 0 ? 0 : REST_OFF + (k - 1) * LAYER_BYTES; }
};

// LDS address of element ((z + S) mod Z) of shared column CI.  zl = z * ST; zbP = slot base + zl,
// zbPW = zbP + one vector.  The wrapped and the unwrapped case differ by one vector, so one
// compare against a constant and one select pick the base register, and everything else is a
// constant that lands in the instruction's offset field.
// WV >= 0: the code is compiled for wave WV of a slot, i.e. for z in [64 WV, 64 WV + 63].  All of a
// wave's lanes wrap, or none does, except in the one wave that holds the boundary Z - S: the others
// need no instruction at all for the address.
template <int WV, class LY, int CI, int S>
__device__ __forceinline__ unsigned ln_addr(unsigned zl, unsigned zbP, unsigned zbPW)
{
    constexpr int Zc = LY::VEC / LY::ST;
    if constexpr (S == 0) return zbP + (unsigned)(LY::P_OFF + CI * LY::VEC);
    else if constexpr (WV >= 0 && 64 * WV >= Zc - S)
        return zbP + (unsigned)(LY::P_OFF + CI * LY::VEC + S * LY::ST - LY::VEC);  // every lane wraps
    else if constexpr (WV >= 0 && 64 * WV + 63 < Zc - S)
        return zbP + (unsigned)(LY::P_OFF + CI * LY::VEC + S * LY::ST);            // no lane wraps
    else {
        unsigned base;  // = zl >= VEC - S ST ? zbP : zbPW
        if constexpr (LY::ST == 4) {
            // written out with the threshold as a literal of the compare: the compiler keeps thresholds in
            // scalar registers (one more instruction each) and separates compare and select by wait states
            asm("v_cmp_le_u32_e32 vcc, %3, %4\n\tv_cndmask_b32_e32 %0, %1, %2, vcc"
                : "=v"(base) : "v"(zbPW), "v"(zbP), "i"(LY::VEC - S * LY::ST), "v"(zl) : "vcc");
        } else {  // double: the blocks would cost the compiler's schedule more registers than the kernel has
            base = zl >= (unsigned)(LY::VEC - S * LY::ST) ? zbP : zbPW;
        }
        return base + (unsigned)(LY::P_OFF + CI * LY::VEC + S * LY::ST - LY::VEC);
    }
}

// LDS accesses by absolute byte address.  The kernels below have no static LDS, so the dynamic
// segment starts at address 0 (the launcher checks that); with plain integers as addresses every
// constant part of an offset ends up in the instruction's immediate field.
template <typename T>
__device__ __forceinline__ T lds_get(unsigned addr)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) T *>((unsigned long long)addr);
}
template <typename T>
__device__ __forceinline__ void lds_put(unsigned addr, T v)
{
    *reinterpret_cast<__attribute__((address_space(3))) T *>((unsigned long long)addr) = v;
}

}  // namespace ldpc
