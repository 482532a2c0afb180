// Row-layered normalised min-sum in packed half precision, TWO frames per lane (LDPC_RULE_LNMS with
// compute_dtype LDPC_F16): the reduced-precision fast mode of decode_lnms.hip.  Neither the rule nor
// the arithmetic has a counterpart in the reference (flooding schedules in float64 only); the mode is
// bit-exact with nothing and is validated statistically against the float32 kernel on the same
// frames (tests/test_gpu_half_lnms.py).
//
// Organisation as in decode_lnms.hip: thread (f, z) owns check (br, z) of every block row of slot f,
// posteriors of the shared block columns in LDS, channel values of the degree-one columns re-read
// from the LLR array, compressed check state in registers / LDS.  A slot holds a PAIR of frames:
// every 32-bit word carries the two frames' half-precision numbers (or 16-bit state fields) side by
// side and every packed instruction works on both, so the address arithmetic, the LDS accesses and
// the loop overhead of a layer are paid once per pair.
//
// Packed instructions have no compare or select.  Both selects of the rule are done on the bit
// patterns instead: with e = 1 where a 16-bit difference is zero (one saturating subtraction) the
// magnitude is A + e (B - A) in INTEGER arithmetic (v_pk_mad_u16), exact for positive half-precision
// patterns.  Check state of one pair:
//   AB     scale * smallest / second smallest input magnitude, clamped to 2000 and cut to the upper
//          eight of their fifteen magnitude bits (five exponent, three mantissa bits), per frame A in
//          the low byte and B in the high byte of its half.  The cut values are the messages that are
//          added to the posteriors as well: what a layer adds is exactly what its next visit takes off.
//   W      per frame 16 bits: bit 15 sign parity of the inputs | bit 14 hard decision of the check's
//          degree-one variable | bits 13-10 position of the smallest input | bits 9-0 input signs
//          (first edge highest).  Checks of more than ten inputs keep a second word for the signs of
//          inputs 10... and use five position bits; checks of at most seven inputs keep, in place of
//          the position, one bit per edge "this edge held the smallest magnitude" (bits 13 down), which
//          saves turning the second pass's outcome into a position and back.
// Hard decisions are the sign bits of the posteriors (a posterior of exactly +0 decides 0 here).
// A frame of a pair that converges is written out at once; its partner goes on.
#include <cstdio>
#include <cstdlib>

#include "spec_common.h"
#include "lnms_plan.h"
#include "half2_ops.h"
#include "extra_kernels.h"
#include "sweep_fused.h"

namespace ldpc {

constexpr int kShortRow = 10;  // inputs whose signs fit the first state word
constexpr int kTinyRow = 7;    // inputs of a check that keeps one "held the minimum" bit per edge instead of a position
constexpr unsigned kMagCut = 0x7f807f80u;  // the upper eight magnitude bits of both halves
constexpr unsigned kMagHalfStep = 0x00400040u;  // half of the lowest kept bit: no carry leaves a half (magnitudes <= 2000)
#ifndef LDPC_LNMSH_DOUBLE_SLOTS
#define LDPC_LNMSH_DOUBLE_SLOTS 1
#endif
#ifndef LDPC_LNMSH_WPE
#define LDPC_LNMSH_WPE 0
#endif
#ifndef LDPC_LNMSH_PREFETCH
#define LDPC_LNMSH_PREFETCH 1
#endif
#ifndef LDPC_LNMSH_RESERVE
#define LDPC_LNMSH_RESERVE 53
#endif

template <class C>
struct LnHPlan {
    typedef LnPlan<C> PL;
    static constexpr bool is_long(int br) { return PL::rdeg(br) > kShortRow; }
    static constexpr bool is_tiny(int br) { return PL::rdeg(br) <= kTinyRow; }
    static constexpr int da(int br) { return is_long(br) ? kShortRow : PL::rdeg(br); }
    static constexpr int words(int br) { return is_long(br) ? 3 : 2; }
    static constexpr bool supported() { return PL::supported() && C::MAX_DC <= 2 * kShortRow; }
    // registers taken by the state of block rows [0, rl)
    static constexpr int regs(int rl)
    {
        int c = 0;
        for (int r = 0; r < rl && r < C::MB; ++r) c += words(r);
        return c;
    }
    // block rows [rl, MB) keep their state in LDS as two words: none of them may be long
    static constexpr bool spill_ok(int rl)
    {
        for (int r = rl; r < C::MB; ++r) if (is_long(r)) return false;
        return true;
    }
};

// FUSED: the channel in front and the error accounting behind run inside the kernel (sweep_fused.h).
// The workgroup draws the LLRs of its 2 F frames into its own slice of `scratch` ([2 F][n] floats in
// device memory, reused group after group) and decodes from there; a frame's bit errors against the
// transmitted word are counted where the decode-only kernel writes its decisions out (at the test it
// passes, or after the last sweep), the per-frame "has an error" flags meet in a second row of LDS
// flags behind the syndrome flags.  llr and the result arrays are unused.
template <class C, int F, int RL, int WPE, bool FUSED = false>
__global__ void __launch_bounds__((LnLayout<C, float, F, RL>::THREADS)) __attribute__((amdgpu_waves_per_eu(WPE, 8)))
lnms_h_kernel(int max_iters, int early_exit, float scale_f, const float *__restrict__ llr, long long frames,
              uint8_t *__restrict__ hard_out, int *__restrict__ iters_out, uint8_t *__restrict__ conv_out, SweepDev sp,
              float *scratch)
{
    typedef LnPlan<C> PL;
    typedef LnHPlan<C> HP;
    typedef LnLayout<C, float, F, RL> LY;  // a pair's word has the size of a float: same layout
    static_assert(HP::supported() && HP::spill_ok(RL), "code shape not supported by the packed layered kernel");
    static_assert(F <= 8, "the per-frame syndrome flags are one row of sixteen");
    constexpr int Z = C::Z, NB = C::NB, MB = C::MB;
    constexpr int n = NB * Z;
    extern __shared__ __align__(16) unsigned char smem[];

    const int tid = threadIdx.x;
    const bool lane_on = tid < F * Z;
    const int f = (F == 1 || !lane_on) ? 0 : tid / Z;
    const int z = lane_on ? tid - f * Z : 0;
    const unsigned zb = (unsigned)z * 4;
    const unsigned zbP = (unsigned)f * LY::SLOT_BYTES + zb;  // absolute LDS addresses from here on
    const unsigned zbPW = zbP + LY::VEC;
    const unsigned stA = zbP;                                // spilled check state of this thread: A, B, W
    int *slot_bad = reinterpret_cast<int *>(smem + LY::FLAG_OFF);  // [2 f + h]: frame h of slot f failed the test
    volatile int *slot_bad_now = slot_bad;                   // polled while other waves may still be writing
    const unsigned long long slot_mask = (Z >= 64 ? ~0ull : ((1ull << (Z & 63)) - 1ull)) << ((f * Z) & 63);
    unsigned long long wave_slot_mask = slot_mask;           // lanes of this wave on the same slot as this lane
    if constexpr (!LY::SINGLE && F > 1) {
        static_for<0, F>([&](auto ff) {
            const unsigned long long m = __ballot(lane_on && f == ff);
            if (f == ff) wave_slot_mask = m;
        });
    } else if constexpr (!LY::SINGLE) {
        wave_slot_mask = ~0ull;
    }
    const h2 scale2 = pack_half2(scale_f, scale_f);
    // fused: the launch adds 192 bytes of LDS behind the syndrome flags.  Nothing of the accounting lives in
    // registers across the sweeps (the check state needs them all): per-frame outcomes and the
    // workgroup's five counters sit in LDS and are added to with LDS atomics where a frame ends.
    int *slot_err = slot_bad + 16;   // [2 f + h]: frame h of slot f has a bit error somewhere
    int *slot_conv = slot_bad + 32;  // [2 f + h]: it passed a syndrome test
    unsigned long long *wg_cnt = reinterpret_cast<unsigned long long *>(slot_bad + 48);  // frames, not converged, converged wrong, bit errors, iterations
    if constexpr (FUSED) {
        if (tid < 5) wg_cnt[tid] = 0ull;  // the first group's staging barrier orders this before any use
    }
    static_assert(!FUSED || (C::NB * C::Z) % 4 == 0, "fused: four positions per draw");

    const long long pairs = (frames + 1) / 2, groups = (pairs + F - 1) / F;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long fr0 = (F == 1 ? grp : grp * F + f) * 2, fr1 = fr0 + 1;
        const bool act0 = lane_on && fr0 < frames, act1 = lane_on && fr1 < frames;
        // loads never depend on the flags: a lane without a frame reads the group's first frame,
        // which exists.  One scalar base address per group, 32-bit offsets per lane.
        const char *lf;
        unsigned o0, o1;  // bytes
        if constexpr (FUSED) {  // every slot of the slice holds a frame (idle ones a copy of the last)
            lf = reinterpret_cast<const char *>(sweep_stage_group<n, 2 * F>(sp, scratch, grp, frames));
            o0 = ((unsigned)(2 * f) * (unsigned)n + (unsigned)z) * 4u;
            o1 = ((unsigned)(2 * f + 1) * (unsigned)n + (unsigned)z) * 4u;
        } else {
            lf = reinterpret_cast<const char *>(llr + grp * (2 * F) * n);
            o0 = ((fr0 < frames ? (unsigned)(2 * f) : 0u) * (unsigned)n + (unsigned)z) * 4u;
            o1 = ((fr1 < frames ? (unsigned)(2 * f + 1) : 0u) * (unsigned)n + (unsigned)z) * 4u;
        }
        const h2 clamp2 = kHalfClamp2, big2 = kHalfMax2;
        auto load_pair = [&](unsigned off) -> h2 {  // off: position inside the frame relative to z
            return pack_half2_clamped(*reinterpret_cast<const float *>(lf + (o0 + off * 4u)),
                                      *reinterpret_cast<const float *>(lf + (o1 + off * 4u)), clamp2);
        };

        unsigned AB[RL > 0 ? RL : 1], W[RL > 0 ? RL : 1], WX[RL > 0 ? RL : 1];
        static_for<0, MB>([&](auto br) {
            if constexpr (br < RL) { AB[br] = 0u; W[br] = 0u; WX[br] = 0u; }
            else if (lane_on) {
                lds_put<unsigned>(stA + LY::state_off(br - RL), 0u);
                lds_put<unsigned>(stA + LY::state_off(br - RL) + 2 * Z * 4, 0u);
            }
        });
        static_for<0, NB>([&](auto bc) {
            if constexpr (PL::shared(bc)) {
                const h2 l = load_pair(bc * Z);
                if (lane_on) lds_put<h2>(ln_addr<-1, LY, PL::lds_index(bc), 0>(zb, zbP, zbPW), l);  // idle lanes alias lane 0
                if constexpr (PL::lds_index(bc) % 8 == 7) __builtin_amdgcn_sched_barrier(0);  // eight columns in flight at a time
            }
        });
        if constexpr (!LY::SINGLE) {
            if (tid < 2 * F) { slot_bad[tid] = 0; if constexpr (FUSED) { slot_err[tid] = 0; slot_conv[tid] = 0; } }
            __syncthreads();
        } else if constexpr (FUSED) {
            if (tid < 2 * F) { slot_err[tid] = 0; slot_conv[tid] = 0; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }

        // wx: the second state word of a long row, and the word that holds bit 14 (the hard decision
        // of the row's degree-one variable) in either case
        auto load_state = [&](auto br, unsigned &ab, unsigned &w, unsigned &wx) {
            if constexpr (br < RL) { ab = AB[br]; w = W[br]; wx = HP::is_long(br) ? WX[br] : w; }
            else {
                ab = lds_get<unsigned>(stA + LY::state_off(br - RL));
                w = lds_get<unsigned>(stA + LY::state_off(br - RL) + 2 * Z * 4);
                wx = w;
            }
        };
        auto load_bits_word = [&](auto br) -> unsigned {
            if constexpr (br < RL) return HP::is_long(br) ? WX[br] : W[br];
            else return lds_get<unsigned>(stA + LY::state_off(br - RL) + 2 * Z * 4);
        };
        // index inside the frame, and the pair's channel values, of the degree-one variable of block row br
        auto priv_index = [&](auto br) -> int {
            constexpr int e = C::row_ptr[br] + PL::priv_pos(br), bc = C::row_col[e], s = C::row_shift[e];
            int zo = z;
            asm volatile("" : "+v"(zo));  // one address per block row, computed where it is used
            if constexpr (s == 0) return bc * Z + zo;
            else { const int v = zo + s; return bc * Z + (v >= Z ? v - Z : v); }
        };
        // the pair's channel values of the degree-one variable of block row br: scalar base address of the
        // group, 32-bit offsets per lane
        auto load_priv_raw = [&](auto br, float &a, float &b) {
            constexpr int e = C::row_ptr[br] + PL::priv_pos(br), bc = C::row_col[e], s = C::row_shift[e];
            unsigned cb = bc * Z * 4;
            asm volatile("" : "+s"(cb));  // formed where it is used: the optimiser would precompute and spill every row's
            if constexpr (s == 0) {
                a = *reinterpret_cast<const float *>(lf + (o0 + cb));
                b = *reinterpret_cast<const float *>(lf + (o1 + cb));
            } else {
                unsigned zz = (unsigned)z;
                asm volatile("" : "+v"(zz));  // computed where it is used
                const unsigned d = (zz + s >= (unsigned)Z ? (unsigned)(s - Z) : (unsigned)s) * 4u + cb;
                a = *reinterpret_cast<const float *>(lf + (o0 + d));
                b = *reinterpret_cast<const float *>(lf + (o1 + d));
            }
        };
        auto load_priv = [&](auto br) -> h2 {
            float a, b;
            load_priv_raw(br, a, b);
            return pack_half2_clamped(a, b, clamp2);
        };
        // Bit 14 of a row's state holds the hard decisions of its degree-one variable whenever a
        // syndrome test may follow; before the first sweep they are the channel's.
        if (early_exit) {
            static_for<0, MB>([&](auto br) {
                if constexpr (PL::is_priv_layer(br)) {
                    const unsigned hb = (load_priv(br) >> 1) & 0x40004000u;
                    if constexpr (br < RL) { if constexpr (HP::is_long(br)) WX[br] = hb; else W[br] = hb; }
                    else if (lane_on) lds_put<unsigned>(stA + LY::state_off(br - RL) + 2 * Z * 4, hb);
                    if constexpr (br % 8 == 7) __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
        // hard decisions, iteration count and flag of frame h (0 / 1) of the pair.  Addresses are
        // formed here, from a scalar base and a 32-bit offset the optimiser cannot move out of the
        // decode loop (dozens of precomputed 64-bit addresses would be spilled).
        uint8_t *const hard_grp = hard_out + grp * (2 * F) * n;
        auto write_half = [&](int h, long long frame, int iters, int conv) {
            const int sh = 15 + 16 * h;
            unsigned ho = (unsigned)(2 * f + h) * (unsigned)n + (unsigned)z;
            asm volatile("" : "+v"(ho));
            unsigned zero = 0;
            asm volatile("" : "+s"(zero));
            uint8_t *hg = hard_grp + zero;
            static_for<0, NB>([&](auto bc) {
                if constexpr (PL::cdeg(bc) != 1) {
                    h2 v;
                    if constexpr (PL::shared(bc)) v = lds_get<h2>(ln_addr<-1, LY, PL::lds_index(bc), 0>(zb, zbP, zbPW));
                    else v = load_pair(bc * Z);  // in no check at all: the channel decides
                    hg[ho + (unsigned)(bc * Z)] = (uint8_t)((v >> sh) & 1u);
                    if constexpr (bc % 4 == 3) __builtin_amdgcn_sched_barrier(0);  // a few columns in flight, not all: registers
                }
            });
            static_for<0, MB>([&](auto br) {
                if constexpr (PL::priv_pos(br) >= 0) {
                    unsigned hb;
                    if constexpr (PL::rdeg(br) >= 2) hb = (load_bits_word(br) >> (sh - 1)) & 1u;
                    else hb = (load_priv(br) >> sh) & 1u;
                    hg[ho + (unsigned)(priv_index(br) - z)] = (uint8_t)hb;
                    if constexpr (br % 4 == 3) __builtin_amdgcn_sched_barrier(0);
                }
            });
            if (z == 0) { iters_out[frame] = iters; conv_out[frame] = (uint8_t)conv; }
        };
        // fused: this lane's share of the bit errors of frame h against the transmitted word, taken where
        // write_half would store the decisions
        auto count_half = [&](int h, long long frame) __attribute__((always_inline)) -> unsigned {
            const int sh = 15 + 16 * h;
            // transmitted word: scalar base of the group and a 32-bit offset formed here, as in write_half
            // (addresses the optimiser could precompute outside the decode loop would be spilled)
            // The word is the copy the staging step put behind the group's LLRs: one straight run of code, no
            // branch per column.
            const uint8_t *rf = sweep_slice_ref<n, 2 * F>(reinterpret_cast<const float *>(lf));
            unsigned ro = (unsigned)(2 * f + h) * (unsigned)n + (unsigned)z;
            asm volatile("" : "+v"(ro));
            unsigned zero = 0;
            asm volatile("" : "+s"(zero));
            rf += zero;
            (void)frame;
            unsigned errs = 0;
            static_for<0, NB>([&](auto bc) {
                if constexpr (PL::cdeg(bc) != 1) {
                    h2 v;
                    if constexpr (PL::shared(bc)) v = lds_get<h2>(ln_addr<-1, LY, PL::lds_index(bc), 0>(zb, zbP, zbPW));
                    else v = load_pair(bc * Z);
                    errs += ((v >> sh) ^ (unsigned)rf[ro + (unsigned)(bc * Z)]) & 1u;
                    if constexpr (bc % 4 == 3) __builtin_amdgcn_sched_barrier(0);
                }
            });
            static_for<0, MB>([&](auto br) {
                if constexpr (PL::priv_pos(br) >= 0) {
                    unsigned hb;
                    if constexpr (PL::rdeg(br) >= 2) hb = (load_bits_word(br) >> (sh - 1)) & 1u;
                    else hb = (load_priv(br) >> sh) & 1u;
                    errs += (hb ^ (unsigned)rf[ro + (unsigned)(priv_index(br) - z)]) & 1u;
                    if constexpr (br % 4 == 3) __builtin_amdgcn_sched_barrier(0);
                }
            });
            return errs;
        };
        // (the closures must be inlined: one that is called would hold the check state arrays by address
        // and move them out of registers)
        auto count_frame = [&](int h, long long frame, int iters, int conv) __attribute__((always_inline)) {
            const unsigned e = count_half(h, frame);
            if (e) {
                atomicAdd(&wg_cnt[3], (unsigned long long)e);
                slot_err[2 * f + h] = 1;
            }
            if (z == 0) {
                atomicAdd(&wg_cnt[0], 1ull);
                if (!conv) atomicAdd(&wg_cnt[1], 1ull);
                atomicAdd(&wg_cnt[4], (unsigned long long)iters);
                slot_conv[2 * f + h] = conv;
            }
        };

        constexpr int NPL = PL::n_priv_layers();
        constexpr int SYN_FIRST = 2, SYN_EVERY = 6;  // block rows between looks at the outcome of the syndrome test
        // channel values of the degree-one variables of the coming layers, requested PF such layers ahead
        // (cyclically over the sweeps, hence a divisor of their number)
        constexpr int PF = NPL == 0 || NPL % LDPC_LNMSH_PREFETCH != 0 ? 1 : LDPC_LNMSH_PREFETCH;
        float lp_a[PF], lp_b[PF];
        static_for<0, (NPL == 0 ? 0 : PF)>([&](auto j) { load_priv_raw(IC<PL::priv_layer(j)>{}, lp_a[j], lp_b[j]); });

        bool done0 = !act0, done1 = !act1;
        int res_conv0 = 0, res_conv1 = 0;
        for (int t = 0;; ++t) {
            const bool last = t == max_iters;
            if (early_exit || last) {
                // ---- syndromes of the two frames: parities of the posteriors' sign bits
                if constexpr (!LY::SINGLE) __syncthreads();
                // no test after the last sweep.  The fused kernel runs it all the same and ignores the outcome:
                // the one place where frames end (and their errors are counted) sits behind it, and a second copy
                // of that code in the loop would push the sweep out of the instruction cache
                if constexpr (!FUSED) { if (last && early_exit) break; }
                const bool slot_done = done0 && done1;
                unsigned bad = 0;   // bit 15 / bit 31: a check of this lane fails in frame 0 / 1
                bool skip = false;  // the same in all lanes of a wave
                if (!slot_done || !LY::SINGLE) {
                    static_for<0, MB>([&](auto br) {
                        if constexpr (br == SYN_FIRST || (br > SYN_FIRST && (br - SYN_FIRST) % SYN_EVERY == 0)) {
                            // a failed test needs no more rows
                            const bool b0 = (bad & 0x8000u) != 0 && !done0, b1 = (bad & 0x80000000u) != 0 && !done1;
                            bool known0 = done0 || (__ballot(b0) & wave_slot_mask) != 0ull;
                            bool known1 = done1 || (__ballot(b1) & wave_slot_mask) != 0ull;
                            if constexpr (!LY::SINGLE) {
                                if (b0) slot_bad_now[2 * f] = 1;
                                if (b1) slot_bad_now[2 * f + 1] = 1;
                                known0 = known0 || slot_bad_now[2 * f] != 0;
                                known1 = known1 || slot_bad_now[2 * f + 1] != 0;
                            }
                            skip = __ballot(!(known0 && known1)) == 0ull;
                        }
                        if (skip || slot_done) return;
                        constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0, PP = PL::priv_pos(br);
                        unsigned zl = zb;
                        asm volatile("" : "+v"(zl));  // keep the rotated addresses out of loop-invariant hoisting
                        h2 v[D > 0 ? D : 1];
                        static_for<0, D>([&](auto p) {  // all reads of the row first, then their signs
                            if constexpr (p != PP) {
                                constexpr int bc = C::row_col[e0 + p], s = C::row_shift[e0 + p];
                                v[p] = lds_get<h2>(ln_addr<-1, LY, PL::lds_index(bc), s>(zl, zbP, zbPW));
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        });
                        unsigned par = 0;
                        static_for<0, D>([&](auto p) {
                            if constexpr (p != PP) par ^= v[p];
                            else if constexpr (D >= 2) par ^= load_bits_word(br) << 1;
                            else par ^= load_priv(br);  // a check of degree one
                        });
                        bad |= par;
                    });
                }
                const bool bad0 = (bad & 0x8000u) != 0 && !done0, bad1 = (bad & 0x80000000u) != 0 && !done1;
                bool ok0, ok1, any_bad = false;
                if constexpr (LY::SINGLE) {
                    ok0 = (__ballot(bad0) & slot_mask) == 0ull;
                    ok1 = (__ballot(bad1) & slot_mask) == 0ull;
                } else {
                    if (bad0) slot_bad[2 * f] = 1;
                    if (bad1) slot_bad[2 * f + 1] = 1;
                    __syncthreads();
                    ok0 = slot_bad[2 * f] == 0;
                    ok1 = slot_bad[2 * f + 1] == 0;
                    static_for<0, 2 * F>([&](auto k) { any_bad |= slot_bad[k] != 0; });
                }
                if constexpr (FUSED) {
                    // frames that end here: those that have just passed the test; after the last sweep every
                    // frame still running (converged = the final test in fixed-work mode, never with early exit)
                    const bool fin0 = last ? !done0 : (!done0 && ok0), fin1 = last ? !done1 : (!done1 && ok1);
                    const int fit = last ? (early_exit ? max_iters - 1 : max_iters) : t;
                    const int fc0 = last ? (early_exit ? 0 : (ok0 ? 1 : 0)) : 1, fc1 = last ? (early_exit ? 0 : (ok1 ? 1 : 0)) : 1;
                    // marked unlikely: the register allocator must take its registers from this block, not
                    // from the sweep (to it a loop inside the decode loop looks hotter than the sweep itself)
                    if (__builtin_expect(fin0 || fin1, 0)) {
#pragma unroll 1
                        for (int h = 0; h < 2; ++h) {
                            if (h ? fin1 : fin0) count_frame(h, h ? fr1 : fr0, fit, h ? fc1 : fc0);
                        }
                    }
                    done0 = done0 || fin0;
                    done1 = done1 || fin1;
                    if (last) break;
                } else {
                    if (last) {
                        res_conv0 = ok0 ? 1 : 0;
                        res_conv1 = ok1 ? 1 : 0;
                        break;
                    }
                    if (!done0 && ok0) { write_half(0, fr0, t, 1); done0 = true; }
                    if (!done1 && ok1) { write_half(1, fr1, t, 1); done1 = true; }
                }
                if constexpr (LY::SINGLE) {
                    if (__ballot(!(done0 && done1)) == 0ull) break;
                } else {
                    if (!any_bad) break;  // every frame that was still running has just converged
                    __syncthreads();
                    if (tid < 2 * F) slot_bad[tid] = 0;
                }
            }
            // ---- one sweep over the layers
            const bool slot_done = done0 && done1;
            static_for<0, MB>([&](auto br) {
                constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0, PP = PL::priv_pos(br);
                constexpr bool LONG = HP::is_long(br), TINY = HP::is_tiny(br);
                constexpr int DA = HP::da(br), DB = D - DA;
                if constexpr (D >= 2) {
                    if constexpr (!LY::SINGLE && PL::barrier_before(br)) __syncthreads();
                    h2 lp = 0u;
                    if constexpr (PP >= 0) {
                        constexpr int j = PL::priv_layer_ord(br);
                        lp = pack_half2_clamped(lp_a[j % PF], lp_b[j % PF], clamp2);
                        if constexpr (NPL > PF) load_priv_raw(IC<PL::priv_layer((j + PF) % NPL)>{}, lp_a[j % PF], lp_b[j % PF]);
                    }
                    if (!slot_done) {
                        h2 q[D], aq[LONG ? 1 : D];  // the inputs and (short rows) their magnitudes
                        unsigned off[D];
                        unsigned ab, w, wx;
                        load_state(br, ab, w, wx);
                        const h2 Ao = pk_shl<7>(ab) & kMagCut, Bo = pk_shr<1>(ab) & kMagCut;
                        unsigned zl = zb;
                        asm volatile("" : "+v"(zl));  // hundreds of hoisted addresses would spill
                        // all posteriors of the row are requested first: one LDS round trip per row
                        static_for<0, D>([&](auto p) {
                            if constexpr (p != PP) {
                                constexpr int bc = C::row_col[e0 + p], s = C::row_shift[e0 + p];
                                const unsigned ad = ln_addr<-1, LY, PL::lds_index(bc), s>(zl, zbP, zbPW);
                                off[LONG ? 0 : (int)p] = ad;
                                q[p] = lds_get<h2>(ad);
                                __builtin_amdgcn_sched_barrier(0);  // address, read, address, read
                            }
                        });
                        // position of the previous minimum, counted down edge by edge: zero at its own edge
                        // (tiny rows: one bit per edge, the current edge's on top)
                        unsigned cnt;
                        if constexpr (TINY) cnt = pk_shl<2>(w);
                        else cnt = pk_shr<10>(w) & (LONG ? 0x001f001fu : 0x000f000fu);
                        const unsigned dba_o = pk_sub_u16(Bo, Ao);
                        // bit 15 of a half of wsa / wsb: sign of the old message on the current edge.  Every edge
                        // shifts it out and the sign of its input in at the bottom: the words end as the new signs.
                        unsigned wsa = pk_shl<16 - DA>(w ^ pk_sign_fill(w));
                        unsigned wsb = 0;
                        if constexpr (LONG) wsb = pk_shl<16 - DB>(wx ^ pk_sign_fill(wx));
                        h2 m1, m2;
                        unsigned sx;  // sign parity of the inputs (bits 15 and 31)
                        static_for<0, D>([&](auto p) {
                            unsigned &ws = (p < DA) ? wsa : wsb;
                            if constexpr (p == PP) {
                                q[p] = lp;  // a variable of degree one always sends its channel value
                                hedge_in_given<p == 0, TINY>(q[p], aq[LONG ? 0 : (int)p], cnt, ws, m1, m2, sx, big2);
                            } else {
                                hedge_in<p == 0, TINY>(q[p], aq[LONG ? 0 : (int)p], cnt, Ao, dba_o, ws, m1, m2, sx, big2);
                            }
                        });
                        const unsigned px = sx & kSignMask;
                        // cut to eight magnitude bits, rounded to nearest (half a step added first)
                        const h2 An = (pk_min(pk_mul(m1, scale2), clamp2) + kMagHalfStep) & kMagCut;
                        const h2 Bn = (pk_min(pk_mul(m2, scale2), clamp2) + kMagHalfStep) & kMagCut;
                        const unsigned Anp = An | px;  // the parity of the inputs folded into the new messages
                        const unsigned dba_n = pk_sub_u16(Bn, An);
                        const unsigned m1p1 = m1 + 0x00010001u;  // bit patterns: no carry leaves a half
                        unsigned mina = 0, minb = 0;   // per edge: does it hold the smallest magnitude?
                        unsigned privbit = 0;
                        static_for<0, D>([&](auto p) {
                            unsigned &mn = (p < DA) ? mina : minb;
                            h2 pn;  // the variable's new posterior
                            if constexpr (LONG) pn = hedge_out_nomag(q[p], m1, Anp, dba_n, mn);
                            else pn = hedge_out(q[p], aq[p], m1p1, Anp, dba_n, mn);
                            if constexpr (p != PP) {
                                if constexpr (LONG) {  // a long row's addresses are formed again: keeping them costs registers
                                    constexpr int bc = C::row_col[e0 + p], s = C::row_shift[e0 + p];
                                    lds_put<h2>(ln_addr<-1, LY, PL::lds_index(bc), s>(zl, zbP, zbPW), pn);
                                } else {
                                    lds_put<h2>(off[p], pn);
                                }
                                __builtin_amdgcn_sched_barrier(0);  // block, write, block, write
                            } else {  // its hard decisions ride in the state word for the syndrome test and the write-out
                                privbit = (pn >> 1) & 0x40004000u;
                            }
                        });
                        unsigned wn;
                        if constexpr (TINY) {
                            // the bits as they are: edge p at bit 13 - p
                            wn = (mina << (14 - D)) | wsa | px;
                        } else {
                            // position of the first smallest input, per frame: edge p sits at bit D - 1 - p
                            unsigned c0, c1;
                            if constexpr (LONG) {
                                c0 = ((mina & 0xffffu) << DB) | (minb & 0xffffu);
                                c1 = ((mina >> 16) << DB) | (minb >> 16);
                            } else {
                                c0 = mina & 0xffffu;
                                c1 = mina >> 16;
                            }
                            const unsigned i0 = (unsigned)__builtin_clz(c0 | 1u) - (unsigned)(32 - D);
                            const unsigned i1 = (unsigned)__builtin_clz(c1 | 1u) - (unsigned)(32 - D);
                            wn = wsa | px | (i0 << 10) | (i1 << 26);
                        }
                        unsigned wxn = 0;
                        if constexpr (LONG) wxn = wsb | px | privbit;
                        else wn |= privbit;
                        const unsigned abn = pk_shr<7>(An) | (Bn << 1);
                        if constexpr (br < RL) {
                            AB[br] = abn; W[br] = wn;
                            if constexpr (LONG) WX[br] = wxn;
                        } else {
                            lds_put<unsigned>(stA + LY::state_off(br - RL), abn);
                            lds_put<unsigned>(stA + LY::state_off(br - RL) + 2 * Z * 4, wn);
                        }
                    }
                }
            });
        }
        // frames that never passed a test (or fixed work: every frame)
        const int res_iters = early_exit ? max_iters - 1 : max_iters;
        if constexpr (!FUSED) {
            if (act0 && !done0) write_half(0, fr0, res_iters, res_conv0);
            if (act1 && !done1) write_half(1, fr1, res_iters, res_conv1);
        } else {
            // converged with an error somewhere: known once every lane of the frame has counted
            if constexpr (LY::SINGLE) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            else __syncthreads();
            if (tid < 2 * F && slot_conv[tid] != 0 && slot_err[tid] != 0) atomicAdd(&wg_cnt[2], 1ull);
        }
        if constexpr (!LY::SINGLE) __syncthreads();
    }
    if constexpr (FUSED) {
        if constexpr (LY::SINGLE) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        else __syncthreads();
        if (tid < 5 && wg_cnt[tid] != 0ull) atomicAdd(&sp.counters[tid], wg_cnt[tid]);
    }
}

// ------------------------------------------------------------------------------------------ host
// How many block rows keep their state in registers: planned like the float kernel (a workgroup of
// four or more waves is the only one on its compute unit, smaller ones run three waves per SIMD),
// with about 53 registers left for everything that is not check state.
template <class C, int F>
struct LnHConfig {
    typedef LnHPlan<C> HP;
    static constexpr int WAVES = ((F * C::Z + 63) / 64);
    static constexpr int WPE = LDPC_LNMSH_WPE > 0 ? LDPC_LNMSH_WPE : WAVES >= 4 ? (WAVES + 3) / 4 : 3;
    static constexpr int VGPRS = WPE <= 2 ? 256 : (512 / WPE) / 8 * 8;
    static constexpr int STATE_REGS = VGPRS - LDPC_LNMSH_RESERVE > 180 ? 180 : VGPRS - LDPC_LNMSH_RESERVE;
    static constexpr int rl()
    {
        int r = 0;
        while (r < C::MB && HP::regs(r + 1) <= STATE_REGS) ++r;
        return r;
    }
    static constexpr int RL = rl();
    static constexpr bool OK = HP::supported() && HP::spill_ok(RL) && WPE <= 4 && F <= 8 &&
                               (LnLayout<C, float, F, RL>::LDS_BYTES <= 160 * 1024 || LDPC_LNMSH_RESERVE != 53);
};
// Slots per workgroup: the packaged default, doubled when that makes the number of waves a multiple
// of the four SIMDs (see decode_lnms.hip).
template <class C, int F>
struct LnHSlots {
    static constexpr int W1 = (F * C::Z + 63) / 64, W2 = (2 * F * C::Z + 63) / 64;
    static constexpr bool DOUBLE = LDPC_LNMSH_DOUBLE_SLOTS && W1 >= 4 && W1 % 4 != 0 && W2 % 4 == 0 && LnHConfig<C, 2 * F>::OK;
    static constexpr int value = DOUBLE ? 2 * F : F;
};

#ifdef LDPC_SPEC_FUSED  // second compilation of this file: the fused-sweep instantiations (see decode_spec.hip)
constexpr bool kFusedH = true;
#else
constexpr bool kFusedH = false;
#endif

template <class C, int F>
static hipError_t launch_lnms_h(const ldpc_decode_config &cfg, const Launch &io, int num_cus, const char **name,
                                const SweepDev *sp, Scratch *ws)
{
    typedef LnHConfig<C, F> CF;
    typedef LnLayout<C, float, F, CF::RL> LY;
    auto k = lnms_h_kernel<C, F, CF::RL, CF::WPE, kFusedH>;
    constexpr int LDS = LY::LDS_BYTES + (kFusedH ? 192 : 0);  // fused: two more rows of sixteen per-frame flags and five counters
    static int static_lds = -1;  // the kernel addresses LDS from 0: it must own the whole segment
    if (static_lds < 0) {
        hipFuncAttributes attr;
        hipError_t ea = hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(k));
        if (ea != hipSuccess) return ea;
        static_lds = (int)attr.sharedSizeBytes;
    }
    if (static_lds != 0) return hipErrorInvalidConfiguration;
    int per_cu = 0;
    hipError_t e = kernel_setup(reinterpret_cast<const void *>(k), LY::THREADS, LDS, &per_cu);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    if (getenv("LDPC_DEBUG_OCCUPANCY"))
        fprintf(stderr, "lnms_h_kernel<%s>: %d frame pairs per workgroup, %d block rows in registers, "
                "%d workgroups of %d threads per CU, %d B LDS each\n",
                C::name, F, CF::RL, per_cu, LY::THREADS, LDS);
    const long long pairs = (io.frames + 1) / 2, groups = (pairs + F - 1) / F;
    const long long full = (long long)num_cus * (long long)per_cu * 2;
    long long grid = full;
    if (grid > groups) grid = groups;
    if (grid < 1) grid = 1;
    float *scratch = nullptr;
    if (kFusedH) {  // one slice of 2 F frames' LLRs per workgroup
        e = scratch_reserve(*ws, sweep_slice_bytes<C::NB * C::Z, 2 * F>(), full, grid);
        if (e != hipSuccess) return e;
        scratch = reinterpret_cast<float *>(ws->ptr);
    }
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(LY::THREADS), LDS, io.stream, cfg.max_iters,
                       cfg.early_exit, (float)cfg.scale, reinterpret_cast<const float *>(io.llr), io.frames,
                       io.hard, io.iters, io.converged, sp ? *sp : SweepDev{}, scratch);
    *name = kFusedH ? "fused_spec_lnms_f16x2" : "spec_lnms_f16x2";
    return hipGetLastError();
}

// Layered min-sum in packed half precision on the compiled-in codes: float LLRs, no posteriors out.
// Compiled with -DLDPC_SPEC_FUSED the same dispatch serves ldpc_sweep_batch (channel, decode and error
// accounting in one launch; io.llr and the result arrays unused).
#ifdef LDPC_SPEC_FUSED
hipError_t launch_spec_lnms_half_fused(int spec_id, const ldpc_decode_config &cfg, const Launch &io, const SweepDev &sweep,
                                       int num_cus, Scratch &workspace, const char **name, bool *supported)
{
    const SweepDev *sp = &sweep;
    Scratch *ws = &workspace;
    *supported = false;
    if (cfg.rule != LDPC_RULE_LNMS) return hipSuccess;
#else
hipError_t launch_spec_lnms_half(int spec_id, const ldpc_decode_config &cfg, const Launch &io, int num_cus,
                                 const char **name, bool *supported)
{
    const SweepDev *sp = nullptr;
    Scratch *ws = nullptr;
    *supported = false;
    if (cfg.rule != LDPC_RULE_LNMS || io.llr_dtype != LDPC_F32 || io.posterior) return hipSuccess;
#endif
    int id = 0;
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) {                                                                           \
        if constexpr (LnHPlan<CODE>::supported()) {                                                \
            constexpr int FS = LnHSlots<CODE, SLOTS>::value;                                       \
            if constexpr (LnHConfig<CODE, FS>::OK) {                                               \
                *supported = true;                                                                 \
                return launch_lnms_h<CODE, FS>(cfg, io, num_cus, name, sp, ws);                    \
            }                                                                                      \
        }                                                                                          \
        return hipSuccess;                                                                         \
    }                                                                                              \
    ++id;
#ifdef LDPC_LNMS_DEV_ONLY  // development aid: instantiate one code only
    (void)id;
    if constexpr (LnHPlan<LDPC_LNMS_DEV_ONLY>::supported()) {
        constexpr int FS = LnHSlots<LDPC_LNMS_DEV_ONLY, 1>::value;
        if constexpr (LnHConfig<LDPC_LNMS_DEV_ONLY, FS>::OK) {
            *supported = true;
            return launch_lnms_h<LDPC_LNMS_DEV_ONLY, FS>(cfg, io, num_cus, name, sp, ws);
        }
    }
#else
    LDPC_FOR_EACH_SPEC_CODE(LDPC_LAUNCH)
#endif
#undef LDPC_LAUNCH
    return hipSuccess;
}

}  // namespace ldpc
