// Row-layered normalised min-sum (build extension, see LDPC_RULE_LNMS in ldpc_amd.h), specialised
// at compile time for the packaged circulant codes.
//
// One layer = one block row.  Thread (f, z) owns check (br, z) of every block row of frame slot f.
// What is kept where:
//   * posteriors P of the variables that take part in two or more checks: LDS, [column][Z];
//   * variables of degree one ("private" to their only check, e.g. the extension parity bits of the
//     NR base graphs): never stored.  Such a variable always sends its channel LLR, which is
//     fetched from the LLR array one layer ahead, and its posterior L + r is rebuilt from the
//     check's state when hard decisions are needed;
//   * the previous messages of a check, compressed to three words: A = scale * m1, B = scale * m2
//     (smallest / second smallest input magnitude) and W = position of the smallest input (bits
//     24-28) | sign parity of the inputs (bit 31) | hard decision of the private variable (bit
//     30) | sign of each input (low bits, first edge highest).  The first RL block rows keep
//     them in registers, the others in LDS.
// A layer reads P (rotated), subtracts the old message, updates (m1, m2, signs), adds the new
// message and writes P back to the same place.  All checks of a block row touch distinct
// variables (blocks that are sums of two circulants are not supported), so the block-parallel
// update equals a check-by-check sweep bit for bit.  Workgroups of more than one wave need a
// barrier only between layers that share a block column; consecutive layers on disjoint columns
// run back to back.
//
// NR BG1 Z = 384 in float: 39 KiB of posteriors + 40.5 KiB of check state in LDS and 111 state
// registers per thread, two workgroups (frames) per compute unit.
#include <cstdio>
#include <cstdlib>

#include "spec_common.h"
#include "lnms_plan.h"
#include "sweep_fused.h"

namespace ldpc {

// FUSED: the channel in front and the error accounting behind run inside the kernel
// (sweep_fused.h).  A whole frame of LLRs does not fit beside the decoder's state in LDS, so the
// workgroup writes the LLRs of its current frames to its own slice of `scratch` ([F][n] floats, in
// device memory but reused frame after frame, hence cache-resident) and decodes from there; llr and
// the result arrays are unused.
//
// WV >= 0: this copy of the code serves wave WV of every slot (lanes z = 64 WV ... 64 WV + 63): for most
// circulants the wrap of the rotated LDS address is then known at compile time (ln_addr).
template <class C, typename T, typename TL, int F, int RL, bool FUSED, int WV>
__device__ __forceinline__ void
lnms_body(const Cfg<T> &g, const TL *__restrict__ llr, long long frames, uint8_t *__restrict__ hard_out,
          int *__restrict__ iters_out, uint8_t *__restrict__ conv_out, T *__restrict__ post_out, const SweepDev &sp,
          float *scratch)
{
    typedef LnPlan<C> PL;
    typedef LnLayout<C, T, F, RL> LY;
    static_assert(PL::supported(), "code shape not supported by the layered kernel");
    static_assert(F <= 16, "the per-slot syndrome flags are one row of sixteen");
    static_assert(!FUSED || (F <= 8 && sizeof(TL) == 4 && (C::NB * C::Z) % 4 == 0), "fused: second flag row, float LLRs, four positions per draw");
    constexpr int Z = C::Z, NB = C::NB, MB = C::MB, ST = sizeof(T);
    constexpr int n = NB * Z;
    extern __shared__ __align__(16) unsigned char smem[];

    const int tid = threadIdx.x;
    const bool lane_on = tid < F * Z;
    const int f = (F == 1 || !lane_on) ? 0 : tid / Z;
    const int z = lane_on ? tid - f * Z : 0;
    const unsigned zb = (unsigned)z * ST;
    const unsigned zbP = (unsigned)f * LY::SLOT_BYTES + zb;          // absolute LDS addresses from here on
    const unsigned zbPW = zbP + LY::VEC;
    const unsigned stA = zbP;                                        // spilled check state of this thread
    const unsigned stW = (unsigned)f * LY::SLOT_BYTES + 2 * Z * ST + (unsigned)z * 4;
    int *slot_bad = reinterpret_cast<int *>(smem + LY::FLAG_OFF);
    const unsigned long long slot_mask = (Z >= 64 ? ~0ull : ((1ull << (Z & 63)) - 1ull)) << ((f * Z) & 63);
    // lanes of this wave that work on the same slot as this lane
    unsigned long long wave_slot_mask = slot_mask;
    if constexpr (!LY::SINGLE && F > 1) {
        static_for<0, F>([&](auto ff) {
            const unsigned long long m = __ballot(lane_on && f == ff);
            if (f == ff) wave_slot_mask = m;
        });
    } else if constexpr (!LY::SINGLE) {
        wave_slot_mask = ~0ull;
    }
    volatile int *slot_bad_now = slot_bad;  // polled while other waves may still be writing
    int *slot_err = slot_bad + 8;           // fused: the slot's frame has a bit error somewhere
    SweepAcc acc;

    const long long groups = (frames + F - 1) / F;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long frame = F == 1 ? grp : grp * F + f;
        const bool active = lane_on && frame < frames;
        // loads never depend on `active`: idle lanes read a frame that exists.  The group's first
        // frame is the same for every lane (a scalar base address), the rest is a 32-bit offset.
        const TL *lf;
        unsigned fo;
        const uint8_t *ref_slice = nullptr;  // fused: the frames' transmitted bits, [F][n] bytes
        if constexpr (FUSED) {
            // channel LLRs (and transmitted bits) of the group's frames into this workgroup's slice
            const float *my_scratch = sweep_stage_group<n, F>(sp, scratch, grp, frames);
            ref_slice = sweep_slice_ref<n, F>(my_scratch);
            lf = reinterpret_cast<const TL *>(my_scratch);
            fo = (unsigned)f * (unsigned)n + (unsigned)z;
        } else {
            lf = llr + grp * F * n;
            fo = (frame < frames ? (unsigned)f : 0u) * (unsigned)n + (unsigned)z;
        }

        T A[RL > 0 ? RL : 1], B[RL > 0 ? RL : 1];
        unsigned W[RL > 0 ? RL : 1];
        static_for<0, MB>([&](auto br) {
            if constexpr (br < RL) { A[br] = (T)0; B[br] = (T)0; W[br] = 0u; }
            else if (lane_on) {
                lds_put<T>(stA + LY::state_off(br - RL), (T)0);
                lds_put<T>(stA + LY::state_off(br - RL) + Z * ST, (T)0);
                lds_put<unsigned>(stW + LY::state_off(br - RL), 0u);
            }
        });
        static_for<0, NB>([&](auto bc) {
            if constexpr (PL::shared(bc)) {
                const T l = (T)lf[fo + (unsigned)(bc * Z)] + (T)0;  // -0.0 -> +0.0: "negative" is "sign bit set" from here on
                if (lane_on) lds_put<T>(ln_addr<WV, LY, PL::lds_index(bc), 0>(zb, zbP, zbPW), l);  // idle lanes alias lane 0
            }
        });
        if constexpr (!LY::SINGLE) {
            if (tid < F) { slot_bad[tid] = 0; if constexpr (FUSED) slot_err[tid] = 0; }
            __syncthreads();
        }

        auto load_state = [&](auto br, T &a, T &b, unsigned &w) {
            if constexpr (br < RL) { a = A[br]; b = B[br]; w = W[br]; }
            else {
                a = lds_get<T>(stA + LY::state_off(br - RL));
                b = lds_get<T>(stA + LY::state_off(br - RL) + Z * ST);
                w = lds_get<unsigned>(stW + LY::state_off(br - RL));
            }
        };
        // index inside the frame, and channel LLR, of the private variable of block row br
        auto priv_index = [&](auto br) -> int {
            constexpr int e = C::row_ptr[br] + PL::priv_pos(br), bc = C::row_col[e], s = C::row_shift[e];
            int zo = z;
            asm volatile("" : "+v"(zo));  // one address per block row, computed where it is used
            if constexpr (s == 0) return bc * Z + zo;
            else { const int v = zo + s; return bc * Z + (v >= Z ? v - Z : v); }
        };
        // scalar base address of the group, 32-bit byte offset per lane; the block column's share of the
        // offset is formed where it is used (hoisted, every row's address would be precomputed and spilled)
        const unsigned fob = fo * (unsigned)sizeof(TL);
        auto load_priv = [&](auto br) -> T {
            constexpr int e = C::row_ptr[br] + PL::priv_pos(br), bc = C::row_col[e], s = C::row_shift[e];
            unsigned cb = bc * Z * (int)sizeof(TL);
            asm volatile("" : "+s"(cb));
            unsigned o = fob + cb;
            if constexpr (s != 0) {
                unsigned zz = (unsigned)z;
                asm volatile("" : "+v"(zz));
                o += (zz + s >= (unsigned)Z ? (unsigned)(s - Z) : (unsigned)s) * (unsigned)sizeof(TL);
            }
            return (T)*reinterpret_cast<const TL *>(reinterpret_cast<const char *>(lf) + o) + (T)0;
        };
        // Bit 30 of W holds the hard decision of the layer's private variable whenever a syndrome
        // test may follow (early exit: always; else after the last sweep), so that the test needs
        // no channel LLRs.  Before the first sweep that decision is the channel's.
        if (g.early_exit) {
            static_for<0, MB>([&](auto br) {
                if constexpr (PL::is_priv_layer(br)) {
                    const unsigned hb = hard_bit<1>(load_priv(br)) << 30;
                    if constexpr (br < RL) W[br] = hb;
                    else if (lane_on) lds_put<unsigned>(stW + LY::state_off(br - RL), hb);
                }
            });
        }
        // posterior L + r of that variable, r rebuilt from the check's state
        auto priv_value = [&](auto br, T l) -> T {
            constexpr int D = PL::rdeg(br), p = PL::priv_pos(br);
            if constexpr (D < 2) return l;
            else {
                T a, b;
                unsigned w;
                load_state(br, a, b, w);
                const T mag = (int)((w >> 24) & 31u) == p ? b : a;
                return l + SNum<T>::flip_sign(mag, ((w << (31 - (D - 1 - p))) ^ w) & 0x80000000u);
            }
        };
        // channel LLRs of private variables are requested PFR block rows ahead of their use
        constexpr int NPR = PL::n_priv_rows(), PFR = 4;
        auto write_out = [&](int iters, int conv) {
            static_for<0, NB>([&](auto bc) {
                if constexpr (PL::cdeg(bc) != 1) {
                    T v;
                    if constexpr (PL::shared(bc)) v = lds_get<T>(ln_addr<WV, LY, PL::lds_index(bc), 0>(zb, zbP, zbPW));
                    else v = (T)lf[fo + (unsigned)(bc * Z)] + (T)0;  // in no check at all: the channel decides
                    hard_out[frame * n + bc * Z + z] = (uint8_t)hard_bit<1>(v);
                    if (post_out) post_out[frame * n + bc * Z + z] = v;
                }
            });
            if (post_out) {  // posteriors wanted: L + r needs the channel LLRs again
                T ring[PFR];
                static_for<0, (PFR < NPR ? PFR : NPR)>([&](auto j) { ring[j] = load_priv(IC<PL::priv_row(j)>{}); });
                static_for<0, MB>([&](auto br) {
                    if constexpr (PL::priv_pos(br) >= 0) {
                        constexpr int j = PL::priv_row_ord(br);
                        const T v = priv_value(br, ring[j % PFR]);
                        if constexpr (j + PFR < NPR) ring[j % PFR] = load_priv(IC<PL::priv_row(j + PFR < NPR ? j + PFR : 0)>{});
                        const int i = priv_index(br);
                        hard_out[frame * n + i] = (uint8_t)hard_bit<1>(v);
                        post_out[frame * n + i] = v;
                    }
                });
            } else {         // hard decisions only: they sit in bit 30 of the state words
                static_for<0, MB>([&](auto br) {
                    if constexpr (PL::priv_pos(br) >= 0) {
                        unsigned hb;
                        if constexpr (PL::rdeg(br) >= 2) {
                            T a, b;
                            unsigned w;
                            load_state(br, a, b, w);
                            hb = (w >> 30) & 1u;
                        } else {
                            hb = hard_bit<1>(load_priv(br));
                        }
                        hard_out[frame * n + priv_index(br)] = (uint8_t)hb;
                    }
                });
            }
            if (z == 0) { iters_out[frame] = iters; conv_out[frame] = (uint8_t)conv; }
        };

        // fused: this lane's share of the frame's bit errors against the transmitted word
        auto count_errs = [&]() -> unsigned {
            const uint8_t *rf = ref_slice + (unsigned)f * (unsigned)n;
            unsigned errs = 0;
            static_for<0, NB>([&](auto bc) {
                if constexpr (PL::cdeg(bc) != 1) {
                    T v;
                    if constexpr (PL::shared(bc)) v = lds_get<T>(ln_addr<WV, LY, PL::lds_index(bc), 0>(zb, zbP, zbPW));
                    else v = (T)lf[fo + (unsigned)(bc * Z)] + (T)0;
                    errs += hard_bit<1>(v) != (unsigned)rf[bc * Z + z];
                }
            });
            static_for<0, MB>([&](auto br) {
                if constexpr (PL::priv_pos(br) >= 0) {
                    unsigned hb;
                    if constexpr (PL::rdeg(br) >= 2) {
                        T a, b;
                        unsigned w;
                        load_state(br, a, b, w);
                        hb = (w >> 30) & 1u;
                    } else {
                        hb = hard_bit<1>(load_priv(br));
                    }
                    errs += hb != (unsigned)rf[priv_index(br)];
                }
            });
            return errs;
        };

        // channel LLRs of the private variables the sweep needs next, requested PFS layers ahead
        // (cyclically over the sweeps, hence a divisor of the number of such layers).  One layer
        // ahead measured fastest: deeper queues cost registers the layers need themselves.
        constexpr int NPL = PL::n_priv_layers();
        constexpr int SYN_FIRST = 2, SYN_EVERY = 6;  // block rows between looks at the outcome of the syndrome test
#ifndef LDPC_LNMS_PFS
#define LDPC_LNMS_PFS 1
#endif
        constexpr int PFS = NPL == 0 || NPL % LDPC_LNMS_PFS != 0 ? 1 : LDPC_LNMS_PFS;
        T lp_ring[PFS];
        static_for<0, (NPL == 0 ? 0 : PFS)>([&](auto j) { lp_ring[j] = load_priv(IC<PL::priv_layer(j)>{}); });

        // A slot that has converged is left alone (its posteriors and check state stay as they were
        // at the successful test) and everything is written once, after the loop.
        bool done = !active;
        int res_iters = g.early_exit ? g.max_iters - 1 : g.max_iters, res_conv = 0;
        for (int t = 0;; ++t) {
            const bool last = t == g.max_iters;
            if (g.early_exit || last) {
                // ---- syndrome of the hard decisions `0 if P > 0 else 1` of the current posteriors
                if constexpr (!LY::SINGLE) __syncthreads();
                if (last && g.early_exit) break;  // no test after the last sweep
                // A failed test needs no more rows: every few block rows a wave stops if each of its
                // slots is known to have failed, from its own lanes or from the flag another wave set.
                unsigned bad = 0;
                bool skip = false;  // the same in all lanes of a wave
                if (!done || !LY::SINGLE) {
                    static_for<0, MB>([&](auto br) {
                        if constexpr (br == SYN_FIRST || (br > SYN_FIRST && (br - SYN_FIRST) % SYN_EVERY == 0)) {
                            bool known = done || (__ballot(bad != 0) & wave_slot_mask) != 0ull;
                            if constexpr (!LY::SINGLE) {
                                if (bad) slot_bad_now[f] = 1;
                                known = known || slot_bad_now[f] != 0;
                            }
                            skip = __ballot(!known) == 0ull;
                        }
                        if (skip || done) return;
                        constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0, PP = PL::priv_pos(br);
                        unsigned zl = zb;
                        asm volatile("" : "+v"(zl));  // keep the rotated addresses out of loop-invariant hoisting
                        T v[D > 0 ? D : 1];
                        static_for<0, D>([&](auto p) {  // all reads of the row first, then their signs
                            if constexpr (p != PP) {
                                constexpr int bc = C::row_col[e0 + p], s = C::row_shift[e0 + p];
                                v[p] = lds_get<T>(ln_addr<WV, LY, PL::lds_index(bc), s>(zl, zbP, zbPW));
                                if constexpr (ST == 4) __builtin_amdgcn_sched_barrier(0);  // as in the sweep below
                            }
                        });
                        unsigned par = 0;
                        static_for<0, D>([&](auto p) {
                            if constexpr (p != PP) par ^= hard_bit<1>(v[p]);
                            else if constexpr (D >= 2) {
                                T a, b;
                                unsigned w;
                                load_state(br, a, b, w);
                                par ^= (w >> 30) & 1u;
                            } else {
                                par ^= hard_bit<1>(load_priv(br));  // a check of degree one
                            }
                        });
                        bad |= par;
                    });
                }
                bool slot_ok, any_bad = false;
                if constexpr (LY::SINGLE) {
                    slot_ok = (__ballot(bad != 0) & slot_mask) == 0ull;
                } else {
                    if (bad) slot_bad[f] = 1;
                    __syncthreads();
                    slot_ok = slot_bad[f] == 0;
                    static_for<0, F>([&](auto ff) { any_bad |= slot_bad[ff] != 0; });
                }
                if (last) {
                    if (!done) res_conv = slot_ok ? 1 : 0;
                    break;
                }
                if (!done && slot_ok) { res_iters = t; res_conv = 1; done = true; }
                if constexpr (LY::SINGLE) {
                    if (__ballot(!done) == 0ull) break;
                } else {
                    if (!any_bad) break;  // every slot that was still running has just converged
                    __syncthreads();
                    if (tid < F) slot_bad[tid] = 0;
                }
            }
            // ---- one sweep over the layers
            const bool want_bits = g.early_exit || t == g.max_iters - 1;
            static_for<0, MB>([&](auto br) {
                constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0, PP = PL::priv_pos(br);
                if constexpr (D >= 2) {
                    if constexpr (!LY::SINGLE && PL::barrier_before(br)) __syncthreads();
                    T lp = (T)0;
                    if constexpr (PP >= 0) {
                        constexpr int j = PL::priv_layer_ord(br);
                        lp = lp_ring[j % PFS];
                        if constexpr (NPL > PFS) lp_ring[j % PFS] = load_priv(IC<PL::priv_layer((j + PFS) % NPL)>{});
                    }
                    if (!done) {
                        T q[D];
                        unsigned off[D];
                        T Ao, Bo;
                        unsigned w;
                        load_state(br, Ao, Bo, w);
                        const int idx_old = (int)((w >> 24) & 31u);
                        const unsigned wp = w ^ (unsigned)((int)w >> 31);  // bit D-1-p: sign of the old message on edge p
                        T m1, m2;
                        unsigned zl = zb;
                        asm volatile("" : "+v"(zl));  // as above: hundreds of hoisted addresses would spill
                        // all posteriors of the row are requested first, so that the row pays one LDS round
                        // trip instead of one per edge
                        static_for<0, D>([&](auto p) {
                            if constexpr (p == PP) {
                                off[p] = 0;
                            } else {
                                constexpr int bc = C::row_col[e0 + p], s = C::row_shift[e0 + p];
                                off[p] = ln_addr<WV, LY, PL::lds_index(bc), s>(zl, zbP, zbPW);
                                q[p] = lds_get<T>(off[p]);
                                if constexpr (ST == 4) __builtin_amdgcn_sched_barrier(0);  // address, read, address, read: two address blocks in a row cost a wait state
                            }
                        });
                        // bit 31 of ws: sign of the old message on the current edge; every edge shifts it out
                        // and the sign of its input in at the other end, so the word ends as the new signs
                        unsigned ws = wp << (32 - D);
                        static_for<0, D>([&](auto p) {
                            if constexpr (p == PP) {
                                q[p] = lp;  // a variable of degree one always sends its channel LLR
                                SNum<T>::template edge_in_given<p == 0>(q[p], ws, m1, m2, Lim<T>::inf());
                            } else {
                                SNum<T>::template edge_in<p, p == 0>(q[p], (unsigned)idx_old, Ao, Bo, ws, m1, m2, Lim<T>::inf());
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
                            const T pn = SNum<T>::pick_add(q[p], m1, Anp, Bnp, minbits);  // the variable's new posterior
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
                            lds_put<T>(stA + LY::state_off(br - RL), An);
                            lds_put<T>(stA + LY::state_off(br - RL) + Z * ST, Bn);
                            lds_put<unsigned>(stW + LY::state_off(br - RL), wn);
                        }
                    }
                }
            });
        }
        if constexpr (FUSED) {
            const unsigned errs = active ? count_errs() : 0u;
            bool any_err;
            if constexpr (LY::SINGLE) {
                any_err = (__ballot(errs != 0) & slot_mask) != 0ull;
            } else {
                if (errs) slot_err[f] = 1;
                __syncthreads();
                any_err = slot_err[f] != 0;
            }
            if (active) acc.book(errs, any_err, z == 0, res_iters, res_conv);
        } else {
            if (active) write_out(res_iters, res_conv);
        }
        if constexpr (!LY::SINGLE) __syncthreads();
    }
    if constexpr (FUSED) acc.flush_wave(sp.counters);
}

// One copy of the code per wave of a slot where a slot is a whole number of waves (NR BG1 Z = 384: six)
// and the workgroup has several; every wave executes the same barriers in the same order in its own copy.
// A switch for kernel experiments, off by default: measured on BG1 in float it removes 9.4 % of the
// vector instructions (28 600 instead of 31 600 per frame-sweep) and nothing of the run time (the
// waiting share of the wave cycles grows from 37 % to 42 %): six copies are about 230 KB of code
// streamed once per sweep, against a 64 KB instruction cache that the single copy fits.
#ifndef LDPC_LNMS_WAVE_COPIES
#define LDPC_LNMS_WAVE_COPIES 0
#endif
template <class C, typename T, int F, bool FUSED>
constexpr int lnms_wave_copies()
{
    return (LDPC_LNMS_WAVE_COPIES && !FUSED && sizeof(T) == 4 && C::Z % 64 == 0 && C::Z / 64 >= 2 && C::Z / 64 <= 8) ? C::Z / 64 : 0;
}

template <class C, typename T, typename TL, int F, int RL, int WPE, bool FUSED = false>
__global__ void __launch_bounds__((LnLayout<C, T, F, RL>::THREADS)) __attribute__((amdgpu_waves_per_eu(WPE, 8)))
lnms_kernel(Cfg<T> g, const TL *__restrict__ llr, long long frames, uint8_t *__restrict__ hard_out,
            int *__restrict__ iters_out, uint8_t *__restrict__ conv_out, T *__restrict__ post_out, SweepDev sp,
            float *scratch)
{
    constexpr int NW = lnms_wave_copies<C, T, F, FUSED>();
    if constexpr (NW == 0) {
        lnms_body<C, T, TL, F, RL, FUSED, -1>(g, llr, frames, hard_out, iters_out, conv_out, post_out, sp, scratch);
    } else {
        const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) % NW;  // wave-uniform: a scalar branch
        static_for<0, NW>([&](auto w) {
            if (wv == w) lnms_body<C, T, TL, F, RL, FUSED, w>(g, llr, frames, hard_out, iters_out, conv_out, post_out, sp, scratch);
        });
    }
}

// ------------------------------------------------------------------------------------------ host
// How many block rows keep their state in registers, and the occupancy the register allocator is
// asked for.  A workgroup of four or more waves is planned as the only one on its compute unit
// (its waves per SIMD fix the register budget); smaller workgroups are planned for three waves
// per SIMD.  About 57 registers are left for everything that is not check state.
template <class C, typename T, int F>
struct LnConfig {
    static constexpr int RPL = 2 * (int)sizeof(T) / 4 + 1;  // registers per block row
    static constexpr int WAVES = ((F * C::Z + 63) / 64);
    static constexpr int WPE = WAVES >= 4 ? (WAVES + 3) / 4 : 3;
    static constexpr int VGPRS = WPE <= 2 ? 256 : (512 / WPE) / 8 * 8;
    static constexpr int STATE_REGS = VGPRS - 57 > 180 ? 180 : VGPRS - 57;
    static constexpr int RL = STATE_REGS <= 0 ? 0 : (C::MB < STATE_REGS / RPL ? C::MB : STATE_REGS / RPL);
    static constexpr bool OK = WPE <= 4 && LnLayout<C, T, F, RL>::LDS_BYTES <= 160 * 1024;
};
// Frames per workgroup: the packaged default, doubled when that makes the number of waves a
// multiple of the four SIMDs (waves of one workgroup are dealt round-robin to the SIMDs, and six
// waves would leave two SIMDs with twice the work of the others).
template <class C, typename T, int F>
struct LnSlots {
    static constexpr int W1 = (F * C::Z + 63) / 64, W2 = (2 * F * C::Z + 63) / 64;
    static constexpr bool DOUBLE = W1 >= 4 && W1 % 4 != 0 && W2 % 4 == 0 && LnConfig<C, T, 2 * F>::OK;
    static constexpr int value = DOUBLE ? 2 * F : F;
};

#ifdef LDPC_SPEC_FUSED  // second compilation of this file: the fused-sweep instantiations (see decode_spec.hip)
constexpr bool kFused = true;
#else
constexpr bool kFused = false;
#endif

template <class C, typename T, typename TL, int F>
static hipError_t launch_lnms(const ldpc_decode_config &cfg, const Launch &io, int num_cus, const char **name,
                              const SweepDev *sp, Scratch *ws)
{
    typedef LnConfig<C, T, F> CF;
    typedef LnLayout<C, T, F, CF::RL> LY;
    Cfg<T> g;
    g.rule = LDPC_RULE_LNMS; g.max_iters = cfg.max_iters; g.early_exit = cfg.early_exit;
    g.float_syndrome0 = 0; g.num_active_checks = -1; g.num_refresh_vars = -1;
    g.scale = (T)cfg.scale; g.clip_tanh_arg = (T)0; g.clip_prod = (T)0; g.clip_post = (T)0; g.offset = (T)0;
    auto k = lnms_kernel<C, T, TL, F, CF::RL, CF::WPE, kFused>;
    static int static_lds = -1;  // the kernel addresses LDS from 0: it must own the whole segment
    if (static_lds < 0) {
        hipFuncAttributes attr;
        hipError_t ea = hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(k));
        if (ea != hipSuccess) return ea;
        static_lds = (int)attr.sharedSizeBytes;
    }
    if (static_lds != 0) return hipErrorInvalidConfiguration;
    int per_cu = 0;  // resident workgroups per compute unit, as the runtime computes it
    hipError_t e = kernel_setup(reinterpret_cast<const void *>(k), LY::THREADS, LY::LDS_BYTES, &per_cu);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    if (getenv("LDPC_DEBUG_OCCUPANCY"))
        fprintf(stderr, "lnms_kernel<%s,%d-byte>: %d frames per workgroup, %d block rows in registers, "
                "%d workgroups of %d threads per CU, %d B LDS each\n",
                C::name, (int)sizeof(T), F, CF::RL, per_cu, LY::THREADS, LY::LDS_BYTES);
    const long long groups = (io.frames + F - 1) / F;
    const long long full = (long long)num_cus * (long long)per_cu * 2;
    long long grid = full;
    if (grid > groups) grid = groups;
    if (grid < 1) grid = 1;
    float *scratch = nullptr;
    if (kFused) {  // one slice of F frames' LLRs per workgroup
        e = scratch_reserve(*ws, sweep_slice_bytes<C::NB * C::Z, F>(), full, grid);
        if (e != hipSuccess) return e;
        scratch = reinterpret_cast<float *>(ws->ptr);
    }
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(LY::THREADS), LY::LDS_BYTES, io.stream, g,
                       reinterpret_cast<const TL *>(io.llr), io.frames, io.hard, io.iters, io.converged,
                       reinterpret_cast<T *>(io.posterior), sp ? *sp : SweepDev{}, scratch);
    *name = kFused ? (sizeof(T) == 8 ? "fused_spec_lnms_f64" : "fused_spec_lnms_f32") : (sizeof(T) == 8 ? "spec_lnms_f64" : "spec_lnms_f32");
    return hipGetLastError();
}

template <class C, int F>
static hipError_t launch_lnms_code(const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                                   int num_cus, const char **name, bool *supported, const SweepDev *sp, Scratch *ws)
{
    *supported = true;
    if constexpr (LnPlan<C>::supported() && (!kFused || (C::NB * C::Z) % 4 == 0)) {
        constexpr int FD = LnSlots<C, double, F>::value, FS = LnSlots<C, float, F>::value;
        if constexpr (LnConfig<C, double, FD>::OK && (!kFused || FD <= 8)) {
            if (compute_dtype == LDPC_F64) {
                if constexpr (!kFused) {
                    if (io.llr_dtype == LDPC_F64) return launch_lnms<C, double, double, FD>(cfg, io, num_cus, name, sp, ws);
                }
                return launch_lnms<C, double, float, FD>(cfg, io, num_cus, name, sp, ws);
            }
        }
        if constexpr (LnConfig<C, float, FS>::OK && (!kFused || FS <= 8)) {
            if (compute_dtype == LDPC_F32 && (kFused || io.llr_dtype == LDPC_F32))
                return launch_lnms<C, float, float, FS>(cfg, io, num_cus, name, sp, ws);
        }
    }
    *supported = false;
    return hipSuccess;
}

#ifdef LDPC_SPEC_FUSED
hipError_t launch_spec_lnms_fused(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                                  const SweepDev &sp, int num_cus, Scratch &workspace, const char **name, bool *supported)
{
    int id = 0;
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) return launch_lnms_code<CODE, SLOTS>(cfg, compute_dtype, io, num_cus, name, supported, &sp, &workspace); \
    ++id;
    LDPC_FOR_EACH_SPEC_CODE(LDPC_LAUNCH)
#undef LDPC_LAUNCH
    *supported = false;
    return hipSuccess;
}
#else
hipError_t launch_spec_lnms(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                            int num_cus, const char **name, bool *supported)
{
#ifdef LDPC_LNMS_DEV_ONLY  // development aid: instantiate one code only
    return launch_lnms_code<LDPC_LNMS_DEV_ONLY, 1>(cfg, compute_dtype, io, num_cus, name, supported, nullptr, nullptr);
#else
    int id = 0;
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) return launch_lnms_code<CODE, SLOTS>(cfg, compute_dtype, io, num_cus, name, supported, nullptr, nullptr); \
    ++id;
    LDPC_FOR_EACH_SPEC_CODE(LDPC_LAUNCH)
#undef LDPC_LAUNCH
    *supported = false;
    return hipSuccess;
#endif
}
#endif

}  // namespace ldpc
