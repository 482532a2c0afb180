// Code-specialised circulant decoders: the base graph is a template parameter.
//
// `Code` (csrc/gen/codes_gen.h, generated from the packaged matrices) carries the circulant tables
// as constant expressions, so the loops over block rows / block columns / edges are unrolled at
// compile time: every shift and array offset is an immediate (rotated LDS addresses cost one or
// two vector instructions instead of five), the per-variable state (channel LLR, own posterior)
// sits in registers, and no table is read at run time.  A handle whose circulant description
// equals a compiled-in code is routed here; any other code runs on the table-driven kernels.
//
// Thread mapping is that of the table-driven kernels: F frames ("slots") per workgroup, thread
// tid = f*Z + z.  When the workgroup is a single wave (F*Z <= 64) no barrier is needed at all:
// LDS operations of one wave execute in order, and per-slot decisions come from a ballot.
//
// Rule R0 (ldpc.py:137-227), compressed check state: the check-node phase stores per check
// (m1, m2) = smallest and second smallest input magnitude, both carrying the parity of the input
// signs in their sign bit; the variable-node phase rebuilds each message
//      |r| = m2 if |b_i| == m1 else m1,   sign(r) = parity xor sign(b_i)      (ldpc.py:184-187)
// and sums in ascending check order (ldpc.py:173,188), then b = 0.75 * (S + L) (ldpc.py:196).
#include <cstring>

#include "spec_common.h"
#include "sweep_fused.h"

namespace ldpc {

template <class C, typename T, int F>
struct SpecLayout {
    static constexpr int THREADS = ((F * C::Z + 63) / 64) * 64;
    static constexpr bool SINGLE = THREADS == 64;
    static constexpr int POST_BYTES = C::NB * C::Z * (int)sizeof(T);
    static constexpr int CS_BYTES = C::MB * C::Z * 2 * (int)sizeof(T);
    // slots are padded so that consecutive slots start 8 banks apart
    static constexpr int SLOT_BYTES = ((POST_BYTES + CS_BYTES + 127) / 128) * 128 + (F > 1 ? 32 : 0);
    static constexpr int FLAG_OFF = F * SLOT_BYTES;
    static constexpr int LDS_BYTES = FLAG_OFF + 16 * (int)sizeof(int);
};

// ------------------------------------------------------------------------------------------ R0
// check-node phase of one thread: every block row in turn.  Returns 1 if any of its checks fails
// (FLOATCHK: the iteration-0 test of ldpc.py:158,165 on the raw LLR sums instead of the parity).
//
// Edges of shift 0 connect check (br, z) with variable (bc, z): both belong to the same lane, so
// where the check states of a lane fit registers (LOCAL, base graphs of at most four block rows)
// such an edge's posterior comes from the lane's own register and its check state goes back the
// same way (`cl`), and only the rotated edges go through LDS.  The values are the same either way.
// ACT: only the checks below `mact` take part in the update (int(self.m) of the constructor, ldpc.py:30,175:
// ccsds.py:18 builds LDPCEncoder(4, 8, 16), i.e. 8 rows); the others send zeros.  The syndrome uses every row.
template <class C, typename T, bool FLOATCHK, bool ACT = false>
__device__ __forceinline__ int spec_r0_checks(const unsigned char *post_b, unsigned char *cs_b, unsigned zb, unsigned zp,
                                              const T (&own)[C::NB], typename SNum<T>::Pair (&cl)[spec_local_edges<C>() ? C::MB : 1],
                                              int z = 0, int mact = 0)
{
    typedef typename SNum<T>::Pair Pair;
    constexpr int Z = C::Z, MB = C::MB, ST = sizeof(T), SP = 2 * sizeof(T);
    constexpr bool LOCAL = spec_local_edges<C>();
    int bad = 0;
    static_for<0, MB>([&](auto br) {
        constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0;
        T m1 = Lim<T>::inf(), m2 = Lim<T>::inf();
        unsigned sx = 0;
        T fsum = (T)0;
        static_for<0, D>([&](auto p) {
            constexpr int bc = C::row_col[e0 + p], s = C::row_shift[e0 + p];
            T b;
            if constexpr (LOCAL && s == 0) b = own[bc];
            else b = *reinterpret_cast<const T *>(post_b + bc * Z * ST + rot_fwd<Z, ST, s>(zb));
            SNum<T>::template book<p>(m1, m2, b);
            sx ^= SNum<T>::sign_word(b);
            if constexpr (FLOATCHK) fsum += b;
        });
        if constexpr (D < 2) { m1 = (T)0; m2 = (T)0; }  // ldpc.py:178
        if constexpr (ACT) {
            if (br * Z + z >= mact) { m1 = (T)0; m2 = (T)0; }
        }
        const Pair cst = SNum<T>::make(SNum<T>::with_sign(m1, sx), SNum<T>::with_sign(m2, sx));
        *reinterpret_cast<Pair *>(cs_b + br * Z * SP + zp) = cst;
        if constexpr (LOCAL) cl[br] = cst;
        if constexpr (FLOATCHK) bad |= !even_integer(fsum);
        else bad |= (int)(((unsigned)D ^ (sx >> 31)) & 1u);  // parity of the hard bits 1 - signbit
    });
    return bad;
}

#ifndef LDPC_SPEC_WAVES
#define LDPC_SPEC_WAVES 0
#endif
#if LDPC_SPEC_WAVES > 0
#define LDPC_SPEC_OCC __attribute__((amdgpu_waves_per_eu(LDPC_SPEC_WAVES, LDPC_SPEC_WAVES)))
#else
#define LDPC_SPEC_OCC
#endif

// FUSED: the channel in front and the error accounting behind run inside the kernel (sweep_fused.h);
// llr and the result arrays are unused, `sp` describes the channel and the counters.
// Registers: for the small base graphs (CCSDS) at least three waves per SIMD in double and four in
// float, as the single-wave workgroups are sized for (a few registers more would silently cost a
// wave per SIMD); larger graphs take what they need.
template <class C, typename T, typename TL, int F, bool FUSED = false, bool ACT = false>
__global__ void __launch_bounds__((SpecLayout<C, T, F>::THREADS))
#if LDPC_SPEC_WAVES > 0
LDPC_SPEC_OCC
#else
__attribute__((amdgpu_waves_per_eu((SpecLayout<C, T, F>::SINGLE && C::EB <= 40 ? (sizeof(T) == 8 ? 3 : 4) : 1), 8)))
#endif
spec_r0_kernel(Cfg<T> g, const TL *__restrict__ llr, long long frames, uint8_t *__restrict__ hard_out,
               int *__restrict__ iters_out, uint8_t *__restrict__ conv_out, T *__restrict__ post_out, SweepDev sp)
{
    typedef SpecLayout<C, T, F> LY;
    typedef typename SNum<T>::Pair Pair;
    constexpr int Z = C::Z, NB = C::NB, MB = C::MB, ST = sizeof(T), SP = 2 * sizeof(T);
    constexpr int n = NB * Z;
    constexpr bool LOCAL = spec_local_edges<C>();
#ifdef LDPC_NO_LOCAL_SWAP
    constexpr bool LOCAL_SWAP = false;
#else
    constexpr bool LOCAL_SWAP = LOCAL;
#endif
    extern __shared__ __align__(16) unsigned char smem[];

    const int tid = threadIdx.x;
    const bool lane_on = tid < F * Z;
    const int f = lane_on ? tid / Z : 0;
    const int z = lane_on ? tid - f * Z : 0;
    unsigned char *slot = smem + f * LY::SLOT_BYTES;
    unsigned char *post_b = slot;                  // [NB][Z] T
    unsigned char *cs_b = slot + LY::POST_BYTES;   // [MB][Z] Pair
    int *slot_bad = reinterpret_cast<int *>(smem + LY::FLAG_OFF);
    const unsigned zb = (unsigned)z * ST, zp = (unsigned)z * SP;
    unsigned long long slot_mask = ~0ull;
    if constexpr (F * Z < 64 || F > 1) {
        slot_mask = (Z >= 64 ? ~0ull : ((1ull << Z) - 1ull)) << ((f * Z) & 63);
    }

    static_assert(!FUSED || LY::SINGLE, "the fused variant is built for single-wave workgroups");
    constexpr bool PREFETCH = !FUSED && LY::SINGLE && llr_prefetch_pays<C, TL>(LY::LDS_BYTES);  // see llr_prefetch
    __shared__ float stage[PREFETCH ? NB * 64 : 1];           // [NB][lane]
    __shared__ __align__(16) float fstage[FUSED ? F * n : 1];  // fused: channel LLRs by position
    __shared__ __align__(4) uint8_t fref[FUSED ? F * n : 4];   //        and the transmitted bits
    SweepAcc acc;
    auto prefetch = [&](long long g2) {
        if constexpr (PREFETCH) llr_prefetch<C, F>(stage, llr, g2, frames, lane_on, f, z);
    };
    const long long groups = (frames + F - 1) / F;
    if (blockIdx.x < groups) prefetch(blockIdx.x);
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long frame = grp * F + f;
        const bool active = lane_on && frame < frames;
        T L[NB], own[NB];
        if constexpr (PREFETCH) __builtin_amdgcn_s_waitcnt(0);  // the staged LLRs have landed
        if constexpr (FUSED) {  // one wave: its LDS operations execute in order, no barrier needed
            sweep_stage_wave<n, F>(sp, fstage, fref, grp, frames, tid);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        static_for<0, NB>([&](auto bc) {
            // + 0.0 turns -0.0 into +0.0: from here on "negative" is exactly "sign bit set"
            T l;
            if constexpr (FUSED) l = (active ? (T)fstage[f * n + bc * Z + z] : (T)0) + (T)0;
            else if constexpr (PREFETCH) l = (active ? (T)stage[bc * 64 + tid] : (T)0) + (T)0;
            else l = (active ? (T)llr[frame * n + bc * Z + z] : (T)0) + (T)0;
            L[bc] = l;
            own[bc] = l;  // ldpc.py:154
            if (lane_on) *reinterpret_cast<T *>(post_b + bc * Z * ST + zb) = l;  // idle lanes alias lane 0
        });
        if constexpr (PREFETCH) {
            __builtin_amdgcn_s_waitcnt(0);  // staging area read out before it is refilled
            if (grp + gridDim.x < groups) prefetch(grp + gridDim.x);
        }
        if constexpr (!LY::SINGLE) {
            if (tid < F) slot_bad[tid] = 0;
            __syncthreads();
        }

        // results of a frame: to the arrays, or (fused) into the error counters
        auto emit = [&](int iters, int conv, bool float_check) {
            unsigned errs = 0;
            static_for<0, NB>([&](auto bc) {
                const T v = own[bc];
                const uint8_t hv = float_check ? (v == (T)0 ? 0 : (v == (T)1 ? 1 : 2)) : (uint8_t)hard_bit<0>(v);
                if constexpr (FUSED) errs += hv != fref[f * n + bc * Z + z];
                else {
                    hard_out[frame * n + bc * Z + z] = hv;
                    if (post_out) post_out[frame * n + bc * Z + z] = v;
                }
            });
            if constexpr (FUSED) acc.book(errs, (__ballot(errs != 0) & slot_mask) != 0ull, z == 0, iters, conv);
            else if (z == 0) { iters_out[frame] = iters; conv_out[frame] = (uint8_t)conv; }
        };

        bool done = !active;
        int t = 0;
        int final_bad = 0;
        Pair cl[LOCAL ? MB : 1];  // check states of this lane's own checks (spec_r0_checks)
        for (; t <= g.max_iters; ++t) {
            // the pass at t == max_iters only exists in fixed-work mode (final syndrome)
            if (t == g.max_iters && g.early_exit) break;
            const bool float_check = t == 0 && g.float_syndrome0 && g.early_exit;
            int bad = 0;
            if (!done) {
                bad = float_check ? spec_r0_checks<C, T, true, ACT>(post_b, cs_b, zb, zp, own, cl, z, g.num_active_checks)
                                  : spec_r0_checks<C, T, false, ACT>(post_b, cs_b, zb, zp, own, cl, z, g.num_active_checks);
            }
            // ---- does every check of my slot hold?
            bool slot_ok;
            if constexpr (LY::SINGLE) {
                slot_ok = (__ballot(bad) & slot_mask) == 0ull;
            } else {
                if (bad && (g.early_exit || t == g.max_iters)) slot_bad[f] = 1;
                __syncthreads();
                slot_ok = slot_bad[f] == 0;
            }
            if (t == g.max_iters) { final_bad = !slot_ok; break; }
            if (g.early_exit) {
                if (!done && slot_ok) {  // ldpc.py:165-170
                    emit(t, 1, float_check);
                    done = true;
                }
                if constexpr (LY::SINGLE) {
                    if (__ballot(!done) == 0ull) break;
                } else {
                    if (!__syncthreads_or(!done)) break;
                    if (tid < F) slot_bad[tid] = 0;
                }
            }
            // ---- variable nodes
            if (!done) {
                static_for<0, NB>([&](auto bc) {
                    constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0;
                    const unsigned so = SNum<T>::sign_word(own[bc]) & 0x80000000u;
                    const T ao = t_abs(own[bc]);
                    T r[D > 0 ? D : 1];
                    unsigned off[D > 0 ? D : 1];
                    static_for<0, D>([&](auto k) { off[k] = rot_bwd<Z, SP, C::col_shift[k0 + k]>(zp); });
                    // two circulants in one block: the messages are summed lower check index first
                    // (ldpc.py:175,188), and which of the two checks that is depends on the lane.  Both lie in
                    // the same block row, so ordering the two offsets (loop-invariant) orders the messages.
                    // (the first two messages of a sum commute: a pair there needs no ordering)
                    // With LOCAL_SWAP a pair one of whose circulants has shift 0 takes that check state from
                    // the lane's registers and puts the two MESSAGES in order instead (a loop-invariant lane
                    // mask: the lane's own check comes first where its offset is the smaller one).
                    static_for<1, (D > 1 ? D - 1 : 1)>([&](auto k) {
                        if constexpr (C::col_pair[k0 + k] != 0) {
                            static_assert(C::col_row[k0 + k] == C::col_row[k0 + k + 1], "a pair shares its block row");
                            constexpr bool has_local = LOCAL_SWAP && (C::col_shift[k0 + k] == 0 || C::col_shift[k0 + k + 1] == 0);
                            if constexpr (!has_local) {
                                const unsigned a = off[k], b = off[k + 1];
                                off[k] = min(a, b);
                                off[k + 1] = max(a, b);
                            }
                        }
                    });
                    static_for<0, D>([&](auto k) {
                        constexpr int br = C::col_row[k0 + k];
                        // a pair whose messages are not the first two of the sum has to be put in order
                        constexpr bool ordered = k > 0 && (C::col_pair[k0 + k] != 0 || (k > 1 && C::col_pair[k0 + (k > 1 ? k - 1 : 0)] != 0));
                        Pair c;
                        if constexpr (LOCAL && C::col_shift[k0 + k] == 0 && (!ordered || LOCAL_SWAP)) c = cl[br];
                        else c = *reinterpret_cast<const Pair *>(cs_b + br * Z * SP + off[k]);
                        // both candidates already carry the check's sign parity; the variable's own sign
                        // is common to all its messages and is applied once, to their sum (negating
                        // every term of a sum negates the sum exactly; a sum of +0.0 that would have
                        // been -0.0 vanishes in the addition of the channel value below)
                        r[k] = ao == t_abs(c.x) ? c.y : c.x;
                    });
                    if constexpr (LOCAL_SWAP) {
                        static_for<1, (D > 1 ? D - 1 : 1)>([&](auto k) {
                            if constexpr (C::col_pair[k0 + k] != 0 && (C::col_shift[k0 + k] == 0 || C::col_shift[k0 + k + 1] == 0)) {
                                constexpr bool local_is_first = C::col_shift[k0 + k] == 0;
                                const bool own_first = zp < (local_is_first ? off[k + 1] : off[k]);
                                const bool swap = own_first != local_is_first;
                                const T a = r[k], b = r[k + 1];
                                r[k] = swap ? b : a;
                                r[k + 1] = swap ? a : b;
                            }
                        });
                    }
                    // ldpc.py:173,188: the reference starts from 0.0; 0.0 + r differs from r only in the sign
                    // of a zero, which the addition of the channel value (never -0.0) removes again
                    T s = D > 0 ? r[0] : (T)0;
                    static_for<1, (D > 1 ? D : 1)>([&](auto k) { s += r[k]; });
                    own[bc] = g.scale * (SNum<T>::flip_sign(s, so) + L[bc]);  // ldpc.py:196
                    *reinterpret_cast<T *>(post_b + bc * Z * ST + zb) = own[bc];
                });
            }
            if constexpr (!LY::SINGLE) __syncthreads();
        }
        if (!done)  // ldpc.py:210-227
            emit(g.early_exit ? g.max_iters - 1 : g.max_iters, g.early_exit ? 0 : !final_bad, false);
        if constexpr (!LY::SINGLE) __syncthreads();
    }
    if constexpr (FUSED) acc.flush_wave(sp.counters);
}

// ------------------------------------------------------------------------------------------ R1
// Rule R1 (bpsk.py:154-260): extrinsic normalised min-sum.  Single-wave workgroups only.
//
//   qv [EB][Z]  variable-to-check messages in VARIABLE order (k = position of the edge in its block
//               column's list): written linearly by the variable (bc, z), read rotated by the check
//   cs [MB][Z]  per check (m1, m2) = smallest / second smallest |q| with the sign parity of all
//               inputs in both sign bits; the variable rebuilds its message as
//               0.75 * (m2 if |q_own| == m1 else m1), sign = parity xor sign(q_own)  (bpsk.py:221-233)
// The variable keeps its own q per edge (registers when they fit, else re-read from qv), sums the
// rebuilt messages in ascending check order and writes q = L + (sum of the others) (bpsk.py:244-247).
// Hard decisions (`0 if b > 0 else 1`, bpsk.py:240-243) never touch memory: a ballot per block
// column gives a 64-bit mask in scalar registers, and the syndrome of block row br is the XOR of
// those masks rotated by the circulant shifts — a handful of scalar instructions per base edge.
template <class C, typename T, int F>
struct R1Layout {
    static constexpr int THREADS = ((F * C::Z + 63) / 64) * 64;
    static constexpr bool QREG = C::EB * (int)sizeof(T) / 4 <= 72;  // own messages in registers
    // edges of shift 0 stay inside the lane (see spec_r0_checks): their messages live in registers only
    // and have no place in LDS; row(e) = place of the message of edge e (variable order) in qv
    static constexpr bool LOCAL = spec_local_edges<C>() && QREG;
    static constexpr bool local(int e) { return LOCAL && C::col_shift[e] == 0; }
    static constexpr int row(int e)
    {
        int k = 0;
        for (int i = 0; i < e; ++i) k += !local(i);
        return k;
    }
    static constexpr int QV_BYTES = row(C::EB) * C::Z * (int)sizeof(T);
    static_assert(QV_BYTES % (2 * (int)sizeof(T)) == 0, "the check states behind the messages are aligned pairs");
    static constexpr int CS_BYTES = C::MB * C::Z * 2 * (int)sizeof(T);
    static constexpr int SLOT_BYTES = ((QV_BYTES + CS_BYTES + 127) / 128) * 128 + (F > 1 ? 32 : 0);
    static constexpr int LDS_BYTES = F * SLOT_BYTES;
};

template <class C, typename T, typename TL, int F, bool OMS, bool FUSED = false>  // OMS: the offset rule (LDPC_RULE_OMS) instead of R1's; FUSED: as in spec_r0_kernel
__global__ void __launch_bounds__(64) LDPC_SPEC_OCC
spec_r1_kernel(Cfg<T> g, const TL *__restrict__ llr, long long frames, uint8_t *__restrict__ hard_out,
               int *__restrict__ iters_out, uint8_t *__restrict__ conv_out, T *__restrict__ post_out, SweepDev sp)
{
    typedef R1Layout<C, T, F> LY;
    typedef typename SNum<T>::Pair Pair;
    static_assert(LY::THREADS == 64, "single-wave workgroups only");
    constexpr int Z = C::Z, NB = C::NB, MB = C::MB, EB = C::EB, ST = sizeof(T), SP = 2 * sizeof(T);
    constexpr int n = NB * Z;
    constexpr bool LOCAL = LY::LOCAL;  // edges of shift 0: message and check state never written to LDS
    // with local edges only THEIR messages have to live in registers; the others can be read back
    // from the lane's own LDS slot, which frees registers for one more wave per SIMD
#ifndef LDPC_R1_QLOCAL
#define LDPC_R1_QLOCAL 0
#endif
    constexpr bool QLOCAL = LOCAL && LDPC_R1_QLOCAL;
    auto q_in_reg = [](auto e) constexpr { return LY::QREG && (!QLOCAL || C::col_shift[e] == 0); };
    extern __shared__ __align__(16) unsigned char smem[];

    const int tid = threadIdx.x;
    const bool lane_on = tid < F * Z;
    const int f = lane_on ? tid / Z : 0;
    const int z = lane_on ? tid - f * Z : 0;
    unsigned char *qv_b = smem + f * LY::SLOT_BYTES;  // [EB][Z] T, variable order
    unsigned char *cs_b = qv_b + LY::QV_BYTES;        // [MB][Z] Pair
    const unsigned zb = (unsigned)z * ST, zp = (unsigned)z * SP;
    const unsigned long long slot_mask = (Z >= 64 ? ~0ull : ((1ull << (Z & 63)) - 1ull)) << ((f * Z) & 63);

    // in double with the messages in registers two waves per SIMD (eight workgroups) are all the registers allow
    constexpr bool PREFETCH = !FUSED && llr_prefetch_pays<C, TL>(LY::LDS_BYTES, sizeof(T) == 8 && LY::QREG ? 8 : 12);  // see llr_prefetch
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
        T q[LY::QREG ? EB : 1];
        unsigned long long hm[NB];
        if constexpr (PREFETCH) __builtin_amdgcn_s_waitcnt(0);  // the staged LLRs have landed
        if constexpr (FUSED) {  // one wave: its LDS operations execute in order, no barrier needed
            sweep_stage_wave<n, F>(sp, fstage, fref, grp, frames, tid);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        static_for<0, NB>([&](auto bc) {
            constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0;
            T l;
            // + 0: a channel value of -0.0 counts as +0.0 (sign(0) := +1, bpsk.py:227-229), and signs are
            // read from the sign bit from here on
            if constexpr (FUSED) l = (active ? (T)fstage[f * n + bc * Z + z] : (T)0) + (T)0;
            else if constexpr (PREFETCH) l = (active ? (T)stage[bc * 64 + tid] : (T)0) + (T)0;
            else l = (active ? (T)llr[frame * n + bc * Z + z] : (T)0) + (T)0;
            L[bc] = l;
            own[bc] = l;
            hm[bc] = __ballot(!(l > (T)0));
            static_for<0, D>([&](auto k) {  // bpsk.py:170-177
                if constexpr (q_in_reg(IC<k0 + k>{})) q[k0 + k] = l;
                if constexpr (!(LOCAL && C::col_shift[k0 + k] == 0)) {
                    if (lane_on) *reinterpret_cast<T *>(qv_b + LY::row(k0 + k) * Z * ST + zb) = l;  // idle lanes alias lane 0
                }
            });
        });
        if constexpr (PREFETCH) {
            __builtin_amdgcn_s_waitcnt(0);  // staging area read out before it is refilled
            if (grp + gridDim.x < groups) llr_prefetch<C, F>(stage, llr, grp + gridDim.x, frames, lane_on, f, z);
        }

        auto write_out = [&](int iters, int conv) {
            unsigned errs = 0;
            static_for<0, NB>([&](auto bc) {
                if constexpr (FUSED) errs += (uint8_t)hard_bit<1>(own[bc]) != fref[f * n + bc * Z + z];
                else {
                    hard_out[frame * n + bc * Z + z] = (uint8_t)hard_bit<1>(own[bc]);
                    if (post_out) post_out[frame * n + bc * Z + z] = own[bc];
                }
            });
            if constexpr (FUSED) acc.book(errs, (__ballot(errs != 0) & slot_mask) != 0ull, z == 0, iters, conv);
            else if (z == 0) { iters_out[frame] = iters; conv_out[frame] = (uint8_t)conv; }
        };

        bool done = !active;
        int t = 0;
        Pair cl[LOCAL ? MB : 1];  // check states of this lane's own checks
        for (; t < g.max_iters; ++t) {
            if (g.early_exit) {  // bpsk.py:205-209
                const bool ok = r1_slot_ok<C, F>(hm, slot_mask);
                if (!done && ok) { write_out(t, 1); done = true; }
                if (__ballot(!done) == 0ull) break;
            }
            if (!done) {
                // ---- check nodes (bpsk.py:213-233)
                static_for<0, MB>([&](auto br) {
                    constexpr int e0 = C::row_ptr[br], D = C::row_ptr[br + 1] - e0;
                    T m1 = Lim<T>::inf(), m2 = Lim<T>::inf();
                    unsigned sx = 0;
                    static_for<0, D>([&](auto p) {
                        constexpr int kpos = C::row_kpos[e0 + p], s = C::row_shift[e0 + p];
                        T v;
                        if constexpr (LOCAL && s == 0) v = q[kpos];
                        else v = *reinterpret_cast<const T *>(qv_b + LY::row(kpos) * Z * ST + rot_fwd<Z, ST, s>(zb));
                        SNum<T>::template book<p>(m1, m2, v);
                        sx ^= SNum<T>::sign_word(v);
                    });
                    if constexpr (D < 2) { m1 = (T)0; m2 = (T)0; }  // bpsk.py:216
                    const Pair cst = SNum<T>::make(SNum<T>::with_sign(m1, sx), SNum<T>::with_sign(m2, sx));
                    *reinterpret_cast<Pair *>(cs_b + br * Z * SP + zp) = cst;
                    if constexpr (LOCAL) cl[br] = cst;
                });
                // ---- variable nodes (bpsk.py:235-247)
                static_for<0, NB>([&](auto bc) {
                    constexpr int k0 = C::col_ptr[bc], D = C::col_ptr[bc + 1] - k0;
                    T r[D > 0 ? D : 1], qn[D > 0 ? D : 1];
                    unsigned off[D > 0 ? D : 1];
                    static_for<0, D>([&](auto k) {
                        constexpr int br = C::col_row[k0 + k], s = C::col_shift[k0 + k];
                        T qo;
                        if constexpr (q_in_reg(IC<k0 + k>{})) qo = q[k0 + k];
                        else qo = *reinterpret_cast<const T *>(qv_b + LY::row(k0 + k) * Z * ST + zb);
                        off[k] = rot_bwd<Z, SP, s>(zp);
                        Pair c;
                        if constexpr (LOCAL && s == 0) c = cl[br];
                        else c = *reinterpret_cast<const Pair *>(cs_b + br * Z * SP + off[k]);
                        const T sel = t_abs(qo) == t_abs(c.x) ? c.y : c.x;
                        // sel carries the sign parity of the check in its sign bit
                        const T mag = OMS ? SNum<T>::flip_sign(oms_magnitude(t_abs(sel), g.scale, g.offset),
                                                               SNum<T>::sign_word(sel) & 0x80000000u)
                                          : g.scale * sel;
                        r[k] = SNum<T>::flip_sign(mag, SNum<T>::sign_word(qo) & 0x80000000u);
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
                        if constexpr (q_in_reg(IC<k0 + k>{})) q[k0 + k] = qn[k];
                        if constexpr (!(LOCAL && C::col_shift[k0 + k] == 0))
                            *reinterpret_cast<T *>(qv_b + LY::row(k0 + k) * Z * ST + zb) = qn[k];
                    });
                    own[bc] = L[bc] + run;  // bpsk.py:237
                    hm[bc] = __ballot(!(own[bc] > (T)0));
                });
            }
        }
        if (!done) {  // bpsk.py:251-260
            if (g.early_exit) write_out(g.max_iters - 1, 0);
            else write_out(g.max_iters, r1_slot_ok<C, F>(hm, slot_mask) ? 1 : 0);
        }
    }
    if constexpr (FUSED) acc.flush_wave(sp.counters);
}

// ------------------------------------------------------------------------------------------ host
// This file is compiled twice: as it is (the decoders behind ldpc_decode_batch) and with
// -DLDPC_SPEC_FUSED (the same kernels with the channel in front and the error accounting behind,
// for ldpc_sweep_batch), so that the two sets of instantiations build side by side.
#ifdef LDPC_SPEC_FUSED
constexpr bool kFused = true;
#else
constexpr bool kFused = false;
#endif

template <class C>
static bool code_matches(int Z, int mb, int nb, int Eb, const int *er, const int *ec, const int *es)
{
    if (Z != C::Z || mb != C::MB || nb != C::NB || Eb != C::EB) return false;
    for (int br = 0; br < mb; ++br)
        for (int e = C::row_ptr[br]; e < C::row_ptr[br + 1]; ++e)
            if (er[e] != br || ec[e] != C::row_col[e] || es[e] != C::row_shift[e]) return false;
    return true;
}

template <class C, typename T, typename TL, int F>
static hipError_t launch_spec_r0(const ldpc_decode_config &cfg, const Launch &io, int num_cus, const char **name,
                                 const SweepDev *sp)
{
    typedef SpecLayout<C, T, F> LY;
    static_assert(LY::LDS_BYTES <= 160 * 1024, "LDS limit");
    Cfg<T> g;
    g.rule = 0; g.max_iters = cfg.max_iters; g.early_exit = cfg.early_exit;
    const bool bounded = !kFused && cfg.num_active_checks >= 0 && cfg.num_active_checks < C::MB * C::Z;
    g.float_syndrome0 = cfg.float_syndrome0; g.num_active_checks = bounded ? cfg.num_active_checks : -1; g.num_refresh_vars = -1;
    g.scale = (T)cfg.scale; g.clip_tanh_arg = (T)0; g.clip_prod = (T)0; g.clip_post = (T)0; g.offset = (T)0;
    auto k = spec_r0_kernel<C, T, TL, F, kFused>;
    if constexpr (!kFused) {
        if (bounded) k = spec_r0_kernel<C, T, TL, F, false, true>;
    }
    int per_cu = 0;
    hipError_t e = kernel_setup(reinterpret_cast<const void *>(k), LY::THREADS, LY::LDS_BYTES, &per_cu);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    const long long groups = (io.frames + F - 1) / F;
    long long grid = (long long)num_cus * (long long)per_cu * 2;
    if (grid > groups) grid = groups;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(LY::THREADS), LY::LDS_BYTES, io.stream, g,
                       reinterpret_cast<const TL *>(io.llr), io.frames, io.hard, io.iters, io.converged,
                       reinterpret_cast<T *>(io.posterior), sp ? *sp : SweepDev{});
    *name = kFused ? (sizeof(T) == 8 ? "fused_spec_r0_f64" : "fused_spec_r0_f32") : (sizeof(T) == 8 ? "spec_r0_f64" : "spec_r0_f32");
    return hipGetLastError();
}

template <class C, typename T, typename TL, int F, bool OMS = false>
static hipError_t launch_spec_r1(const ldpc_decode_config &cfg, const Launch &io, int num_cus, const char **name,
                                 const SweepDev *sp)
{
    typedef R1Layout<C, T, F> LY;
    constexpr int n = C::NB * C::Z;
    Cfg<T> g;
    g.rule = cfg.rule; g.max_iters = cfg.max_iters; g.early_exit = cfg.early_exit;
    g.float_syndrome0 = 0; g.num_active_checks = -1; g.num_refresh_vars = -1;
    g.scale = (T)cfg.scale; g.clip_tanh_arg = (T)0; g.clip_prod = (T)0; g.clip_post = (T)0; g.offset = (T)cfg.offset;
    size_t per_cu = (160 * 1024) / (LY::LDS_BYTES + (kFused ? F * n * 5 : 0));
    if (per_cu > 32) per_cu = 32;
    if (per_cu < 1) per_cu = 1;
    const long long groups = (io.frames + F - 1) / F;
    long long grid = (long long)num_cus * (long long)per_cu * 2;
    if (grid > groups) grid = groups;
    if (grid < 1) grid = 1;
    auto k = spec_r1_kernel<C, T, TL, F, OMS, kFused>;
    hipError_t e = kernel_setup(reinterpret_cast<const void *>(k), 64, LY::LDS_BYTES, nullptr);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(64), LY::LDS_BYTES, io.stream, g,
                       reinterpret_cast<const TL *>(io.llr), io.frames, io.hard, io.iters, io.converged,
                       reinterpret_cast<T *>(io.posterior), sp ? *sp : SweepDev{});
    if (kFused) *name = OMS ? (sizeof(T) == 8 ? "fused_spec_oms_f64" : "fused_spec_oms_f32") : (sizeof(T) == 8 ? "fused_spec_r1_f64" : "fused_spec_r1_f32");
    else *name = OMS ? (sizeof(T) == 8 ? "spec_oms_f64" : "spec_oms_f32") : (sizeof(T) == 8 ? "spec_r1_f64" : "spec_r1_f32");
    return hipGetLastError();
}

// sp != NULL only in the fused build, where the LLR type plays no role (LLRs never exist in memory)
template <class C, int F>
static hipError_t launch_spec_code(const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                                   int num_cus, const char **name, bool *supported, const SweepDev *sp)
{
    *supported = true;
    const bool llr64 = !kFused && io.llr_dtype == LDPC_F64, llr32 = kFused || io.llr_dtype == LDPC_F32;
    if constexpr (C::EB <= 100 && (!kFused || SpecLayout<C, double, F>::SINGLE)) {  // flooding kernels unroll the whole base graph: small codes only
        if (cfg.rule == 0) {
            if (compute_dtype == LDPC_F64) {
                if constexpr (!kFused) {
                    if (llr64) return launch_spec_r0<C, double, double, F>(cfg, io, num_cus, name, sp);
                }
                return launch_spec_r0<C, double, float, F>(cfg, io, num_cus, name, sp);
            }
            if (llr32) return launch_spec_r0<C, float, float, F>(cfg, io, num_cus, name, sp);
        }
    }
    if constexpr (C::EB <= 100 && R1Layout<C, double, F>::THREADS == 64 && R1Layout<C, double, F>::LDS_BYTES <= 160 * 1024) {
        if (cfg.rule == 1) {
            if (compute_dtype == LDPC_F64) {
                if constexpr (!kFused) {
                    if (llr64) return launch_spec_r1<C, double, double, F>(cfg, io, num_cus, name, sp);
                }
                return launch_spec_r1<C, double, float, F>(cfg, io, num_cus, name, sp);
            }
            if (llr32) return launch_spec_r1<C, float, float, F>(cfg, io, num_cus, name, sp);
        }
        if (cfg.rule == LDPC_RULE_OMS) {
            if (compute_dtype == LDPC_F64) {
                if constexpr (!kFused) {
                    if (llr64) return launch_spec_r1<C, double, double, F, true>(cfg, io, num_cus, name, sp);
                }
                return launch_spec_r1<C, double, float, F, true>(cfg, io, num_cus, name, sp);
            }
            if (llr32) return launch_spec_r1<C, float, float, F, true>(cfg, io, num_cus, name, sp);
        }
    }
    *supported = false;
    return hipSuccess;
}

#ifdef LDPC_SPEC_FUSED
// Fused sweep on the single-wave flooding kernels (rules R0, R1, OMS on the small compiled-in
// codes); the layered rule has its own fused variant in decode_lnms.hip.
hipError_t launch_spec_fused(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                             const SweepDev &sp, int num_cus, Scratch &workspace, const char **name, bool *supported)
{
    if (cfg.rule == LDPC_RULE_LNMS)
        return launch_spec_lnms_fused(spec_id, cfg, compute_dtype, io, sp, num_cus, workspace, name, supported);
    int id = 0;
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) return launch_spec_code<CODE, SLOTS>(cfg, compute_dtype, io, num_cus, name, supported, &sp); \
    ++id;
    LDPC_FOR_EACH_SPEC_CODE(LDPC_LAUNCH)
#undef LDPC_LAUNCH
    *supported = false;
    return hipSuccess;
}
#else
int spec_match(int Z, int mb, int nb, int Eb, const int *er, const int *ec, const int *es)
{
    int id = 0;
#define LDPC_MATCH(CODE, SLOTS)                                       \
    if (code_matches<CODE>(Z, mb, nb, Eb, er, ec, es)) return id;     \
    ++id;
    LDPC_FOR_EACH_SPEC_CODE(LDPC_MATCH)
#undef LDPC_MATCH
    return -1;
}

// The reference's loop bounds (num_active_checks under R0, num_refresh_vars under R1-R4) on the compiled-in
// kernels that carry them as template switches: the single-wave R0 kernel, and the multi-wave R1 and
// sum-product kernels of the large base graphs.  supported = false: the table-driven kernels serve the call.
hipError_t launch_spec_bounded(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                               int num_cus, const char **name, bool *supported)
{
    *supported = false;
    if (cfg.rule == 1) return launch_spec_r1w(spec_id, cfg, compute_dtype, io, num_cus, name, supported);
    if (cfg.rule == 2) return launch_spec_spaw_rule_2(spec_id, cfg, compute_dtype, io, num_cus, name, supported);
    if (cfg.rule == 3) return launch_spec_spaw_rule_3(spec_id, cfg, compute_dtype, io, num_cus, name, supported);
    if (cfg.rule == 4) return launch_spec_spaw_rule_4(spec_id, cfg, compute_dtype, io, num_cus, name, supported);
    if (cfg.rule != 0) return hipSuccess;
    int id = 0;
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) return launch_spec_code<CODE, SLOTS>(cfg, compute_dtype, io, num_cus, name, supported, nullptr); \
    ++id;
    LDPC_FOR_EACH_SPEC_CODE(LDPC_LAUNCH)
#undef LDPC_LAUNCH
    return hipSuccess;
}

hipError_t launch_spec(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                       int num_cus, const char **name, bool *supported)
{
    if (cfg.rule == LDPC_RULE_LNMS)  // decode_lnms.hip
        return launch_spec_lnms(spec_id, cfg, compute_dtype, io, num_cus, name, supported);
    if (cfg.rule >= 2 && cfg.rule <= 4) {  // decode_spec_spa.hip, else decode_spec_spaw.hip
        hipError_t e = launch_spec_spa(spec_id, cfg, compute_dtype, io, num_cus, name, supported);
        if (e != hipSuccess || *supported) return e;
        if (cfg.rule == 2) return launch_spec_spaw_rule_2(spec_id, cfg, compute_dtype, io, num_cus, name, supported);
        if (cfg.rule == 3) return launch_spec_spaw_rule_3(spec_id, cfg, compute_dtype, io, num_cus, name, supported);
        return launch_spec_spaw_rule_4(spec_id, cfg, compute_dtype, io, num_cus, name, supported);
    }
    if (cfg.rule == 0 || cfg.rule == 1 || cfg.rule == LDPC_RULE_OMS) {  // decode_spec_r1w.hip: codes too large for one wave per workgroup
        hipError_t e = cfg.rule == 0   ? launch_spec_r0w(spec_id, cfg, compute_dtype, io, num_cus, name, supported)
                       : cfg.rule == 1 ? launch_spec_r1w(spec_id, cfg, compute_dtype, io, num_cus, name, supported)
                                       : launch_spec_omsw(spec_id, cfg, compute_dtype, io, num_cus, name, supported);
        if (e != hipSuccess || *supported) return e;
    }
    int id = 0;
#define LDPC_LAUNCH(CODE, SLOTS)                                                                   \
    if (spec_id == id) return launch_spec_code<CODE, SLOTS>(cfg, compute_dtype, io, num_cus, name, supported, nullptr); \
    ++id;
    LDPC_FOR_EACH_SPEC_CODE(LDPC_LAUNCH)
#undef LDPC_LAUNCH
    *supported = false;
    return hipSuccess;
}
#endif

}  // namespace ldpc
