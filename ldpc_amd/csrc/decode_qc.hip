// Circulant (quasi-cyclic) flooding decoder: rules R0-R4, float or double.
//
// H is an mb x nb array of Z x Z circulant blocks.  A workgroup decodes F frames at once
// ("slots"); a thread works on index z of slot f.  In the check-node phase thread (f, z) owns
// check (br, z) of the block rows assigned to its wave, in the variable-node phase variable
// (bc, z) of its wave's block columns (qc_map_thread below).  The base graph (which block column and
// which shift each circulant has) is the same for every thread, so it is read through uniform
// (scalar) loads and every loop bound is wave-uniform.
//
// LDS layout (element indices, FZ = F*Z):
//   msg [Eb][FZ]   one message per edge, updated in place: check (br, z) reads the
//                  variable-to-check message at [e][f*Z + z] and overwrites it with its
//                  check-to-variable message; variable (bc, z') finds that edge at
//                  [e][f*Z + (z' - shift_e) mod Z] and overwrites it with the next
//                  variable-to-check message.  Check-side access is linear in tid, variable-side
//                  access is a rotation of it: both are free of bank conflicts.
//   L   [nb][FZ]   channel LLRs
//   post[nb][FZ]   posteriors (R0: always, it is the check-node input; else only when requested)
//   hb  [nb][FZ]   hard decisions, one byte each (R1-R4; R0 derives them from post)
//
// Summation order follows the reference exactly (ascending check index per variable,
// bpsk.py:236-247; ascending accumulation ldpc.py:173-188); two circulants in the same block are
// ordered per lane by their check index.
#include <cstdlib>

#include "bp_common.h"
#include "bp_math.h"
#include "spec_common.h"  // spa_check_tanh

namespace ldpc {

// The circulant tables are read through the constant address space: with a wave-uniform index the
// compiler then issues scalar loads and unpacks the entries on the scalar unit.
typedef const unsigned __attribute__((address_space(4))) *const_u32;
typedef const int __attribute__((address_space(4))) *const_i32;
__device__ __forceinline__ const_u32 as_const(const unsigned *p) { return (const_u32)(unsigned long long)p; }
__device__ __forceinline__ const_i32 as_const(const int *p) { return (const_i32)(unsigned long long)p; }

struct QcCtx {
    int Z, F, FZ, mb, nb;
    int tid, fz0, z;  // tid = f*Z + z: position of this thread's (slot, z) pair in an [..][FZ] row; fz0 = f*Z
    int w0, wstep;    // this wave takes block rows / block columns w0, w0 + wstep, ...
    bool lane_on;     // thread maps to a (slot, z) pair at all
};

// Thread mapping shared by the table-driven kernels.  A wave covers 64 consecutive z of one slot
// (Z >= 64; a block row then spans ceil(Z/64) waves) or floor(64/Z) slots side by side (Z < 64).
// W such wave sets work on one slot group at a time and split its block rows (check phase) and
// block columns (variable phase) between them; G groups share a workgroup.  So every wave sees a
// single block row per step (its degree and table entries are wave-uniform) and a frame of a code
// with a small Z and a large base graph still spreads over many waves.
__device__ __forceinline__ int qc_map_thread(QcCtx &x, int Z, int W)
{
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int S = Z >= 64 ? 1 : 64 / Z, Zw = (Z + 63) / 64;
    const int group = wave / (W * Zw), within = wave % (W * Zw);
    const int s = Z >= 64 ? 0 : lane / Z;
    x.w0 = within / Zw;
    x.wstep = W;
    x.z = Z >= 64 ? (within % Zw) * 64 + lane : lane - s * Z;
    x.lane_on = Z >= 64 ? x.z < Z : s < S;
    const int f = x.lane_on ? group * S + s : 0;
    if (!x.lane_on) x.z = 0;
    x.fz0 = f * Z;
    x.tid = x.fz0 + x.z;
    return f;
}

#define LDPC_SHIFT_MASK 0xfffu
#define LDPC_PAIR_BIT 0x8000u

// ------------------------------------------------------------------------------------------
// check node of degree D for thread (f, z); returns the parity of the hard decisions (or, for the
// R0 iteration-0 check on raw LLRs, whether the float row sum fails the "% 2 == 0" test).
template <int RULE, typename T, int D>
__device__ __forceinline__ int check_row(const QcCtx &x, const Cfg<T> &g, const_u32 info,
                                         int e0, T *msg, const T *post, const uint8_t *hb,
                                         bool want_syndrome, bool float_check, bool update)
{
    T v[D];
    unsigned par = 0;
#pragma unroll
    for (int p = 0; p < D; ++p) {
        const unsigned w = info[e0 + p];
        const int bc = (int)(w >> 16), s = (int)(w & LDPC_SHIFT_MASK);
        int col = x.z + s;
        col -= col >= x.Z ? x.Z : 0;
        if (RULE == 0) {
            v[p] = post[bc * x.FZ + x.fz0 + col];
        } else {
            v[p] = msg[(e0 + p) * x.FZ + x.tid];
            if (want_syndrome) par ^= hb[bc * x.FZ + x.fz0 + col];
        }
    }
    int bad = 0;
    if (RULE == 0 && want_syndrome) {
        if (float_check) {  // ldpc.py:158,165
            T s = (T)0;
#pragma unroll
            for (int p = 0; p < D; ++p) s += v[p];
            bad = !even_integer(s);
        } else {
#pragma unroll
            for (int p = 0; p < D; ++p) par ^= hard_bit<0>(v[p]);
            bad = (int)par;
        }
    } else {
        bad = (int)par;
    }
    if (!update) return bad;

    if (RULE <= 1 || RULE == 6) {
        if (D >= 2) {  // ldpc.py:178, bpsk.py:216
            MinSumState<T> st;
            st.init();
#pragma unroll
            for (int p = 0; p < D; ++p) st.add(p, v[p]);
#pragma unroll
            for (int p = 0; p < D; ++p)
                msg[(e0 + p) * x.FZ + x.tid] = RULE == 0 ? r0_message(st, p)
                                               : (RULE == 6 ? oms_message(st, p, g.scale, g.offset) : r1_message(st, p, g.scale));
        } else {
            msg[e0 * x.FZ + x.tid] = (T)0;
        }
    } else if (RULE == 2 || RULE == 3) {
        T o[D];  // the whole check at once: see spa_check_tanh (spec_common.h)
        spa_check_tanh<T, D>(v, o, g.clip_tanh_arg, g.clip_prod, g.scale, RULE == 3);
#pragma unroll
        for (int p = 0; p < D; ++p) msg[(e0 + p) * x.FZ + x.tid] = o[p];
    } else {
        T ph[D], suf[D];
        unsigned neg = 0;
#pragma unroll
        for (int p = 0; p < D; ++p) {
            neg |= (unsigned)(v[p] < (T)0) << p;  // channel.py:332-337: 0 and 1e-9 count as +
            ph[p] = phi(t_abs(v[p]));
        }
        T run = (T)0;
#pragma unroll
        for (int p = D - 1; p >= 0; --p) { suf[p] = run; run += ph[p]; }
        T pre = (T)0;
#pragma unroll
        for (int p = 0; p < D; ++p) {
            const bool negative = (__popc(neg & ~(1u << p)) & 1) != 0;
            msg[(e0 + p) * x.FZ + x.tid] = phi_message(pre + suf[p], negative);
            pre += ph[p];
        }
    }
    return bad;
}

// ------------------------------------------------------------------------------------------
// variable node of degree D for thread (f, z)
template <int RULE, typename T, int D>
__device__ __forceinline__ void var_col(const QcCtx &x, const Cfg<T> &g, const_u32 info,
                                        int k0, int bc, T *msg, const T *L, T *post, uint8_t *hb,
                                        bool want_post)
{
    T r[D > 0 ? D : 1];
    int idx[D > 0 ? D : 1];
    int zz[D > 0 ? D : 1];
    bool pair[D > 0 ? D : 1];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const unsigned w = info[k0 + k];
        const int slot = (int)(w >> 16), s = (int)(w & LDPC_SHIFT_MASK);
        pair[k] = (w & LDPC_PAIR_BIT) != 0;
        int q = x.z - s;
        q += q < 0 ? x.Z : 0;
        zz[k] = q;
        idx[k] = slot * x.FZ + x.fz0 + q;
        r[k] = msg[idx[k]];
    }
    // two circulants of one block: the check with the smaller index inside the block row comes
    // first in ascending check order
    bool sw[D > 0 ? D : 1];
#pragma unroll
    for (int k = 0; k + 1 < D; ++k) {
        sw[k] = pair[k] && zz[k] > zz[k + 1];
        if (pair[k]) {
            T a = r[k], b = r[k + 1];
            r[k] = sw[k] ? b : a;
            r[k + 1] = sw[k] ? a : b;
        }
    }
    const int vi = bc * x.FZ + x.fz0 + x.z;
    const T l = L[vi];
    T b;
    if (RULE == 0) {
        T s = (T)0;  // ldpc.py:173,188
#pragma unroll
        for (int k = 0; k < D; ++k) s += r[k];
        b = g.scale * (s + l);  // ldpc.py:196
        post[vi] = b;
    } else {
        T q[D > 0 ? D : 1];
        T run = (T)0;
#pragma unroll
        for (int k = 0; k < D; ++k) {  // bpsk.py:244-247
            T ext = run;
#pragma unroll
            for (int kk = k + 1; kk < D; ++kk) ext += r[kk];
            run += r[k];
            q[k] = l + ext;
        }
#pragma unroll
        for (int k = 0; k + 1 < D; ++k) {
            if (pair[k]) {
                T a = q[k], c = q[k + 1];
                q[k] = sw[k] ? c : a;
                q[k + 1] = sw[k] ? a : c;
            }
        }
        // bpsk.py:244, channel.py:241,358,480: the refresh loop runs over range(self.n), the constructor's
        // n; a variable beyond it keeps sending its channel value
        const bool refresh = g.num_refresh_vars < 0 || bc * x.Z + x.z < g.num_refresh_vars;
#pragma unroll
        for (int k = 0; k < D; ++k) msg[idx[k]] = refresh ? q[k] : l;
        b = l + run;  // bpsk.py:237
        if (g.clip_post > (T)0) b = clip(b, -g.clip_post, g.clip_post);
        hb[vi] = (uint8_t)hard_bit<RULE>(b);
        if (want_post) post[vi] = b;
    }
}

#define LDPC_CASE_ROW(N)                                                                          \
    case N:                                                                                       \
        if constexpr (N <= DC)                                                                    \
            bad |= check_row<RULE, T, N>(x, g, as_const(q.row_info), e0, msg, post, hb, want_syn, float_check, update); \
        break;
#define LDPC_CASE_COL(N)                                                                          \
    case N:                                                                                       \
        if constexpr (N <= DV)                                                                    \
            var_col<RULE, T, N>(x, g, as_const(q.col_info), k0, bc, msg, L, post, hb, want_post);            \
        break;

template <int RULE, typename T, int DC>
__device__ __forceinline__ int check_phase(const QcCtx &x, const Cfg<T> &g, const QcDev &q, T *msg,
                                           const T *post, const uint8_t *hb, bool want_syn,
                                           bool float_check, bool update)
{
    int bad = 0;
    for (int br = x.w0; br < x.mb; br += x.wstep) {
        const int e0 = as_const(q.row_ptr)[br], d = as_const(q.row_ptr)[br + 1] - e0;
        switch (d) {
            LDPC_CASE_ROW(1) LDPC_CASE_ROW(2) LDPC_CASE_ROW(3) LDPC_CASE_ROW(4) LDPC_CASE_ROW(5)
            LDPC_CASE_ROW(6) LDPC_CASE_ROW(7) LDPC_CASE_ROW(8) LDPC_CASE_ROW(9) LDPC_CASE_ROW(10)
            LDPC_CASE_ROW(11) LDPC_CASE_ROW(12) LDPC_CASE_ROW(13) LDPC_CASE_ROW(14) LDPC_CASE_ROW(15)
            LDPC_CASE_ROW(16) LDPC_CASE_ROW(17) LDPC_CASE_ROW(18) LDPC_CASE_ROW(19) LDPC_CASE_ROW(20)
            default: break;
        }
    }
    return bad;
}

template <int RULE, typename T, int DV>
__device__ __forceinline__ void var_phase(const QcCtx &x, const Cfg<T> &g, const QcDev &q, T *msg,
                                          const T *L, T *post, uint8_t *hb, bool want_post)
{
    for (int bc = x.w0; bc < x.nb; bc += x.wstep) {
        const int k0 = as_const(q.col_ptr)[bc], d = as_const(q.col_ptr)[bc + 1] - k0;
        switch (d) {
            case 0: var_col<RULE, T, 0>(x, g, as_const(q.col_info), k0, bc, msg, L, post, hb, want_post); break;
            LDPC_CASE_COL(1) LDPC_CASE_COL(2) LDPC_CASE_COL(3) LDPC_CASE_COL(4) LDPC_CASE_COL(5)
            LDPC_CASE_COL(6) LDPC_CASE_COL(7) LDPC_CASE_COL(8) LDPC_CASE_COL(9) LDPC_CASE_COL(10)
            LDPC_CASE_COL(11) LDPC_CASE_COL(12) LDPC_CASE_COL(13) LDPC_CASE_COL(14) LDPC_CASE_COL(15)
            LDPC_CASE_COL(16) LDPC_CASE_COL(17) LDPC_CASE_COL(18) LDPC_CASE_COL(19) LDPC_CASE_COL(20)
            LDPC_CASE_COL(21) LDPC_CASE_COL(22) LDPC_CASE_COL(23) LDPC_CASE_COL(24) LDPC_CASE_COL(25)
            LDPC_CASE_COL(26) LDPC_CASE_COL(27) LDPC_CASE_COL(28) LDPC_CASE_COL(29) LDPC_CASE_COL(30)
            LDPC_CASE_COL(31) LDPC_CASE_COL(32)
            default: break;
        }
    }
}

// ------------------------------------------------------------------------------------------
template <int RULE, typename T, typename TL, int DC, int DV>
__global__ void __launch_bounds__(512) decode_qc_kernel(QcDev q, Cfg<T> g, int F, int W, const TL *__restrict__ llr, long long frames,
                                 uint8_t *__restrict__ hard_out, int *__restrict__ iters_out,
                                 uint8_t *__restrict__ conv_out, T *__restrict__ post_out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    QcCtx x;
    x.Z = q.Z; x.F = F; x.FZ = F * q.Z; x.mb = q.mb; x.nb = q.nb;
    const int f = qc_map_thread(x, q.Z, W);
    const bool reporter = x.lane_on && x.z == 0 && x.w0 == 0;  // one thread per slot writes iters / converged
    const int n = q.nb * q.Z;
    const bool want_post = post_out != nullptr;
    const bool keep_post = RULE == 0 || want_post;

    T *msg = reinterpret_cast<T *>(smem);
    T *L = msg + (size_t)q.Eb * x.FZ;
    T *post = L + (size_t)q.nb * x.FZ;
    uint8_t *hb = reinterpret_cast<uint8_t *>(post + (keep_post ? (size_t)q.nb * x.FZ : 0));
    int *slot_bad = reinterpret_cast<int *>(hb + (((RULE == 0 ? 0 : (size_t)q.nb * x.FZ) + 15) & ~(size_t)15));

    const long long groups = (frames + F - 1) / F;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long frame = grp * F + f;
        const bool active = x.lane_on && frame < frames;
        // ---- channel LLRs in, messages initialised (ldpc.py:151-158, bpsk.py:170-196)
        if (x.lane_on) {
            for (int bc = x.w0; bc < x.nb; bc += x.wstep) {
                const int vi = bc * x.FZ + x.tid;
                T l = active ? (T)llr[frame * n + bc * x.Z + x.z] : (T)0;
                L[vi] = l;
                if (keep_post) post[vi] = l;
                if (RULE != 0) hb[vi] = (uint8_t)hard_bit<RULE>(l);
            }
        }
        if ((int)threadIdx.x < F) slot_bad[threadIdx.x] = 0;
        __syncthreads();
        if (RULE != 0 && x.lane_on) {
            for (int br = x.w0; br < x.mb; br += x.wstep)
                for (int e = q.row_ptr[br]; e < q.row_ptr[br + 1]; ++e) {
                    const unsigned w = q.row_info[e];
                    int col = x.z + (int)(w & LDPC_SHIFT_MASK);
                    col -= col >= x.Z ? x.Z : 0;
                    msg[e * x.FZ + x.tid] = L[(int)(w >> 16) * x.FZ + x.fz0 + col];
                }
        }
        __syncthreads();

        bool done = !active;
        int t = 0;
        for (; t < g.max_iters; ++t) {
            const bool float_check = RULE == 0 && t == 0 && g.float_syndrome0;
            // ---- syndrome of the current hard decisions fused with the check-node update
            if (!done) {
                int bad = check_phase<RULE, T, DC>(x, g, q, msg, post, hb, g.early_exit != 0,
                                                   float_check, true);
                if (bad && g.early_exit) slot_bad[f] = 1;
            }
            __syncthreads();
            if (g.early_exit) {
                if (!done && !slot_bad[f]) {
                    // ldpc.py:165-170: every parity check holds -> this frame is finished
                    for (int bc = x.w0; bc < x.nb; bc += x.wstep) {
                        const int vi = bc * x.FZ + x.tid;
                        uint8_t h;
                        if (RULE == 0) {
                            T v = post[vi];
                            h = float_check ? (v == (T)0 ? 0 : (v == (T)1 ? 1 : 2)) : (uint8_t)hard_bit<0>(v);
                        } else {
                            h = hb[vi];
                        }
                        hard_out[frame * n + bc * x.Z + x.z] = h;
                        if (want_post) post_out[frame * n + bc * x.Z + x.z] = post[vi];
                    }
                    if (reporter) { iters_out[frame] = t; conv_out[frame] = 1; }
                    done = true;
                }
                if (!__syncthreads_or(!done)) break;  // every slot of the workgroup has finished
                if ((int)threadIdx.x < F) slot_bad[threadIdx.x] = 0;
            }
            // ---- variable nodes
            if (!done) var_phase<RULE, T, DV>(x, g, q, msg, L, post, hb, want_post);
            __syncthreads();
        }
        // ---- frames that never passed a check at the top of the loop (ldpc.py:210-227)
        if (!g.early_exit) {
            int bad = 0;
            if (!done) bad = check_phase<RULE, T, DC>(x, g, q, msg, post, hb, true, false, false);
            if (bad) slot_bad[f] = 1;
            __syncthreads();
        }
        if (!done) {
            for (int bc = x.w0; bc < x.nb; bc += x.wstep) {
                const int vi = bc * x.FZ + x.tid;
                hard_out[frame * n + bc * x.Z + x.z] = RULE == 0 ? (uint8_t)hard_bit<0>(post[vi]) : hb[vi];
                if (want_post) post_out[frame * n + bc * x.Z + x.z] = post[vi];
            }
            if (reporter) {
                iters_out[frame] = g.early_exit ? g.max_iters - 1 : g.max_iters;
                conv_out[frame] = g.early_exit ? 0 : (uint8_t)!slot_bad[f];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
template <typename T>
static Cfg<T> make_cfg(const ldpc_decode_config &c)
{
    Cfg<T> g;
    g.rule = c.rule; g.max_iters = c.max_iters; g.early_exit = c.early_exit;
    g.float_syndrome0 = c.float_syndrome0; g.num_active_checks = c.num_active_checks;
    g.num_refresh_vars = c.num_refresh_vars;
    g.scale = (T)c.scale; g.clip_tanh_arg = (T)c.clip_tanh_arg; g.clip_prod = (T)c.clip_prod;
    g.clip_post = (T)c.clip_post; g.offset = (T)c.offset;
    return g;
}

static const size_t kLdsLimit = 160 * 1024;

template <typename T>
static size_t qc_lds_bytes(const QcDev &q, int rule, int F, bool want_post)
{
    const size_t FZ = (size_t)F * q.Z;
    size_t b = (size_t)q.Eb * FZ * sizeof(T) + (size_t)q.nb * FZ * sizeof(T);
    if (rule == 0 || want_post) b += (size_t)q.nb * FZ * sizeof(T);
    if (rule != 0) b += ((size_t)q.nb * FZ + 15) & ~(size_t)15;
    b += (size_t)F * sizeof(int) + 16;
    return b;
}

// Workgroup shape: G slot groups x W row-parallel wave sets x ceil(Z/64) waves, at most 8 waves.
// Small frames: several groups (independent frames) per workgroup; large frames: one group whose
// block rows / columns are split over up to 8 waves.
struct QcShape { int F, W, threads; };
static QcShape pick_shape(const QcDev &q, size_t bytes_per_slot)
{
    const int S = q.Z >= 64 ? 1 : 64 / q.Z, Zw = (q.Z + 63) / 64;
    int G = 1;
    while (G < 4 && (size_t)(G + 1) * S * bytes_per_slot <= kLdsLimit / 2 && (G + 1) * Zw <= 8) ++G;
    int W = 8 / (G * Zw);
    if (W < 1) W = 1;
    const int most = q.mb < q.nb ? q.mb : q.nb;
    if (W > most) W = most;
    return QcShape{G * S, W, G * W * Zw * 64};
}

template <int RULE, typename T, typename TL, int DC, int DV>
static hipError_t launch_qc_t(const QcDev &q, const ldpc_decode_config &cfg, const Launch &io,
                              int num_cus, const char **name, bool *supported)
{
    const bool want_post = io.posterior != nullptr;
    const int S = q.Z >= 64 ? 1 : 64 / q.Z;
    const size_t per_slot = qc_lds_bytes<T>(q, RULE, 1, want_post);
    const QcShape sh = pick_shape(q, per_slot);
    const size_t lds = qc_lds_bytes<T>(q, RULE, sh.F, want_post);
    if ((size_t)S * per_slot > kLdsLimit || lds > kLdsLimit || sh.threads > 512) { *supported = false; return hipSuccess; }
    *supported = true;
    const int F = sh.F;
    size_t per_cu = kLdsLimit / lds;
    const size_t wave_cap = 32 / (sh.threads / 64);
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu < 1) per_cu = 1;
    const long long groups = (io.frames + F - 1) / F;
    long long grid = (long long)num_cus * (long long)per_cu * 4;
    if (grid > groups) grid = groups;
    if (grid < 1) grid = 1;
    auto k = decode_qc_kernel<RULE, T, TL, DC, DV>;
    hipError_t e = kernel_setup(reinterpret_cast<const void *>(k), sh.threads, lds, nullptr);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(sh.threads), lds, io.stream, q, make_cfg<T>(cfg), F, sh.W,
                       reinterpret_cast<const TL *>(io.llr), io.frames, io.hard, io.iters,
                       io.converged, reinterpret_cast<T *>(io.posterior));
    *name = sizeof(T) == 8 ? "qc_f64" : "qc_f32";
    return hipGetLastError();
}

template <int RULE, typename T, typename TL>
static hipError_t launch_qc_deg(const QcDev &q, const ldpc_decode_config &cfg, const Launch &io,
                                int num_cus, const char **name, bool *supported)
{
    if (q.max_dc <= 8 && q.max_dv <= 12)
        return launch_qc_t<RULE, T, TL, 8, 12>(q, cfg, io, num_cus, name, supported);
    if (q.max_dc <= 10 && q.max_dv <= 24)
        return launch_qc_t<RULE, T, TL, 10, 24>(q, cfg, io, num_cus, name, supported);
    if (q.max_dc <= 20 && q.max_dv <= 32)
        return launch_qc_t<RULE, T, TL, 20, 32>(q, cfg, io, num_cus, name, supported);
    *supported = false;
    return hipSuccess;
}

template <int RULE>
static hipError_t launch_qc_rule(const QcDev &q, const ldpc_decode_config &cfg, int compute_dtype,
                                 const Launch &io, int num_cus, const char **name, bool *supported)
{
    if (compute_dtype == LDPC_F64) {
        if (io.llr_dtype == LDPC_F64)
            return launch_qc_deg<RULE, double, double>(q, cfg, io, num_cus, name, supported);
        return launch_qc_deg<RULE, double, float>(q, cfg, io, num_cus, name, supported);
    }
    if (io.llr_dtype == LDPC_F32)
        return launch_qc_deg<RULE, float, float>(q, cfg, io, num_cus, name, supported);
    *supported = false;  // float arithmetic on double LLRs: edge-list kernel converts on load
    return hipSuccess;
}

#ifndef LDPC_QC_RULE
#error "compile with -DLDPC_QC_RULE=<1..4 or 6>: one translation unit per rule"
#endif

#define LDPC_CAT2(a, b) a##b
#define LDPC_CAT(a, b) LDPC_CAT2(a, b)

hipError_t LDPC_CAT(launch_qc_rule_, LDPC_QC_RULE)(const QcDev &q, const ldpc_decode_config &cfg,
                                                   int compute_dtype, const Launch &io, int num_cus,
                                                   const char **name, bool *supported)
{
    return launch_qc_rule<LDPC_QC_RULE>(q, cfg, compute_dtype, io, num_cus, name, supported);
}

}  // namespace ldpc
