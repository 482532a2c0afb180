// Row-layered normalised min-sum (build extension, see LDPC_RULE_LNMS in ldpc_amd.h) for ANY
// circulant code without double-circulant blocks: the base graph comes from device tables, so a
// caller's own NR base-graph table runs here without being compiled in.  Arithmetic, schedule and
// results are those of the specialised kernel (decode_lnms.hip) and of the CPU restatement.
//
// F frames ("slots") per workgroup, thread (f, z) owns check (br, z) of every block row.
//   P [nb][F*Z]        posteriors of all variables, LDS
//   A, B, W [mb][F*Z]  compressed previous messages of each check (A = scale * m1, B = scale * m2,
//                      W = argmin position << 24 | sign parity << 31 | one sign bit per edge), in
//                      LDS when P and they are small enough for five workgroups per compute unit,
//                      else in a per-workgroup slice of the handle's global workspace
//                      (thread-private, coalesced, L2-resident)
// One layer = one block row; its degree selects an unrolled body through a wave-uniform switch.
// A variable of degree one sends its channel LLR (re-read from the LLR array); its posterior
// L + r is still written to P so that syndrome and output treat every column alike.
#include <cstdlib>

#include "spec_common.h"

namespace ldpc {

typedef const unsigned __attribute__((address_space(4))) *lc_u32;
typedef const int __attribute__((address_space(4))) *lc_i32;
__device__ __forceinline__ lc_u32 lc(const unsigned *p) { return (lc_u32)(unsigned long long)p; }
__device__ __forceinline__ lc_i32 lc(const int *p) { return (lc_i32)(unsigned long long)p; }

constexpr int kMaxLayerDegree = 24;  // sign bits and argmin position share one 32-bit word

template <typename T>
struct LnState {  // where a thread finds the state of check (br, z): element index br * FZ + tid
    T *A, *B;
    unsigned *W;
};

template <typename T, typename TL, int D>
__device__ __forceinline__ void qc_layer(const QcDev &q, const Cfg<T> &g, int br, int e0, int FZ, int fz0, int z,
                                         T *P, const LnState<T> &st, int sidx, const TL *__restrict__ lf)
{
    const lc_u32 info = lc(q.row_info);
    const lc_i32 cptr = lc(q.col_ptr);
    const int Z = q.Z;
    T Ao = st.A[sidx], Bo = st.B[sidx];
    const unsigned w = st.W[sidx];
    const int idx_old = (int)((w >> 24) & 31u);
    const unsigned wp = w ^ (unsigned)((int)w >> 31);  // bit D-1-p: sign of the old message on edge p
    T qv[D];
    int off[D];
    T m1 = Lim<T>::inf(), m2 = Lim<T>::inf();
    unsigned sq = 0;
#pragma unroll
    for (int p = 0; p < D; ++p) {
        const unsigned ie = info[e0 + p];
        const int bc = (int)(ie >> 16), s = (int)(ie & 0xfffu);
        int zz = z + s;
        zz -= zz >= Z ? Z : 0;
        off[p] = bc * FZ + fz0 + zz;
        if (cptr[bc + 1] - cptr[bc] == 1) {  // wave-uniform: a variable of degree one sends its channel LLR
            qv[p] = (T)lf[bc * Z + zz] + (T)0;
        } else {
            const T mag = idx_old == p ? Bo : Ao;
            qv[p] = P[off[p]] - SNum<T>::with_sign(mag, wp << (31 - (D - 1 - p)));
        }
        m2 = SNum<T>::second(m1, m2, qv[p]);
        m1 = SNum<T>::min_abs(m1, qv[p]);
        sq = (sq << 1) | (SNum<T>::sign_word(qv[p]) >> 31);
    }
    const unsigned px = (unsigned)__popc(sq) << 31;  // sign parity of the inputs
    const T An = g.scale * m1, Bn = g.scale * m2;
    const T Anp = SNum<T>::flip_sign(An, px), Bnp = SNum<T>::flip_sign(Bn, px);
    unsigned minbits = 0;
#pragma unroll
    for (int pp = 0; pp < D; ++pp) {
        const int p = D - 1 - pp;
        const T mag = SNum<T>::pick_min(qv[p], m1, Anp, Bnp, minbits);
        P[off[p]] = qv[p] + SNum<T>::flip_sign(mag, SNum<T>::sign_word(qv[p]) & 0x80000000u);
    }
    st.A[sidx] = An;
    st.B[sidx] = Bn;
    st.W[sidx] = sq | ((unsigned)__builtin_ctz(minbits) << 24) | px;
}

template <typename T, typename TL, bool STATE_IN_LDS>
__global__ void qc_lnms_kernel(QcDev q, Cfg<T> g, const TL *__restrict__ llr, long long frames,
                               uint8_t *__restrict__ hard_out, int *__restrict__ iters_out,
                               uint8_t *__restrict__ conv_out, T *__restrict__ post_out,
                               unsigned char *__restrict__ ws, unsigned long long ws_per_wg, int F)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int Z = q.Z, mb = q.mb, nb = q.nb, FZ = F * Z, n = nb * Z;
    const int tid = threadIdx.x;
    const bool lane_on = tid < FZ;
    const int f = lane_on ? tid / Z : 0;
    const int z = lane_on ? tid - f * Z : 0;
    const int fz0 = f * Z;
    const bool single = blockDim.x <= 64;
    const lc_i32 rptr = lc(q.row_ptr);
    const lc_u32 info = lc(q.row_info);

    T *P = reinterpret_cast<T *>(smem);  // [nb][FZ]
    unsigned char *after = smem + (size_t)nb * FZ * sizeof(T);
    LnState<T> st;
    if (STATE_IN_LDS) {
        st.A = reinterpret_cast<T *>(after);
        st.B = st.A + (size_t)mb * FZ;
        st.W = reinterpret_cast<unsigned *>(st.B + (size_t)mb * FZ);
        after = reinterpret_cast<unsigned char *>(st.W + (size_t)mb * FZ);
    } else {
        unsigned char *mine = ws + (unsigned long long)blockIdx.x * ws_per_wg;
        st.A = reinterpret_cast<T *>(mine);
        st.B = st.A + (size_t)mb * FZ;
        st.W = reinterpret_cast<unsigned *>(st.B + (size_t)mb * FZ);
    }
    int *slot_bad = reinterpret_cast<int *>(after);  // [F]

    const long long groups = (frames + F - 1) / F;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long frame = grp * F + f;
        const bool active = lane_on && frame < frames;
        const TL *__restrict__ lf = llr + (frame < frames ? frame : frames - 1) * n;  // loads never depend on `active`
        if (lane_on) {
            for (int bc = 0; bc < nb; ++bc) P[bc * FZ + tid] = (T)lf[bc * Z + z] + (T)0;  // -0.0 -> +0.0
            for (int br = 0; br < mb; ++br) {
                st.A[br * FZ + tid] = (T)0;
                st.B[br * FZ + tid] = (T)0;
                st.W[br * FZ + tid] = 0u;
            }
        }
        if (tid < F) slot_bad[tid] = 0;
        __syncthreads();

        bool done = !active;
        int res_iters = g.early_exit ? g.max_iters - 1 : g.max_iters, res_conv = 0;
        for (int t = 0;; ++t) {
            const bool last = t == g.max_iters;
            if (g.early_exit || last) {
                if (last && g.early_exit) break;  // no test after the last sweep
                // ---- syndrome of the hard decisions `0 if P > 0 else 1`
                unsigned bad = 0;
                if (!done) {
                    volatile int *slot_bad_now = slot_bad;  // polled while other waves may still be writing
                    for (int br = 0; br < mb; ++br) {
                        // a failed test needs no more rows: every four block rows the wave stops if each
                        // of its running slots has a failure on record (its own lanes' included)
                        if ((br & 3) == 2) {
                            if (bad) slot_bad_now[f] = 1;
                            if (__ballot(slot_bad_now[f] == 0) == 0ull) break;
                        }
                        unsigned par = 0;
                        for (int e = rptr[br]; e < rptr[br + 1]; ++e) {
                            const unsigned ie = info[e];
                            int zz = z + (int)(ie & 0xfffu);
                            zz -= zz >= Z ? Z : 0;
                            par ^= hard_bit<1>(P[(int)(ie >> 16) * FZ + fz0 + zz]);
                        }
                        bad |= par;
                    }
                }
                if (bad) slot_bad[f] = 1;
                __syncthreads();
                const bool slot_ok = slot_bad[f] == 0;
                bool any_bad = false;
                for (int ff = 0; ff < F; ++ff) any_bad |= slot_bad[ff] != 0;
                if (last) {
                    if (!done) res_conv = slot_ok ? 1 : 0;
                    break;
                }
                if (!done && slot_ok) { res_iters = t; res_conv = 1; done = true; }
                if (!any_bad) break;  // every slot that was still running has just converged
                __syncthreads();
                if (tid < F) slot_bad[tid] = 0;
            }
            // ---- one sweep over the layers
            for (int br = 0; br < mb; ++br) {
                const int e0 = rptr[br], d = rptr[br + 1] - e0;
                if (d >= 2 && !done) {
                    const int sidx = br * FZ + tid;
                    switch (d) {
#define LDPC_LAYER_CASE(N) case N: qc_layer<T, TL, N>(q, g, br, e0, FZ, fz0, z, P, st, sidx, lf); break;
                        LDPC_LAYER_CASE(2) LDPC_LAYER_CASE(3) LDPC_LAYER_CASE(4) LDPC_LAYER_CASE(5)
                        LDPC_LAYER_CASE(6) LDPC_LAYER_CASE(7) LDPC_LAYER_CASE(8) LDPC_LAYER_CASE(9)
                        LDPC_LAYER_CASE(10) LDPC_LAYER_CASE(11) LDPC_LAYER_CASE(12) LDPC_LAYER_CASE(13)
                        LDPC_LAYER_CASE(14) LDPC_LAYER_CASE(15) LDPC_LAYER_CASE(16) LDPC_LAYER_CASE(17)
                        LDPC_LAYER_CASE(18) LDPC_LAYER_CASE(19) LDPC_LAYER_CASE(20) LDPC_LAYER_CASE(21)
                        LDPC_LAYER_CASE(22) LDPC_LAYER_CASE(23) LDPC_LAYER_CASE(24)
#undef LDPC_LAYER_CASE
                        default: break;
                    }
                }
                if (!single) __syncthreads();
            }
            if (single) __syncthreads();  // one wave: LDS operations are in order; this only keeps the flags tidy
        }
        if (active) {
            for (int bc = 0; bc < nb; ++bc) {
                const T v = P[bc * FZ + tid];
                hard_out[frame * n + bc * Z + z] = (uint8_t)hard_bit<1>(v);
                if (post_out) post_out[frame * n + bc * Z + z] = v;
            }
            if (z == 0) { iters_out[frame] = res_iters; conv_out[frame] = (uint8_t)res_conv; }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ host
template <typename T, typename TL>
static hipError_t launch_qc_lnms_t(const QcDev &code, const ldpc_decode_config &cfg, const Launch &io,
                                   int num_cus, Scratch &workspace, const char **name, bool *supported)
{
    *supported = false;
    if (code.max_dc > kMaxLayerDegree) return hipSuccess;
    const int Z = code.Z;
    // frames per workgroup: whole frames side by side up to 128 threads for small Z
    const int F = Z >= 64 ? 1 : (128 / Z < 1 ? 1 : 128 / Z);
    const size_t FZ = (size_t)F * Z;
    const size_t p_bytes = (size_t)code.nb * FZ * sizeof(T);
    const size_t st_bytes = (size_t)code.mb * FZ * (2 * sizeof(T) + 4);
    const size_t flags = (FZ / Z * sizeof(int) + 63) / 64 * 64;  // one syndrome flag per slot
    if (p_bytes + flags > 150 * 1024) return hipSuccess;  // posteriors must fit LDS
    // state beside the posteriors only while that still leaves room for five workgroups per compute unit
    const bool in_lds = p_bytes + st_bytes + flags <= 32 * 1024;
    const size_t lds = p_bytes + (in_lds ? st_bytes : 0) + flags;
    const int threads = (int)((FZ + 63) / 64 * 64);
    if (threads > 1024) return hipSuccess;
    *supported = true;

    Cfg<T> g;
    g.rule = LDPC_RULE_LNMS; g.max_iters = cfg.max_iters; g.early_exit = cfg.early_exit;
    g.float_syndrome0 = 0; g.num_active_checks = -1; g.num_refresh_vars = -1;
    g.scale = (T)cfg.scale; g.clip_tanh_arg = (T)0; g.clip_prod = (T)0; g.clip_post = (T)0; g.offset = (T)0;
    auto launch = [&](auto k) -> hipError_t {
        int per_cu = 0;
        hipError_t e = kernel_setup(reinterpret_cast<const void *>(k), threads, lds, &per_cu);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        const long long groups = (io.frames + F - 1) / F;
        long long grid = (long long)num_cus * per_cu * 2;
        if (grid > groups) grid = groups;
        if (grid < 1) grid = 1;
        unsigned char *ws = nullptr;
        if (!in_lds) {
            e = scratch_reserve(workspace, st_bytes, (long long)num_cus * per_cu * 2, grid);
            if (e != hipSuccess) return e;
            ws = reinterpret_cast<unsigned char *>(workspace.ptr);
        }
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(threads), lds, io.stream, code, g,
                           reinterpret_cast<const TL *>(io.llr), io.frames, io.hard, io.iters, io.converged,
                           reinterpret_cast<T *>(io.posterior), ws, (unsigned long long)st_bytes, F);
        return hipGetLastError();
    };
    *name = sizeof(T) == 8 ? (in_lds ? "qc_lnms_f64" : "qc_lnms_global_f64") : (in_lds ? "qc_lnms_f32" : "qc_lnms_global_f32");
    return in_lds ? launch(qc_lnms_kernel<T, TL, true>) : launch(qc_lnms_kernel<T, TL, false>);
}

hipError_t launch_qc_lnms(const QcDev &code, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                          int num_cus, Scratch &workspace, const char **name, bool *supported)
{
    if (compute_dtype == LDPC_F64) {
        return io.llr_dtype == LDPC_F64
                   ? launch_qc_lnms_t<double, double>(code, cfg, io, num_cus, workspace, name, supported)
                   : launch_qc_lnms_t<double, float>(code, cfg, io, num_cus, workspace, name, supported);
    }
    return io.llr_dtype == LDPC_F64
               ? launch_qc_lnms_t<float, double>(code, cfg, io, num_cus, workspace, name, supported)
               : launch_qc_lnms_t<float, float>(code, cfg, io, num_cus, workspace, name, supported);
}

}  // namespace ldpc
