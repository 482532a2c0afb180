// Edge-list (CSR/CSC) flooding decoder: any parity-check matrix, rules R0-R4, float or double.
//
// One frame per workgroup.  Messages of the frame live in LDS when they fit (160 KiB per CU),
// otherwise in a per-workgroup slice of a global workspace; channel LLRs, posteriors and hard
// bits live in LDS too unless even they do not fit (NR BG1 Z = 384 in double), in which case the
// whole state of the frame goes to the workspace.  Threads stride over checks in the check-node phase and over
// variables in the variable-node phase; the edge lists are read from global memory (shared by
// every frame, so they stay in L2).  This is the general path and the one that honours the
// reference's loop-bound quirks (num_active_checks, num_refresh_vars); codes with circulant
// structure normally take decode_qc.hip instead.
//
// Per iteration (reference: ldpc.py:161-207, bpsk.py:200-248):
//   1. syndrome of the current hard decisions; all-zero => the frame leaves the loop
//   2. check nodes: read variable-to-check messages (R0: posteriors), write check-to-variable
//   3. variable nodes: sum check-to-variable messages in ascending check order, write the
//      posterior, the hard bit and the new variable-to-check messages
#include "bp_common.h"
#include "bp_math.h"

namespace ldpc {

template <int RULE, typename T, typename TL, int WHERE>  // 0: all state in LDS, 1: messages in the
                                                         // global workspace, 2: everything there
__global__ void decode_csr_kernel(CsrDev c, Cfg<T> g, const TL *__restrict__ llr, long long frames,
                                  uint8_t *__restrict__ hard_out, int *__restrict__ iters_out,
                                  uint8_t *__restrict__ conv_out, T *__restrict__ post_out,
                                  T *__restrict__ gws)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, NT = blockDim.x;
    const int n = c.n, m = c.m, E = c.E;
    const int mact = (g.num_active_checks < 0 || g.num_active_checks > m) ? m : g.num_active_checks;
    const int nref = (g.num_refresh_vars < 0 || g.num_refresh_vars > n) ? n : g.num_refresh_vars;

    const size_t msg_elems = (size_t)(RULE == 0 ? E : 2 * E);
    T *L, *post, *c2v, *v2c;
    uint8_t *hb;
    if (WHERE == 0) {
        L = reinterpret_cast<T *>(smem);
        post = L + n;
        c2v = post + n;
        v2c = c2v + E;  // unused by R0
        hb = reinterpret_cast<uint8_t *>(c2v + msg_elems);
    } else if (WHERE == 1) {
        L = reinterpret_cast<T *>(smem);
        post = L + n;
        hb = reinterpret_cast<uint8_t *>(post + n);
        c2v = gws + (size_t)blockIdx.x * msg_elems;
        v2c = c2v + E;
    } else {
        // per workgroup: messages | L | post | hard bytes (rounded up to whole T)
        const size_t per_wg = msg_elems + 2 * (size_t)n + ((size_t)n + sizeof(T) - 1) / sizeof(T);
        c2v = gws + (size_t)blockIdx.x * per_wg;
        v2c = c2v + E;
        L = c2v + msg_elems;
        post = L + n;
        hb = reinterpret_cast<uint8_t *>(post + n);
    }

    for (long long f = blockIdx.x; f < frames; f += gridDim.x) {
        // ---- load channel LLRs (ldpc.py:151-158, bpsk.py:190-196)
        for (int i = tid; i < n; i += NT) {
            T l = (T)llr[f * n + i];
            L[i] = l;
            post[i] = l;
            hb[i] = (uint8_t)hard_bit<RULE>(l);
        }
        __syncthreads();
        for (int e = tid; e < E; e += NT) {
            c2v[e] = (T)0;  // checks that are never visited (inactive rows, degree < 2) contribute 0
            if (RULE != 0) v2c[e] = L[c.col_idx[e]];  // bpsk.py:172-177
        }
        __syncthreads();

        int t = 0, converged = 0, special = 0;
        for (; t < g.max_iters; ++t) {
            // ---- 1. syndrome (ldpc.py:76-80)
            if (g.early_exit) {
                int bad = 0;
                const bool float_check = RULE == 0 && t == 0 && g.float_syndrome0;
                for (int j = tid; j < m; j += NT) {
                    const int e0 = c.row_ptr[j], e1 = c.row_ptr[j + 1];
                    if (float_check) {
                        // numpy.dot(H, llr) % 2 == 0 on the float LLRs (ldpc.py:158,165)
                        T s = (T)0;
                        for (int e = e0; e < e1; ++e) s += post[c.col_idx[e]];
                        bad |= !even_integer(s);
                    } else {
                        unsigned p = 0;
                        for (int e = e0; e < e1; ++e) p ^= hb[c.col_idx[e]];
                        bad |= (int)p;
                    }
                }
                if (!__syncthreads_or(bad)) { converged = 1; special = float_check; break; }
            }
            // ---- 2. check nodes
            for (int j = tid; j < (RULE == 0 ? mact : m); j += NT) {
                const int e0 = c.row_ptr[j], d = c.row_ptr[j + 1] - e0;
                if (RULE <= 1 || RULE == 6) {
                    if (d < 2) continue;  // ldpc.py:178, bpsk.py:216
                    MinSumState<T> s;
                    s.init();
                    for (int p = 0; p < d; ++p)
                        s.add(p, RULE == 0 ? post[c.col_idx[e0 + p]] : v2c[e0 + p]);
                    for (int p = 0; p < d; ++p)
                        c2v[e0 + p] = RULE == 0 ? r0_message(s, p)
                                      : (RULE == 6 ? oms_message(s, p, g.scale, g.offset) : r1_message(s, p, g.scale));
                } else if (RULE == 2 || RULE == 3) {
                    // product of the other tanh values as prefix * suffix
                    for (int p = 0; p < d; ++p) c2v[e0 + p] = spa_tanh_half(v2c[e0 + p], g.clip_tanh_arg);
                    T suf = (T)1;
                    for (int p = d - 1; p >= 0; --p) { v2c[e0 + p] = suf; suf *= c2v[e0 + p]; }
                    T pre = (T)1;
                    for (int p = 0; p < d; ++p) {
                        T th = c2v[e0 + p];
                        c2v[e0 + p] = spa_message(pre * v2c[e0 + p], g.clip_prod, g.scale, RULE == 3);
                        pre *= th;
                    }
                } else {
                    // sum of the other phi values as prefix + suffix; sign from the negatives
                    unsigned long long neg = 0ull;
                    for (int p = 0; p < d; ++p) {
                        T q = v2c[e0 + p];
                        neg |= (unsigned long long)(q < (T)0) << p;  // channel.py:332-337
                        c2v[e0 + p] = phi(t_abs(q));
                    }
                    T suf = (T)0;
                    for (int p = d - 1; p >= 0; --p) { v2c[e0 + p] = suf; suf += c2v[e0 + p]; }
                    T pre = (T)0;
                    for (int p = 0; p < d; ++p) {
                        T ph = c2v[e0 + p];
                        bool negative = (__popcll(neg & ~(1ull << p)) & 1) != 0;
                        c2v[e0 + p] = phi_message(pre + v2c[e0 + p], negative);
                        pre += ph;
                    }
                }
            }
            __syncthreads();
            // ---- 3. variable nodes
            for (int i = tid; i < n; i += NT) {
                const int k0 = c.col_ptr[i], k1 = c.col_ptr[i + 1];
                T b;
                if (RULE == 0) {
                    T s = (T)0;  // messagesReceivedByBits, ascending check order (ldpc.py:173,188)
                    for (int k = k0; k < k1; ++k) s += c2v[c.csc_edge[k]];
                    b = g.scale * (s + L[i]);  // ldpc.py:196
                } else {
                    // ext_k = L + (sum over the other checks, ascending) (bpsk.py:244-247); the
                    // part of that sum before k is the running total itself
                    const T l = L[i];
                    T run = (T)0;
                    for (int k = k0; k < k1; ++k) {
                        T ext = run;
                        for (int kk = k + 1; kk < k1; ++kk) ext += c2v[c.csc_edge[kk]];
                        run += c2v[c.csc_edge[k]];
                        v2c[c.csc_edge[k]] = i < nref ? l + ext : l;
                    }
                    b = l + run;  // bpsk.py:237
                    if (g.clip_post > (T)0) b = clip(b, -g.clip_post, g.clip_post);
                }
                post[i] = b;
                hb[i] = (uint8_t)hard_bit<RULE>(b);
            }
            __syncthreads();
        }
        if (!converged && !g.early_exit) {
            int bad = 0;
            for (int j = tid; j < m; j += NT) {
                unsigned p = 0;
                for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; ++e) p ^= hb[c.col_idx[e]];
                bad |= (int)p;
            }
            converged = !__syncthreads_or(bad);
        }
        // ---- results (ldpc.py:166-170 on success, ldpc.py:210-216 on failure)
        for (int i = tid; i < n; i += NT) {
            uint8_t h = hb[i];
            if (special) {  // the reference hands back the LLR floats themselves
                T v = post[i];
                h = v == (T)0 ? 0 : (v == (T)1 ? 1 : 2);
            }
            hard_out[f * n + i] = h;
            if (post_out) post_out[f * n + i] = post[i];
        }
        if (tid == 0) {
            iters_out[f] = converged ? t : (g.early_exit ? g.max_iters - 1 : g.max_iters);
            conv_out[f] = (uint8_t)converged;
        }
        __syncthreads();
    }
}

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

template <int RULE, typename T, typename TL>
static hipError_t launch_csr_t(const CsrDev &code, const ldpc_decode_config &cfg, const Launch &io,
                               int num_cus, Scratch &workspace, const char **kernel_name)
{
    const size_t msgs = (size_t)(RULE == 0 ? 1 : 2) * code.E * sizeof(T);
    const size_t base = (size_t)2 * code.n * sizeof(T) + (size_t)code.n + 16;
    const int where = base + msgs <= kLdsLimit ? 0 : (base <= kLdsLimit ? 1 : 2);
    const size_t lds = where == 0 ? base + msgs : (where == 1 ? base : 16);
    int nt = 64;
    const int widest = code.n > code.m ? code.n : code.m;
    while (nt < 512 && nt * 2 <= widest) nt *= 2;
    const size_t per_cu = kLdsLimit / lds;
    long long resident = (long long)num_cus * (long long)(per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu));
    long long grid = io.frames < resident * 4 ? io.frames : resident * 4;
    if (where != 0) grid = io.frames < resident ? io.frames : resident;
    if (grid < 1) grid = 1;
    T *gws = nullptr;
    if (where != 0) {
        const size_t hard_elems = ((size_t)code.n + sizeof(T) - 1) / sizeof(T);
        const size_t per_wg = where == 1 ? msgs : msgs + ((size_t)2 * code.n + hard_elems) * sizeof(T);
        hipError_t e = scratch_reserve(workspace, per_wg, resident, grid);
        if (e != hipSuccess) return e;
        gws = reinterpret_cast<T *>(workspace.ptr);
    }
    Cfg<T> g = make_cfg<T>(cfg);
    auto go = [&](auto k, const char *n64, const char *n32) -> hipError_t {
        hipError_t e = kernel_setup(reinterpret_cast<const void *>(k), nt, lds, nullptr);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(nt), lds, io.stream, code, g,
                           reinterpret_cast<const TL *>(io.llr), io.frames, io.hard, io.iters,
                           io.converged, reinterpret_cast<T *>(io.posterior), gws);
        *kernel_name = sizeof(T) == 8 ? n64 : n32;
        return hipGetLastError();
    };
    if (where == 0) return go(decode_csr_kernel<RULE, T, TL, 0>, "csr_lds_f64", "csr_lds_f32");
    if (where == 1) return go(decode_csr_kernel<RULE, T, TL, 1>, "csr_global_f64", "csr_global_f32");
    return go(decode_csr_kernel<RULE, T, TL, 2>, "csr_global_all_f64", "csr_global_all_f32");
}

template <typename T, typename TL>
static hipError_t launch_csr_rule(const CsrDev &code, const ldpc_decode_config &cfg, const Launch &io,
                                  int num_cus, Scratch &ws, const char **name)
{
    switch (cfg.rule) {
        case 0: return launch_csr_t<0, T, TL>(code, cfg, io, num_cus, ws, name);
        case 1: return launch_csr_t<1, T, TL>(code, cfg, io, num_cus, ws, name);
        case 2: return launch_csr_t<2, T, TL>(code, cfg, io, num_cus, ws, name);
        case 3: return launch_csr_t<3, T, TL>(code, cfg, io, num_cus, ws, name);
        case 4: return launch_csr_t<4, T, TL>(code, cfg, io, num_cus, ws, name);
        case 6: return launch_csr_t<6, T, TL>(code, cfg, io, num_cus, ws, name);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_csr(const CsrDev &code, const ldpc_decode_config &cfg, int compute_dtype,
                      const Launch &io, int num_cus, Scratch &workspace, const char **kernel_name)
{
    if (compute_dtype == LDPC_F64) {
        return io.llr_dtype == LDPC_F64
                   ? launch_csr_rule<double, double>(code, cfg, io, num_cus, workspace, kernel_name)
                   : launch_csr_rule<double, float>(code, cfg, io, num_cus, workspace, kernel_name);
    }
    return io.llr_dtype == LDPC_F64
               ? launch_csr_rule<float, double>(code, cfg, io, num_cus, workspace, kernel_name)
               : launch_csr_rule<float, float>(code, cfg, io, num_cus, workspace, kernel_name);
}

}  // namespace ldpc
