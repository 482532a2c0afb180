// Host side of the C ABI declared in include/ldpc_amd.h: handle management, argument checks,
// kernel selection.  No decode arithmetic happens on the host; without a HIP device every compute
// entry point fails with LDPC_ERR_NO_DEVICE / LDPC_ERR_HIP.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <dlfcn.h>

#include <memory>
#include <array>
#include <vector>

#include "bp_common.h"
#include "extra_kernels.h"
#include "sweep_fused.h"

using namespace ldpc;

static thread_local std::string g_err;
static thread_local const char *g_kernel = "";

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(LDPC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

struct ldpc_code {
    int device = 0, num_cus = 0;
    int m = 0, n = 0, E = 0;
    int max_dc = 0, max_dv = 0;
    std::vector<int> row_ptr, col_idx, col_ptr, csc_edge;
    int *d_ints = nullptr;  // one allocation: row_ptr | col_idx | col_ptr | csc_edge
    CsrDev csr{};
    bool has_qc = false;
    bool qc_pairs = false;  // some block is the sum of two circulants
    int spec_id = -1;       // compiled-in specialisation of this code, if any
    QcDev qc{};
    int *d_qc = nullptr;    // one allocation: row_ptr | row_info | col_ptr | col_info
    BgHandle bg;            // run-time tables of the base-graph layered kernel, if the description matches one
    // scratch of the kernels that keep frame state in global memory, one buffer per stream a decode
    // was enqueued on: launches on different streams may run at the same time and must not share slices
    std::vector<std::pair<hipStream_t, Scratch>> workspaces;
    // device staging of ldpc_decode_batch_host (inputs and results of one call); grows, never shrinks
    char *host_stage = nullptr;
    size_t host_stage_bytes = 0;
    std::vector<char> host_bounce;  // host side of the small-result copy of ldpc_decode_batch_host
    Scratch &workspace_for(hipStream_t stream)
    {
        for (size_t i = 0; i < workspaces.size(); ++i)
            if (workspaces[i].first == stream) {
                if (i + 1 != workspaces.size()) std::swap(workspaces[i], workspaces.back());  // most recent last
                return workspaces.back().second;
            }
        // a caller that creates and destroys streams must not grow the list without bound: beyond eight
        // streams the buffer of the one used longest ago goes (hipFree waits for the device first)
        if (workspaces.size() >= 8) {
            if (workspaces.front().second.ptr) (void)hipFree(workspaces.front().second.ptr);
            workspaces.erase(workspaces.begin());
        }
        workspaces.emplace_back(stream, Scratch{});
        return workspaces.back().second;
    }
};

// Scratch of the allocation-free entry point: one array per device, shared by every call on that
// device (it keeps the last call's contents; the kernels initialise what they read).
constexpr size_t kNoallocScratchBytes = LDPC_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kNoallocScratchBytes];
constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static void *g_scratch_ptr[kMaxDevices] = {};  // device address of `scratch` on each device, once looked up

// Device address of this device's copy of `scratch` (the device must be current).
static int noalloc_scratch(int device, Scratch *out)
{
    if (device < 0 || device >= kMaxDevices) return fail(LDPC_ERR_UNSUPPORTED, "device index %d beyond %d", device, kMaxDevices - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) return fail(LDPC_ERR_HIP, "hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
        g_scratch_ptr[device] = ptr;
    }
    out->ptr = g_scratch_ptr[device];
    out->bytes = kNoallocScratchBytes;
    out->fixed = true;
    return LDPC_OK;
}

extern "C" const char *ldpc_last_error(void) { return g_err.c_str(); }
extern "C" const char *ldpc_last_kernel(void) { return g_kernel; }
extern "C" int32_t ldpc_abi_version(void) { return 3; }

extern "C" int32_t ldpc_device_count(int32_t *count)
{
    if (!count) return fail(LDPC_ERR_ARG, "count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0) {
        *count = 0;
        return fail(LDPC_ERR_NO_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e));
    }
    *count = c;
    return LDPC_OK;
}

extern "C" int32_t ldpc_code_create_csr(int32_t m, int32_t n, const int32_t *row_ptr,
                                        const int32_t *col_idx, ldpc_code **out)
{
    if (!out) return fail(LDPC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (m <= 0 || n <= 0 || !row_ptr || !col_idx) return fail(LDPC_ERR_ARG, "empty matrix or NULL edge list");
    if (row_ptr[0] != 0) return fail(LDPC_ERR_ARG, "row_ptr[0] must be 0");
    for (int j = 0; j < m; ++j)
        if (row_ptr[j + 1] < row_ptr[j]) return fail(LDPC_ERR_ARG, "row_ptr not monotone at row %d", j);
    const int E = row_ptr[m];
    if (E <= 0) return fail(LDPC_ERR_ARG, "matrix has no edges");
    int max_dc = 0;
    std::vector<int> deg_v(n, 0);
    for (int j = 0; j < m; ++j) {
        max_dc = std::max(max_dc, row_ptr[j + 1] - row_ptr[j]);
        for (int e = row_ptr[j]; e < row_ptr[j + 1]; ++e) {
            if (col_idx[e] < 0 || col_idx[e] >= n) return fail(LDPC_ERR_ARG, "column %d out of range in row %d", col_idx[e], j);
            if (e > row_ptr[j] && col_idx[e] <= col_idx[e - 1])
                return fail(LDPC_ERR_ARG, "columns of row %d are not strictly ascending", j);
            deg_v[col_idx[e]]++;
        }
    }
    if (max_dc > kMaxCheckDegree)
        return fail(LDPC_ERR_UNSUPPORTED, "check degree %d exceeds the supported maximum %d", max_dc, kMaxCheckDegree);
    int count = 0;
    if (ldpc_device_count(&count) != LDPC_OK) return LDPC_ERR_NO_DEVICE;

    ldpc_code *c = new ldpc_code;
    c->m = m; c->n = n; c->E = E; c->max_dc = max_dc;
    c->row_ptr.assign(row_ptr, row_ptr + m + 1);
    c->col_idx.assign(col_idx, col_idx + E);
    c->col_ptr.assign(n + 1, 0);
    for (int i = 0; i < n; ++i) { c->col_ptr[i + 1] = c->col_ptr[i] + deg_v[i]; c->max_dv = std::max(c->max_dv, deg_v[i]); }
    c->csc_edge.resize(E);
    {   // rows are visited in ascending order, so each variable's list comes out in ascending check order
        std::vector<int> fill(c->col_ptr.begin(), c->col_ptr.end() - 1);
        for (int j = 0; j < m; ++j)
            for (int e = row_ptr[j]; e < row_ptr[j + 1]; ++e) c->csc_edge[fill[col_idx[e]]++] = e;
    }
    hipError_t e = hipGetDevice(&c->device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, c->device);
    if (e != hipSuccess) { delete c; return fail(LDPC_ERR_HIP, "device query: %s", hipGetErrorString(e)); }
    c->num_cus = prop.multiProcessorCount;
    const size_t total = (size_t)(m + 1) + E + (n + 1) + E;
    e = hipMalloc(reinterpret_cast<void **>(&c->d_ints), total * sizeof(int));
    if (e != hipSuccess) { delete c; return fail(LDPC_ERR_HIP, "hipMalloc(edge lists): %s", hipGetErrorString(e)); }
    std::vector<int> host(total);
    int *p = host.data();
    std::copy(c->row_ptr.begin(), c->row_ptr.end(), p); p += m + 1;
    std::copy(c->col_idx.begin(), c->col_idx.end(), p); p += E;
    std::copy(c->col_ptr.begin(), c->col_ptr.end(), p); p += n + 1;
    std::copy(c->csc_edge.begin(), c->csc_edge.end(), p);
    e = hipMemcpy(c->d_ints, host.data(), total * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(c->d_ints); delete c; return fail(LDPC_ERR_HIP, "hipMemcpy(edge lists): %s", hipGetErrorString(e)); }
    c->csr.m = m; c->csr.n = n; c->csr.E = E;
    c->csr.row_ptr = c->d_ints;
    c->csr.col_idx = c->d_ints + (m + 1);
    c->csr.col_ptr = c->d_ints + (m + 1) + E;
    c->csr.csc_edge = c->d_ints + (m + 1) + E + (n + 1);
    *out = c;
    return LDPC_OK;
}

extern "C" int32_t ldpc_code_attach_qc(ldpc_code *c, int32_t mb, int32_t nb, int32_t Z, int32_t Eb,
                                       const int32_t *er, const int32_t *ec, const int32_t *es)
{
    if (!c || !er || !ec || !es) return fail(LDPC_ERR_ARG, "NULL argument");
    if (mb <= 0 || nb <= 0 || Z <= 0 || Eb <= 0) return fail(LDPC_ERR_ARG, "bad QC dimensions");
    if ((long long)mb * Z != c->m || (long long)nb * Z != c->n)
        return fail(LDPC_ERR_ARG, "QC dimensions %dx%d (Z=%d) do not match the %dx%d matrix", mb, nb, Z, c->m, c->n);
    if (Z > 4095 || Eb > 65535 || nb > 65535)
        return fail(LDPC_ERR_UNSUPPORTED, "QC description too large for the packed tables");
    if ((long long)Eb * Z != c->E) return fail(LDPC_ERR_ARG, "QC description has %lld edges, matrix has %d", (long long)Eb * Z, c->E);
    for (int e = 0; e < Eb; ++e) {
        if (er[e] < 0 || er[e] >= mb || ec[e] < 0 || ec[e] >= nb || es[e] < 0 || es[e] >= Z)
            return fail(LDPC_ERR_ARG, "base edge %d out of range", e);
        if (e > 0) {
            const bool ordered = er[e] > er[e - 1] || (er[e] == er[e - 1] && (ec[e] > ec[e - 1] || (ec[e] == ec[e - 1] && es[e] > es[e - 1])));
            if (!ordered) return fail(LDPC_ERR_ARG, "base edges must be sorted by (row, col, shift) without repeats");
        }
    }
    // the description must expand to exactly the matrix of the handle
    {
        std::vector<int> cols;
        for (int br = 0, e = 0; br < mb; ++br) {
            int e1 = e;
            while (e1 < Eb && er[e1] == br) ++e1;
            for (int z = 0; z < Z; ++z) {
                cols.clear();
                for (int k = e; k < e1; ++k) cols.push_back(ec[k] * Z + (z + es[k]) % Z);
                std::sort(cols.begin(), cols.end());
                const int j = br * Z + z;
                if ((int)cols.size() != c->row_ptr[j + 1] - c->row_ptr[j] ||
                    !std::equal(cols.begin(), cols.end(), c->col_idx.begin() + c->row_ptr[j]))
                    return fail(LDPC_ERR_ARG, "QC description does not expand to the matrix (check row %d)", j);
            }
            e = e1;
        }
    }
    std::vector<int> row_ptr(mb + 1, 0), col_ptr(nb + 1, 0);
    std::vector<unsigned> row_info(Eb), col_info(Eb), colr_info(Eb);
    for (int e = 0; e < Eb; ++e) { row_ptr[er[e] + 1]++; col_ptr[ec[e] + 1]++; }
    for (int i = 0; i < mb; ++i) row_ptr[i + 1] += row_ptr[i];
    for (int i = 0; i < nb; ++i) col_ptr[i + 1] += col_ptr[i];
    int max_dc = 0, max_dv = 0;
    for (int i = 0; i < mb; ++i) max_dc = std::max(max_dc, row_ptr[i + 1] - row_ptr[i]);
    for (int i = 0; i < nb; ++i) max_dv = std::max(max_dv, col_ptr[i + 1] - col_ptr[i]);
    for (int e = 0; e < Eb; ++e) row_info[e] = ((unsigned)ec[e] << 16) | (unsigned)es[e];
    {
        std::vector<int> fill(col_ptr.begin(), col_ptr.end() - 1);
        std::vector<int> last_row(nb, -1), run(nb, 0);
        c->qc_pairs = false;
        for (int e = 0; e < Eb; ++e) {  // ascending block row inside each column
            const int bc = ec[e], k = fill[bc]++;
            col_info[k] = ((unsigned)e << 16) | (unsigned)es[e];
            colr_info[k] = ((unsigned)er[e] << 16) | (unsigned)es[e];
            if (last_row[bc] == er[e]) {
                if (++run[bc] > 2)
                    return fail(LDPC_ERR_UNSUPPORTED, "block (%d,%d) is the sum of more than two circulants", er[e], bc);
                col_info[k - 1] |= 0x8000u;
                colr_info[k - 1] |= 0x8000u;
                c->qc_pairs = true;
            } else {
                run[bc] = 1;
            }
            last_row[bc] = er[e];
        }
    }
    HIP_TRY(hipSetDevice(c->device));
    const size_t total = (size_t)(mb + 1) + Eb + (nb + 1) + Eb + Eb;
    std::vector<int> host(total);
    int *p = host.data();
    std::copy(row_ptr.begin(), row_ptr.end(), p); p += mb + 1;
    std::memcpy(p, row_info.data(), Eb * sizeof(int)); p += Eb;
    std::copy(col_ptr.begin(), col_ptr.end(), p); p += nb + 1;
    std::memcpy(p, col_info.data(), Eb * sizeof(int)); p += Eb;
    std::memcpy(p, colr_info.data(), Eb * sizeof(int));
    if (c->d_qc) { (void)hipFree(c->d_qc); c->d_qc = nullptr; c->has_qc = false; }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_qc), total * sizeof(int)));
    HIP_TRY(hipMemcpy(c->d_qc, host.data(), total * sizeof(int), hipMemcpyHostToDevice));
    c->qc.mb = mb; c->qc.nb = nb; c->qc.Z = Z; c->qc.Eb = Eb;
    c->qc.row_ptr = c->d_qc;
    c->qc.row_info = reinterpret_cast<const unsigned *>(c->d_qc + (mb + 1));
    c->qc.col_ptr = c->d_qc + (mb + 1) + Eb;
    c->qc.col_info = reinterpret_cast<const unsigned *>(c->d_qc + (mb + 1) + Eb + (nb + 1));
    c->qc.colr_info = reinterpret_cast<const unsigned *>(c->d_qc + (mb + 1) + Eb + (nb + 1) + Eb);
    c->qc.max_dc = max_dc; c->qc.max_dv = max_dv;
    c->has_qc = true;
    c->spec_id = spec_match(Z, mb, nb, Eb, er, ec, es);
    if (c->bg.d_tab) { (void)hipFree(c->bg.d_tab); c->bg = BgHandle{}; }
    const int graph = c->qc_pairs ? -1 : bg_graph_match(Z, mb, nb, Eb, er, ec);
    if (graph >= 0) {
        HIP_TRY(bg_tables_create(graph, Z, mb, es, &c->bg.d_tab));
        c->bg.graph = graph; c->bg.Z = Z; c->bg.mb = mb; c->bg.nb = nb;
    }
    return LDPC_OK;
}

extern "C" int32_t ldpc_code_destroy(ldpc_code *c)
{
    if (!c) return LDPC_OK;
    if (c->d_ints) (void)hipFree(c->d_ints);
    if (c->d_qc) (void)hipFree(c->d_qc);
    if (c->bg.d_tab) (void)hipFree(c->bg.d_tab);
    for (auto &w : c->workspaces)
        if (w.second.ptr) (void)hipFree(w.second.ptr);
    if (c->host_stage) (void)hipFree(c->host_stage);
    delete c;
    return LDPC_OK;
}

extern "C" int32_t ldpc_code_info(const ldpc_code *c, int32_t *m, int32_t *n, int64_t *edges, int32_t *Z)
{
    if (!c) return fail(LDPC_ERR_ARG, "code is NULL");
    if (m) *m = c->m;
    if (n) *n = c->n;
    if (edges) *edges = c->E;
    if (Z) *Z = c->has_qc ? c->qc.Z : 0;
    return LDPC_OK;
}

static int check_cfg(const ldpc_decode_config *cfg)
{
    if (!cfg) return fail(LDPC_ERR_ARG, "cfg is NULL");
    if (cfg->rule < 0 || cfg->rule > LDPC_RULE_BF) return fail(LDPC_ERR_ARG, "rule %d is not one of R0..R4, LNMS, OMS or BF", cfg->rule);
    if (cfg->rule == LDPC_RULE_OMS && !(cfg->offset >= 0.0)) return fail(LDPC_ERR_ARG, "offset must not be negative");
    if (cfg->max_iters < 1 || cfg->max_iters > 100000) return fail(LDPC_ERR_ARG, "max_iters %d out of range", cfg->max_iters);
    if ((cfg->rule == 2 || cfg->rule == 3) && !(cfg->clip_prod > 0.0 && cfg->clip_prod <= 1.0))
        return fail(LDPC_ERR_ARG, "clip_prod must lie in (0, 1] for the tanh rules");
    return LDPC_OK;
}

// Shared body of ldpc_decode_batch and ldpc_decode_batch_noalloc; `noalloc` selects the scratch.
static int decode_impl(ldpc_code *c, const ldpc_decode_config *cfg, const void *llr_dev,
                       int32_t llr_dtype, int32_t compute_dtype, int32_t kernel,
                       int64_t frames, uint8_t *hard_dev, int32_t *iters_dev,
                       uint8_t *converged_dev, void *posterior_dev, void *stream, bool noalloc)
{
    if (!c) return fail(LDPC_ERR_ARG, "code is NULL");
    if (int rc = check_cfg(cfg)) return rc;
    if (frames < 0) return fail(LDPC_ERR_ARG, "negative frame count");
    if ((llr_dtype != LDPC_F32 && llr_dtype != LDPC_F64) ||
        (compute_dtype != LDPC_F32 && compute_dtype != LDPC_F64 && compute_dtype != LDPC_F16))
        return fail(LDPC_ERR_ARG, "llr_dtype must be LDPC_F32 or LDPC_F64, compute_dtype LDPC_F32, LDPC_F64 or LDPC_F16");
    if (kernel < LDPC_KERNEL_AUTO || kernel > LDPC_KERNEL_GRAPH) return fail(LDPC_ERR_ARG, "unknown kernel selector %d", kernel);
    if (frames == 0) { g_kernel = "none"; return LDPC_OK; }
    if (!llr_dev || !hard_dev || !iters_dev || !converged_dev) return fail(LDPC_ERR_ARG, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    if (compute_dtype == LDPC_F16) {  // reduced-precision fast mode: its own kernel, nothing to fall back to
        const bool plain = cfg->num_active_checks < 0 && cfg->num_refresh_vars < 0 && !(cfg->clip_post > 0.0) && kernel == LDPC_KERNEL_AUTO;
        bool supported = false;
        if (plain && c->has_qc && c->spec_id >= 0 && !posterior_dev) {
            Launch ioh{llr_dev, llr_dtype, (long long)frames, hard_dev, iters_dev, converged_dev, nullptr,
                       reinterpret_cast<hipStream_t>(stream)};
            hipError_t e = cfg->rule == LDPC_RULE_LNMS
                               ? launch_spec_lnms_half(c->spec_id, *cfg, ioh, c->num_cus, &g_kernel, &supported)
                               : launch_spec_half(c->spec_id, *cfg, ioh, c->num_cus, &g_kernel, &supported);
            if (e != hipSuccess) return fail(LDPC_ERR_HIP, "half-precision kernel launch: %s", hipGetErrorString(e));
        }
        if (!supported)
            return fail(LDPC_ERR_UNSUPPORTED, "the half-precision fast mode serves rule R1 on the small compiled-in codes and "
                                              "layered min-sum on the compiled-in codes without double-circulant blocks "
                                              "(float LLRs, no posteriors, no loop-bound quirks, automatic kernel choice)");
        return LDPC_OK;
    }

    Launch io{llr_dev, llr_dtype, (long long)frames, hard_dev, iters_dev, converged_dev, posterior_dev,
              reinterpret_cast<hipStream_t>(stream)};
    Scratch fixed_ws;
    if (noalloc)
        if (int rc = noalloc_scratch(c->device, &fixed_ws)) return rc;
    Scratch &ws = noalloc ? fixed_ws : c->workspace_for(io.stream);
    auto scratch_full = [&](hipError_t e) { return noalloc && e == hipErrorOutOfMemory && ws.exhausted; };
    const char *too_big = "one frame's state exceeds the %zu bytes of scratch of the allocation-free entry point; use ldpc_decode_batch";
    if (cfg->rule == LDPC_RULE_BF) {  // hard-decision bit flipping: its own kernel, any matrix
        if (posterior_dev) return fail(LDPC_ERR_UNSUPPORTED, "bit flipping has no posteriors: pass posterior_dev = NULL");
        bool supported = false;
        hipError_t e = launch_bitflip(c->csr, cfg->max_iters, cfg->num_active_checks, io, c->num_cus, &g_kernel, &supported);
        if (e != hipSuccess) return fail(LDPC_ERR_HIP, "bit-flipping kernel launch: %s", hipGetErrorString(e));
        if (!supported) return fail(LDPC_ERR_UNSUPPORTED, "bit flipping keeps 5 n + m bytes of a frame in LDS: the matrix is too large");
        return LDPC_OK;
    }
    // the loop-bound quirks exist in the edge-list kernel, in the table-driven circulant kernels and, as template
    // switches, in some compiled-in kernels (launch_spec_bounded)
    const bool quirks = (cfg->rule == 0 && cfg->num_active_checks >= 0 && cfg->num_active_checks < c->m) ||
                        (cfg->rule != 0 && cfg->num_refresh_vars >= 0 && cfg->num_refresh_vars < c->n);
    if (kernel == LDPC_KERNEL_QC && !c->has_qc) return fail(LDPC_ERR_ARG, "no QC description attached to this code");
    if (kernel == LDPC_KERNEL_GRAPH && (cfg->rule != LDPC_RULE_LNMS || c->bg.graph < 0))
        return fail(LDPC_ERR_ARG, "the base-graph kernel serves layered min-sum on handles whose circulant description is a "
                                  "row prefix of an NR base graph");
    if (cfg->rule == LDPC_RULE_LNMS && (quirks || cfg->clip_post > 0.0))
        return fail(LDPC_ERR_UNSUPPORTED, "layered min-sum has no num_refresh_vars / clip_post");
    // layered min-sum on the first num_active_checks rows: the decoder sees that sub-matrix only (what 5G
    // rate matching leaves of the matrix); served by the base-graph kernel for whole block rows
    int lnms_rows = -1;
    if (cfg->rule == LDPC_RULE_LNMS && cfg->num_active_checks >= 0 && cfg->num_active_checks < c->m) {
        if (c->bg.graph < 0 || cfg->num_active_checks % c->bg.Z != 0 || (kernel != LDPC_KERNEL_AUTO && kernel != LDPC_KERNEL_GRAPH))
            return fail(LDPC_ERR_UNSUPPORTED, "layered min-sum takes num_active_checks only as a whole number of block rows, on a "
                                              "handle whose circulant description is a row prefix of an NR base graph");
        lnms_rows = cfg->num_active_checks / c->bg.Z;
    }
    // the compiled-in R1 kernels carry no posterior clip (the reference's R1 has none, bpsk.py:235-237)
    const bool spec_ok = !((cfg->rule == LDPC_RULE_R1 || cfg->rule == LDPC_RULE_OMS) && cfg->clip_post > 0.0);
    if (kernel == LDPC_KERNEL_AUTO && c->has_qc && !quirks && c->spec_id >= 0 && spec_ok && lnms_rows < 0) {
        bool supported = false;
        hipError_t e = launch_spec(c->spec_id, *cfg, compute_dtype, io, c->num_cus, &g_kernel, &supported);
        if (e != hipSuccess) return fail(LDPC_ERR_HIP, "specialised kernel launch: %s", hipGetErrorString(e));
        if (supported) return LDPC_OK;
    }
    if (kernel == LDPC_KERNEL_AUTO && c->has_qc && quirks && c->spec_id >= 0 && spec_ok && cfg->rule != LDPC_RULE_LNMS) {
        bool supported = false;  // the loop bounds as template switches of some compiled-in kernels
        hipError_t e = launch_spec_bounded(c->spec_id, *cfg, compute_dtype, io, c->num_cus, &g_kernel, &supported);
        if (e != hipSuccess) return fail(LDPC_ERR_HIP, "specialised kernel launch: %s", hipGetErrorString(e));
        if (supported) return LDPC_OK;
    }
    if (cfg->rule == LDPC_RULE_LNMS) {
        if ((kernel == LDPC_KERNEL_AUTO || kernel == LDPC_KERNEL_GRAPH) && c->bg.graph >= 0) {
            bool supported = false;
            hipError_t e = launch_bg_lnms(c->bg, lnms_rows < 0 ? c->bg.mb : lnms_rows, *cfg, compute_dtype, io, c->num_cus,
                                          &g_kernel, &supported);
            if (e != hipSuccess) return fail(LDPC_ERR_HIP, "base-graph layered kernel launch: %s", hipGetErrorString(e));
            if (supported) return LDPC_OK;
            if (kernel == LDPC_KERNEL_GRAPH || lnms_rows >= 0)
                return fail(LDPC_ERR_UNSUPPORTED, "the base-graph layered kernel needs the core block rows active, a lifting size "
                                                  "of at most 768 (384 in double) and float LLRs for float arithmetic");
        }
        if (kernel != LDPC_KERNEL_CSR && c->has_qc && !c->qc_pairs) {
            bool supported = false;
            hipError_t e = launch_qc_lnms(c->qc, *cfg, compute_dtype, io, c->num_cus, ws, &g_kernel, &supported);
            if (scratch_full(e)) return fail(LDPC_ERR_UNSUPPORTED, too_big, kNoallocScratchBytes);
            if (e != hipSuccess) return fail(LDPC_ERR_HIP, "layered kernel launch: %s", hipGetErrorString(e));
            if (supported) return LDPC_OK;
        }
        return fail(LDPC_ERR_UNSUPPORTED, "layered min-sum needs a circulant description without double-circulant blocks, "
                                          "check degrees up to 24 and posteriors that fit LDS in the chosen arithmetic");
    }
    if (kernel != LDPC_KERNEL_CSR && c->has_qc) {
        bool supported = false;
        hipError_t e = hipSuccess;
        switch (cfg->rule) {
            case 0: e = launch_qc_rule_0(c->qc, *cfg, compute_dtype, io, c->num_cus, &g_kernel, &supported); break;
            case 1: e = launch_qc_rule_1(c->qc, *cfg, compute_dtype, io, c->num_cus, &g_kernel, &supported); break;
            case 2: e = launch_qc_rule_2(c->qc, *cfg, compute_dtype, io, c->num_cus, &g_kernel, &supported); break;
            case 3: e = launch_qc_rule_3(c->qc, *cfg, compute_dtype, io, c->num_cus, &g_kernel, &supported); break;
            case 4: e = launch_qc_rule_4(c->qc, *cfg, compute_dtype, io, c->num_cus, &g_kernel, &supported); break;
            case 6: e = launch_qc_rule_6(c->qc, *cfg, compute_dtype, io, c->num_cus, &g_kernel, &supported); break;
        }
        if (e != hipSuccess) return fail(LDPC_ERR_HIP, "circulant kernel launch: %s", hipGetErrorString(e));
        if (supported) return LDPC_OK;
        if (kernel == LDPC_KERNEL_QC)
            return fail(LDPC_ERR_UNSUPPORTED, "circulant kernel cannot serve this code/dtype (LDS footprint or degree)");
    }
    hipError_t e = launch_csr(c->csr, *cfg, compute_dtype, io, c->num_cus, ws, &g_kernel);
    if (scratch_full(e)) return fail(LDPC_ERR_UNSUPPORTED, too_big, kNoallocScratchBytes);
    if (e != hipSuccess) return fail(LDPC_ERR_HIP, "edge-list kernel launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

extern "C" int32_t ldpc_decode_batch(ldpc_code *c, const ldpc_decode_config *cfg, const void *llr_dev,
                                     int32_t llr_dtype, int32_t compute_dtype, int32_t kernel,
                                     int64_t frames, uint8_t *hard_dev, int32_t *iters_dev,
                                     uint8_t *converged_dev, void *posterior_dev, void *stream)
{
    return decode_impl(c, cfg, llr_dev, llr_dtype, compute_dtype, kernel, frames, hard_dev, iters_dev,
                       converged_dev, posterior_dev, stream, false);
}

extern "C" int32_t ldpc_decode_batch_noalloc(ldpc_code *c, const ldpc_decode_config *cfg, const void *llr_dev,
                                             int32_t llr_dtype, int32_t compute_dtype, int32_t kernel,
                                             int64_t frames, uint8_t *hard_dev, int32_t *iters_dev,
                                             uint8_t *converged_dev, void *posterior_dev, void *stream)
{
    return decode_impl(c, cfg, llr_dev, llr_dtype, compute_dtype, kernel, frames, hard_dev, iters_dev,
                       converged_dev, posterior_dev, stream, true);
}

extern "C" int64_t ldpc_noalloc_scratch_bytes(void) { return (int64_t)kNoallocScratchBytes; }

extern "C" int64_t ldpc_code_workspace_bytes(const ldpc_code *c)
{
    if (!c) return 0;
    size_t total = 0;
    for (const auto &w : c->workspaces) total += w.second.bytes;
    return (int64_t)total;
}

extern "C" int32_t ldpc_decode_batch_host(ldpc_code *c, const ldpc_decode_config *cfg, const void *llr,
                                          int32_t llr_dtype, int32_t compute_dtype, int32_t kernel,
                                          int64_t frames, uint8_t *hard, int32_t *iters,
                                          uint8_t *converged, void *posterior)
{
    if (!c) return fail(LDPC_ERR_ARG, "code is NULL");
    if (frames < 0) return fail(LDPC_ERR_ARG, "negative frame count");
    if (frames == 0) return ldpc_decode_batch(c, cfg, nullptr, llr_dtype, compute_dtype, kernel, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (!llr || !hard || !iters || !converged) return fail(LDPC_ERR_ARG, "NULL buffer");
    if ((llr_dtype != LDPC_F32 && llr_dtype != LDPC_F64) ||
        (compute_dtype != LDPC_F32 && compute_dtype != LDPC_F64 && compute_dtype != LDPC_F16))
        return fail(LDPC_ERR_ARG, "llr_dtype must be LDPC_F32 or LDPC_F64, compute_dtype LDPC_F32, LDPC_F64 or LDPC_F16");
    if (compute_dtype == LDPC_F16 && posterior) return fail(LDPC_ERR_UNSUPPORTED, "the half-precision fast mode hands back no posteriors");
    HIP_TRY(hipSetDevice(c->device));
    const size_t cells = (size_t)frames * c->n;
    const size_t llr_b = cells * (llr_dtype == LDPC_F64 ? 8 : 4);
    const size_t post_b = posterior ? cells * (compute_dtype == LDPC_F64 ? 8 : 4) : 0;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t total = up(llr_b) + up(cells) + up((size_t)frames * 4) + up((size_t)frames) + up(post_b);
    if (c->host_stage_bytes < total) {  // the staging area is kept between calls: a frame-at-a-time caller pays it once
        if (c->host_stage) (void)hipFree(c->host_stage);
        c->host_stage = nullptr;
        c->host_stage_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->host_stage), total));
        c->host_stage_bytes = total;
    }
    char *d = c->host_stage;
    char *d_llr = d, *d_hard = d_llr + up(llr_b), *d_it = d_hard + up(cells),
         *d_cv = d_it + up((size_t)frames * 4), *d_post = d_cv + up((size_t)frames);
    int rc = LDPC_OK;
    hipError_t e = hipMemcpy(d_llr, llr, llr_b, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = ldpc_decode_batch(c, cfg, d_llr, llr_dtype, compute_dtype, kernel, frames,
                               reinterpret_cast<uint8_t *>(d_hard), reinterpret_cast<int32_t *>(d_it),
                               reinterpret_cast<uint8_t *>(d_cv), posterior ? d_post : nullptr, nullptr);
        if (rc == LDPC_OK) {
            // hard | iters | converged lie next to each other in the staging area: a frame-at-a-time caller
            // (the reference-shaped classes) gets them back in one copy instead of three
            const size_t span = (size_t)(d_cv - d_hard) + (size_t)frames;
            if (span <= (size_t)1 << 20) {
                c->host_bounce.resize(span);
                e = hipMemcpy(c->host_bounce.data(), d_hard, span, hipMemcpyDeviceToHost);  // also waits for the kernel
                if (e == hipSuccess) {
                    memcpy(hard, c->host_bounce.data(), cells);
                    memcpy(iters, c->host_bounce.data() + (d_it - d_hard), (size_t)frames * 4);
                    memcpy(converged, c->host_bounce.data() + (d_cv - d_hard), (size_t)frames);
                }
            } else {
                e = hipMemcpy(hard, d_hard, cells, hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipMemcpy(iters, d_it, (size_t)frames * 4, hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipMemcpy(converged, d_cv, (size_t)frames, hipMemcpyDeviceToHost);
            }
            if (e == hipSuccess && posterior) e = hipMemcpy(posterior, d_post, post_b, hipMemcpyDeviceToHost);
        }
    }
    if (rc != LDPC_OK) return rc;
    if (e != hipSuccess) return fail(LDPC_ERR_HIP, "host decode: %s", hipGetErrorString(e));
    return LDPC_OK;
}

extern "C" int32_t ldpc_count_errors(const uint8_t *hard_dev, const uint8_t *ref_dev, int64_t ref_stride,
                                     const int32_t *iters_dev, const uint8_t *converged_dev,
                                     int64_t frames, int32_t n, uint64_t *counters_dev, void *stream)
{
    if (frames < 0 || n <= 0) return fail(LDPC_ERR_ARG, "bad frame count or n");
    if (frames == 0) return LDPC_OK;
    if (!hard_dev || !ref_dev || !iters_dev || !converged_dev || !counters_dev) return fail(LDPC_ERR_ARG, "NULL buffer");
    if (ref_stride != 0 && ref_stride < n) return fail(LDPC_ERR_ARG, "ref_stride must be 0 or at least n");
    hipError_t e = launch_count_errors(hard_dev, ref_dev, ref_stride, iters_dev, converged_dev, frames, n,
                                       reinterpret_cast<unsigned long long *>(counters_dev),
                                       reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(LDPC_ERR_HIP, "count_errors launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

static int channel_impl(const uint8_t *ref_dev, int64_t ref_stride, int64_t frames, int32_t n, double sigma2,
                        double gain, int32_t plus_is_bit, int32_t prefix_len, double prefix_llr, uint64_t seed,
                        uint64_t first_frame, double cfo, double phase0, void *llr_dev, int32_t llr_dtype, void *stream)
{
    if (frames < 0 || n <= 0) return fail(LDPC_ERR_ARG, "bad frame count or n");
    if (frames == 0) return LDPC_OK;
    if (!llr_dev) return fail(LDPC_ERR_ARG, "NULL buffer");
    if (llr_dtype != LDPC_F32 && llr_dtype != LDPC_F64) return fail(LDPC_ERR_ARG, "dtype must be LDPC_F32 or LDPC_F64");
    if (ref_dev && ref_stride != 0 && ref_stride != n) return fail(LDPC_ERR_ARG, "ref_stride must be 0 or n");
    if (!(sigma2 >= 0.0) || prefix_len < 0) return fail(LDPC_ERR_ARG, "negative noise variance or prefix length");
    if (!(cfo > -0.5 && cfo < 0.5) || !(phase0 > -128.0 && phase0 < 128.0))
        return fail(LDPC_ERR_ARG, "frequency offset must lie within half a cycle per symbol, phase within 128 cycles");
    hipError_t e = launch_channel_llr(ref_dev, ref_stride, frames, n, sigma2, gain, plus_is_bit != 0, prefix_len,
                                      prefix_llr, seed, first_frame, cfo, phase0, llr_dev, llr_dtype,
                                      reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(LDPC_ERR_HIP, "channel_llr launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

extern "C" int32_t ldpc_channel_llr(const uint8_t *ref_dev, int64_t ref_stride, int64_t frames, int32_t n,
                                    double sigma2, double gain, int32_t plus_is_bit, int32_t prefix_len,
                                    double prefix_llr, uint64_t seed, uint64_t first_frame, void *llr_dev,
                                    int32_t llr_dtype, void *stream)
{
    return channel_impl(ref_dev, ref_stride, frames, n, sigma2, gain, plus_is_bit, prefix_len, prefix_llr, seed,
                        first_frame, 0.0, 0.0, llr_dev, llr_dtype, stream);
}

extern "C" int32_t ldpc_channel_llr_doppler(const uint8_t *ref_dev, int64_t ref_stride, int64_t frames, int32_t n,
                                            double sigma2, double gain, int32_t plus_is_bit, int32_t prefix_len,
                                            double prefix_llr, uint64_t seed, uint64_t first_frame,
                                            double cfo_cycles_per_symbol, double phase0_cycles, void *llr_dev,
                                            int32_t llr_dtype, void *stream)
{
    return channel_impl(ref_dev, ref_stride, frames, n, sigma2, gain, plus_is_bit, prefix_len, prefix_llr, seed,
                        first_frame, cfo_cycles_per_symbol, phase0_cycles, llr_dev, llr_dtype, stream);
}

// ------------------------------------------------------------------------------------------ collective
// RCCL is looked up at the first call, not linked: a process that already holds an RCCL (PyTorch
// bundles one) must run the collective in THAT copy, the one its communicator belongs to.
typedef int (*nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
static nccl_allreduce_fn find_allreduce(std::string *where)
{
    static std::mutex mu;
    static nccl_allreduce_fn fn = nullptr;
    static std::string found;
    std::lock_guard<std::mutex> lock(mu);
    if (!fn) {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *h = nullptr;
        for (const char *nm : names)  // a copy that is already in the process first
            if ((h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) { found = std::string(nm) + " (already loaded)"; break; }
        if (!h)
            for (const char *nm : names)
                if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) { found = nm; break; }
        if (h) fn = reinterpret_cast<nccl_allreduce_fn>(dlsym(h, "ncclAllReduce"));
    }
    if (where) *where = found;
    return fn;
}

extern "C" int32_t ldpc_allreduce_counters(void *rccl_comm, uint64_t *counters_dev, int32_t count, void *stream)
{
    if (!counters_dev || count < 0) return fail(LDPC_ERR_ARG, "NULL buffer or negative count");
    if (!rccl_comm || count == 0) return LDPC_OK;  // one rank without a communicator: nothing to add
    std::string where;
    nccl_allreduce_fn fn = find_allreduce(&where);
    if (!fn) {
        const char *why = dlerror();
        return fail(LDPC_ERR_UNSUPPORTED, "no RCCL library found (librccl.so.1): %s", why ? why : "not on the library path");
    }
    const int rc = fn(counters_dev, counters_dev, (size_t)count, /*ncclUint64*/ 5, /*ncclSum*/ 0, rccl_comm,
                      reinterpret_cast<hipStream_t>(stream));
    if (rc != 0) return fail(LDPC_ERR_HIP, "ncclAllReduce in %s failed with ncclResult_t %d", where.c_str(), rc);
    return LDPC_OK;
}

// ------------------------------------------------------------------------------------------ encoder
struct ldpc_encoder {
    int device = 0, num_cus = 0;
    EncDev dev{};
    int *d_ints = nullptr;  // one allocation: info_cols | step_out | step_ptr | step_level | terms
    unsigned long long *d_taps = nullptr;  // dense schedules with Z <= 64: [n_steps][n_cols] tap masks, else NULL
};

extern "C" int32_t ldpc_encoder_create(int32_t Z, int32_t n_cols, int32_t nb, int32_t k_cols, const int32_t *info_cols,
                                       int32_t n_steps, const int32_t *step_out, const int32_t *step_ptr,
                                       const int32_t *step_level, const int32_t *term_col, const int32_t *term_shift,
                                       int32_t device, ldpc_encoder **out)
{
    if (!out) return fail(LDPC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (Z < 1 || nb < 1 || n_cols < nb || k_cols < 1 || k_cols > nb || n_steps < 0 || n_cols > 65535)
        return fail(LDPC_ERR_ARG, "bad encoder dimensions");
    if (!info_cols || !step_ptr || (n_steps > 0 && (!step_out || !step_level))) return fail(LDPC_ERR_ARG, "NULL table");
    if (Z > 1024) return fail(LDPC_ERR_UNSUPPORTED, "lifting size %d above 1024", Z);
    // every step must read columns that are known by then: information columns or outputs of earlier levels
    std::vector<int> known_at(n_cols, -2);  // level at which a column becomes known; -1 = information
    for (int j = 0; j < k_cols; ++j) {
        if (info_cols[j] < 0 || info_cols[j] >= nb || known_at[info_cols[j]] != -2) return fail(LDPC_ERR_ARG, "bad information column list");
        known_at[info_cols[j]] = -1;
    }
    if (step_ptr[0] != 0) return fail(LDPC_ERR_ARG, "step_ptr must start at 0");
    for (int s = 0; s < n_steps; ++s) {
        if (step_ptr[s + 1] < step_ptr[s]) return fail(LDPC_ERR_ARG, "step_ptr must not decrease");
        if (step_level[s] < 0 || (s > 0 && step_level[s] < step_level[s - 1])) return fail(LDPC_ERR_ARG, "step levels must be non-negative and non-decreasing");
        if (step_out[s] < 0 || step_out[s] >= n_cols || known_at[step_out[s]] != -2) return fail(LDPC_ERR_ARG, "step %d writes a column that is already known", s);
        for (int t = step_ptr[s]; t < step_ptr[s + 1]; ++t) {
            if (!term_col || !term_shift) return fail(LDPC_ERR_ARG, "NULL table");
            if (term_col[t] < 0 || term_col[t] >= n_cols || term_shift[t] < 0 || term_shift[t] >= Z) return fail(LDPC_ERR_ARG, "term %d out of range", t);
            const int ka = known_at[term_col[t]];
            if (ka == -2 || ka >= step_level[s]) return fail(LDPC_ERR_ARG, "step %d reads column %d before it is known", s, term_col[t]);
        }
        known_at[step_out[s]] = step_level[s];
    }
    for (int c = 0; c < nb; ++c)
        if (known_at[c] == -2) return fail(LDPC_ERR_ARG, "codeword column %d is never produced", c);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(LDPC_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0) HIP_TRY(hipGetDevice(&device));  // the calling thread's current device, as ldpc_code_create_csr takes it
    if (device < 0 || device >= ndev) return fail(LDPC_ERR_ARG, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<ldpc_encoder> enc(new ldpc_encoder);
    enc->device = device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    enc->num_cus = prop.multiProcessorCount;
    const int n_terms = step_ptr[n_steps];
    std::vector<int> host;
    host.insert(host.end(), info_cols, info_cols + k_cols);
    host.insert(host.end(), step_out, step_out + n_steps);
    host.insert(host.end(), step_ptr, step_ptr + n_steps + 1);
    host.insert(host.end(), step_level, step_level + n_steps);
    for (int t = 0; t < n_terms; ++t) host.push_back((int)(((unsigned)term_col[t] << 16) | (unsigned)term_shift[t]));
    // the word-wide kernel's copy of the schedule (see EncDev)
    const bool words4 = Z % 4 == 0 && Z >= 8 && n_cols < 65535;
    size_t ptr4_at = 0, terms4_at = 0, islot_at = 0, sslot_at = 0;
    int n_slots = 0;
    if (words4) {
        std::vector<int> slot(n_cols, -1);  // LDS places go to the columns that are read
        for (int t = 0; t < n_terms; ++t)
            if (slot[term_col[t]] < 0) slot[term_col[t]] = n_slots++;
        islot_at = host.size();
        for (int j = 0; j < k_cols; ++j) host.push_back(slot[info_cols[j]]);
        sslot_at = host.size();
        for (int s = 0; s < n_steps; ++s) host.push_back(slot[step_out[s]]);
        ptr4_at = host.size();
        // terms in groups of four, decoded for the kernel: per group twelve words - the four word-aligned shifts,
        // the four byte offsets of the source slots (a slot is Z + 4 bytes), the four byte shifts
        std::vector<std::array<int, 3>> padded;
        for (int s = 0; s < n_steps; ++s) {
            host.push_back((int)padded.size());
            for (int t = step_ptr[s]; t < step_ptr[s + 1]; ++t)
                padded.push_back({term_shift[t] & ~3, slot[term_col[t]] * (Z + 4), term_shift[t] & 3});
            while (padded.size() % 4) padded.push_back({0, n_slots * (Z + 4), 0});
        }
        host.push_back((int)padded.size());
        while (host.size() % 4) host.push_back(0);
        terms4_at = host.size();
        for (size_t g = 0; g < padded.size(); g += 4)
            for (int w = 0; w < 3; ++w)
                for (int u = 0; u < 4; ++u) host.push_back(padded[g + u][w]);
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&enc->d_ints), host.size() * sizeof(int) + 16));
    hipError_t e = hipMemcpy(enc->d_ints, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(enc->d_ints); return fail(LDPC_ERR_HIP, "encoder tables: %s", hipGetErrorString(e)); }
    EncDev &d = enc->dev;
    d.Z = Z; d.n_cols = n_cols; d.nb = nb; d.k_cols = k_cols; d.n_steps = n_steps;
    d.info_cols = enc->d_ints;
    d.step_out = d.info_cols + k_cols;
    d.step_ptr = d.step_out + n_steps;
    d.step_level = d.step_ptr + n_steps + 1;
    d.terms = reinterpret_cast<const unsigned *>(d.step_level + n_steps);
    d.n_slots4 = words4 ? n_slots : 0;
    d.info_slot4 = words4 ? enc->d_ints + islot_at : nullptr;
    d.step_slot4 = words4 ? enc->d_ints + sslot_at : nullptr;
    d.step_ptr4 = words4 ? enc->d_ints + ptr4_at : nullptr;
    d.terms4 = words4 ? reinterpret_cast<const unsigned *>(enc->d_ints + terms4_at) : nullptr;
    // a schedule with many taps per (output, input) column pair and Z <= 64 runs on ballot masks
    // (its kernel holds 256 column masks and the random words of 4096 message bits)
    if (Z <= 64 && n_cols <= 256 && (long long)k_cols * Z <= 4096 && n_steps > 0) {
        std::vector<unsigned long long> taps((size_t)n_steps * n_cols, 0ull);
        long long pairs = 0;
        for (int s = 0; s < n_steps; ++s)
            for (int t = step_ptr[s]; t < step_ptr[s + 1]; ++t) {
                unsigned long long &m = taps[(size_t)s * n_cols + term_col[t]];
                pairs += m == 0ull;
                m ^= 1ull << term_shift[t];  // a shift listed twice cancels, as it does in the XOR of the terms
            }
        if (pairs > 0 && n_terms >= 4 * pairs) {
            e = hipMalloc(reinterpret_cast<void **>(&enc->d_taps), taps.size() * sizeof(unsigned long long));
            if (e != hipSuccess) {
                (void)hipFree(enc->d_ints);
                return fail(LDPC_ERR_HIP, "encoder tap masks: %s", hipGetErrorString(e));
            }
            e = hipMemcpy(enc->d_taps, taps.data(), taps.size() * sizeof(unsigned long long), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                (void)hipFree(enc->d_taps);
                (void)hipFree(enc->d_ints);
                return fail(LDPC_ERR_HIP, "encoder tables: %s", hipGetErrorString(e));
            }
        }
    }
    *out = enc.release();
    return LDPC_OK;
}

extern "C" void ldpc_encoder_destroy(ldpc_encoder *enc)
{
    if (!enc) return;
    (void)hipSetDevice(enc->device);
    if (enc->d_ints) (void)hipFree(enc->d_ints);
    if (enc->d_taps) (void)hipFree(enc->d_taps);
    delete enc;
}

extern "C" int32_t ldpc_encode_batch(ldpc_encoder *enc, const uint8_t *msg_dev, int64_t frames, uint8_t *cw_dev, void *stream)
{
    if (!enc) return fail(LDPC_ERR_ARG, "encoder is NULL");
    if (frames < 0) return fail(LDPC_ERR_ARG, "negative frame count");
    if (frames == 0) return LDPC_OK;
    if (!msg_dev || !cw_dev) return fail(LDPC_ERR_ARG, "NULL buffer");
    HIP_TRY(hipSetDevice(enc->device));
    bool supported = false;
    hipError_t e = launch_encode_qc(enc->dev, msg_dev, frames, cw_dev, enc->num_cus, reinterpret_cast<hipStream_t>(stream), &supported,
                                    0, 0, nullptr, enc->d_taps);
    if (e != hipSuccess) return fail(LDPC_ERR_HIP, "encoder launch: %s", hipGetErrorString(e));
    if (!supported) return fail(LDPC_ERR_UNSUPPORTED, "one frame's block columns exceed LDS");
    return LDPC_OK;
}

extern "C" int32_t ldpc_encode_random(ldpc_encoder *enc, uint64_t seed, uint64_t first_frame, int64_t frames,
                                      uint8_t *cw_dev, uint8_t *msg_out_dev, void *stream)
{
    if (!enc) return fail(LDPC_ERR_ARG, "encoder is NULL");
    if (frames < 0) return fail(LDPC_ERR_ARG, "negative frame count");
    if (frames == 0) return LDPC_OK;
    if (!cw_dev) return fail(LDPC_ERR_ARG, "NULL buffer");
    HIP_TRY(hipSetDevice(enc->device));
    bool supported = false;
    hipError_t e = launch_encode_qc(enc->dev, nullptr, frames, cw_dev, enc->num_cus, reinterpret_cast<hipStream_t>(stream), &supported,
                                    seed, first_frame, msg_out_dev, enc->d_taps);
    if (e != hipSuccess) return fail(LDPC_ERR_HIP, "encoder launch: %s", hipGetErrorString(e));
    if (!supported) return fail(LDPC_ERR_UNSUPPORTED, "one frame's block columns exceed LDS");
    return LDPC_OK;
}

// ------------------------------------------------------------------------------------------ rate matching
static int check_rm(const ldpc_rate_match_config *c)
{
    if (!c) return fail(LDPC_ERR_ARG, "rate matching configuration is NULL");
    if (c->n_cb < 1 || c->E < 1 || c->offset < 0 || c->n_out < 1) return fail(LDPC_ERR_ARG, "bad rate matching sizes");
    if (c->k0 < 0 || c->k0 >= c->n_cb) return fail(LDPC_ERR_ARG, "k0 outside the circular buffer");
    if (c->filler_begin < 0 || c->filler_end < c->filler_begin) return fail(LDPC_ERR_ARG, "bad filler range");
    const int fa = c->filler_begin < c->n_cb ? c->filler_begin : c->n_cb, fb = c->filler_end < c->n_cb ? c->filler_end : c->n_cb;
    if (c->n_cb - (fb - fa) < 1) return fail(LDPC_ERR_ARG, "the circular buffer holds fillers only");
    if (c->Qm < 1 || c->E % c->Qm != 0) return fail(LDPC_ERR_ARG, "E must be a multiple of Qm");
    return LDPC_OK;
}

extern "C" int32_t ldpc_rate_match(const ldpc_rate_match_config *cfg, const uint8_t *cw_dev, int64_t cw_stride,
                                   int64_t frames, uint8_t *out_dev, void *stream)
{
    if (int rc = check_rm(cfg)) return rc;
    if (frames < 0) return fail(LDPC_ERR_ARG, "negative frame count");
    if (frames == 0) return LDPC_OK;
    if (!cw_dev || !out_dev) return fail(LDPC_ERR_ARG, "NULL buffer");
    if (cw_stride < (int64_t)cfg->offset + cfg->n_cb) return fail(LDPC_ERR_ARG, "cw_stride shorter than offset + n_cb");
    hipError_t e = launch_rate_match(*cfg, cw_dev, cw_stride, frames, out_dev, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(LDPC_ERR_HIP, "rate_match launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

extern "C" int32_t ldpc_rate_recover(const ldpc_rate_match_config *cfg, const void *llr_dev, int32_t llr_dtype,
                                     int64_t frames, double filler_llr, int32_t accumulate, void *out_dev,
                                     int32_t out_dtype, void *stream)
{
    if (int rc = check_rm(cfg)) return rc;
    if (frames < 0) return fail(LDPC_ERR_ARG, "negative frame count");
    if ((llr_dtype != LDPC_F32 && llr_dtype != LDPC_F64) || (out_dtype != LDPC_F32 && out_dtype != LDPC_F64))
        return fail(LDPC_ERR_ARG, "dtype must be LDPC_F32 or LDPC_F64");
    if (frames == 0) return LDPC_OK;
    if (!llr_dev || !out_dev) return fail(LDPC_ERR_ARG, "NULL buffer");
    hipError_t e = launch_rate_recover(*cfg, llr_dev, llr_dtype, frames, filler_llr, accumulate != 0, out_dev, out_dtype,
                                       reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(LDPC_ERR_HIP, "rate_recover launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

static int sweep_impl(ldpc_code *c, const ldpc_decode_config *cfg, const ldpc_channel_config *ch,
                      const uint8_t *ref_dev, int64_t ref_stride, bool packed, int64_t frames, int32_t compute_dtype,
                      uint64_t *counters_dev, void *stream)
{
    if (!c) return fail(LDPC_ERR_ARG, "code is NULL");
    if (int rc = check_cfg(cfg)) return rc;
    if (!ch) return fail(LDPC_ERR_ARG, "channel is NULL");
    if (frames < 0) return fail(LDPC_ERR_ARG, "negative frame count");
    if (compute_dtype != LDPC_F32 && compute_dtype != LDPC_F64 && compute_dtype != LDPC_F16)
        return fail(LDPC_ERR_ARG, "dtype must be LDPC_F32, LDPC_F64 or LDPC_F16");
    if (!counters_dev) return fail(LDPC_ERR_ARG, "NULL buffer");
    if (packed) {
        if (ref_dev && ref_stride != 0 && ref_stride < (c->n + 7) / 8)
            return fail(LDPC_ERR_ARG, "ref_stride must be 0 or at least (n + 7) / 8 bytes");
    } else {
        if (ref_dev && ref_stride != 0 && ref_stride != c->n) return fail(LDPC_ERR_ARG, "ref_stride must be 0 or n");
        if (ref_dev && (reinterpret_cast<uintptr_t>(ref_dev) & 3u)) return fail(LDPC_ERR_ARG, "ref_dev must be 4-byte aligned");
    }
    if (!(ch->sigma2 >= 0.0) || ch->prefix_len < 0) return fail(LDPC_ERR_ARG, "negative noise variance or prefix length");
    if (!(ch->cfo_cycles_per_symbol > -0.5 && ch->cfo_cycles_per_symbol < 0.5) || !(ch->phase0_cycles > -128.0 && ch->phase0_cycles < 128.0))
        return fail(LDPC_ERR_ARG, "frequency offset must lie within half a cycle per symbol, phase within 128 cycles");
    if (frames == 0) { g_kernel = "none"; return LDPC_OK; }
    HIP_TRY(hipSetDevice(c->device));
    const bool quirks = (cfg->num_active_checks >= 0 && cfg->num_active_checks < c->m) ||
                        (cfg->num_refresh_vars >= 0 && cfg->num_refresh_vars < c->n);
    const bool spa = cfg->rule >= 2 && cfg->rule <= 4;  // the sum-product kernels apply the posterior clip themselves
    if (c->spec_id < 0 || !c->has_qc || quirks || (cfg->clip_post > 0.0 && !spa))
        return fail(LDPC_ERR_UNSUPPORTED, "the fused sweep serves compiled-in codes without loop-bound quirks (and, for the min-sum rules, "
                                          "without posterior clip); use ldpc_channel_llr + ldpc_decode_batch + ldpc_count_errors");
    SweepDev sp;
    sp.ch = make_channel_dev(ch->sigma2, ch->gain, ch->plus_is_bit != 0, ch->prefix_len, ch->prefix_llr, ch->seed,
                             ch->first_frame, ch->cfo_cycles_per_symbol, ch->phase0_cycles);
    sp.ref = ref_dev;
    sp.ref_stride = ref_dev ? ref_stride : 0;
    sp.ref_packed = packed ? 1 : 0;
    sp.counters = reinterpret_cast<unsigned long long *>(counters_dev);
    Launch io{nullptr, LDPC_F32, (long long)frames, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<hipStream_t>(stream)};
    bool supported = false;
    hipError_t e = hipSuccess;
    if (spa) {  // sum-product: the single-wave kernel where it serves the code, else the multi-wave one
        if (compute_dtype != LDPC_F16) {
            e = launch_spec_spa_fused(c->spec_id, *cfg, compute_dtype, io, sp, c->num_cus, &g_kernel, &supported);
            if (e == hipSuccess && !supported) {
                Scratch &ws = c->workspace_for(io.stream);
                e = cfg->rule == 2   ? launch_spec_spaw_fused_rule_2(c->spec_id, *cfg, compute_dtype, io, sp, c->num_cus, ws, &g_kernel, &supported)
                    : cfg->rule == 3 ? launch_spec_spaw_fused_rule_3(c->spec_id, *cfg, compute_dtype, io, sp, c->num_cus, ws, &g_kernel, &supported)
                                     : launch_spec_spaw_fused_rule_4(c->spec_id, *cfg, compute_dtype, io, sp, c->num_cus, ws, &g_kernel, &supported);
            }
        }
    } else if (compute_dtype == LDPC_F16) {
        e = cfg->rule == LDPC_RULE_LNMS
                ? launch_spec_lnms_half_fused(c->spec_id, *cfg, io, sp, c->num_cus, c->workspace_for(io.stream), &g_kernel, &supported)
                : launch_spec_half_fused(c->spec_id, *cfg, io, sp, c->num_cus, &g_kernel, &supported);
    } else {
        e = launch_spec_fused(c->spec_id, *cfg, compute_dtype, io, sp, c->num_cus, c->workspace_for(io.stream), &g_kernel, &supported);
        if (e == hipSuccess && !supported) {  // the multi-wave flooding kernels
            if (cfg->rule == 0) e = launch_spec_r0w_fused(c->spec_id, *cfg, compute_dtype, io, sp, c->num_cus, c->workspace_for(io.stream), &g_kernel, &supported);
            else if (cfg->rule == 1) e = launch_spec_r1w_fused(c->spec_id, *cfg, compute_dtype, io, sp, c->num_cus, c->workspace_for(io.stream), &g_kernel, &supported);
            else if (cfg->rule == LDPC_RULE_OMS) e = launch_spec_omsw_fused(c->spec_id, *cfg, compute_dtype, io, sp, c->num_cus, c->workspace_for(io.stream), &g_kernel, &supported);
        }
    }
    if (e != hipSuccess) return fail(LDPC_ERR_HIP, "fused sweep launch: %s", hipGetErrorString(e));
    if (!supported)
        return fail(LDPC_ERR_UNSUPPORTED, "no fused sweep kernel for this code, rule and arithmetic; "
                                          "use ldpc_channel_llr + ldpc_decode_batch + ldpc_count_errors");
    return LDPC_OK;
}

extern "C" int32_t ldpc_sweep_batch(ldpc_code *c, const ldpc_decode_config *cfg, const ldpc_channel_config *ch,
                                    const uint8_t *ref_dev, int64_t ref_stride, int64_t frames, int32_t compute_dtype,
                                    uint64_t *counters_dev, void *stream)
{
    return sweep_impl(c, cfg, ch, ref_dev, ref_stride, false, frames, compute_dtype, counters_dev, stream);
}

extern "C" int32_t ldpc_sweep_batch_packed(ldpc_code *c, const ldpc_decode_config *cfg, const ldpc_channel_config *ch,
                                           const uint8_t *ref_bits_dev, int64_t ref_stride_bytes, int64_t frames,
                                           int32_t compute_dtype, uint64_t *counters_dev, void *stream)
{
    return sweep_impl(c, cfg, ch, ref_bits_dev, ref_stride_bytes, true, frames, compute_dtype, counters_dev, stream);
}

extern "C" int32_t ldpc_pack_bits(const uint8_t *bytes_dev, int64_t stride, int64_t frames, int32_t n, uint8_t *bits_dev,
                                  int64_t bits_stride, void *stream)
{
    if (frames < 0 || n < 0) return fail(LDPC_ERR_ARG, "negative size");
    if (frames == 0 || n == 0) return LDPC_OK;
    if (!bytes_dev || !bits_dev) return fail(LDPC_ERR_ARG, "NULL buffer");
    if (stride < n || bits_stride < (n + 7) / 8) return fail(LDPC_ERR_ARG, "stride shorter than a word");
    hipError_t e = launch_pack_bits(bytes_dev, stride, frames, n, bits_dev, bits_stride, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(LDPC_ERR_HIP, "pack_bits launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

extern "C" int32_t ldpc_decode_segments(const ldpc_segment *segs, int32_t n_segs, const ldpc_decode_config *cfg,
                                        int32_t llr_dtype, int32_t compute_dtype, int32_t kernel)
{
    if (n_segs < 0 || (n_segs > 0 && !segs)) return fail(LDPC_ERR_ARG, "bad segment list");
    for (int i = 0; i < n_segs; ++i) {
        const ldpc_segment &s = segs[i];
        if (int rc = decode_impl(s.code, cfg, s.llr_dev, llr_dtype, compute_dtype, kernel, s.frames, s.hard_dev,
                                 s.iters_dev, s.converged_dev, s.posterior_dev, s.stream, false)) {
            g_err = "segment " + std::to_string(i) + ": " + g_err;
            return rc;
        }
    }
    return LDPC_OK;
}
