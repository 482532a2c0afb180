// Shared host/device declarations of the decoder library (gfx950 only).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <tuple>
#include <utility>

#include "ldpc_amd.h"

namespace ldpc {

constexpr int kMaxCheckDegree = 64;  // sign bits of one check are kept in one 64-bit mask
// bytes of the file-scope device array behind ldpc_decode_batch_noalloc (held per GPU until the process ends)
#ifndef LDPC_NOALLOC_SCRATCH_BYTES
#define LDPC_NOALLOC_SCRATCH_BYTES (32u << 20)
#endif

// Edge lists on the device (all int32).
struct CsrDev {
    int m, n, E;
    const int *row_ptr;   // [m+1]
    const int *col_idx;   // [E] ascending column inside a row
    const int *col_ptr;   // [n+1]
    const int *csc_edge;  // [E] per variable, ascending check: position of the edge in CSR order
};

// Quasi-cyclic description on the device.  Base edges are stored twice: by block row (the order
// check nodes visit them) and by block column (the order variable nodes sum them: ascending block
// row).  The messages of base edge e (by-row position) form one vector indexed by the CHECK's
// index z inside its block row.
//   row_info[e] = block column << 16 | shift
//   col_info[k] = by-row position of the edge << 16 | pair bit (0x8000) | shift
//   colr_info[k] = block row of the edge << 16 | pair bit (0x8000) | shift
// The pair bit says that entry k+1 lies in the same block (a block that is the sum of two
// circulants); which of the two checks has the smaller index then depends on the lane.
struct QcDev {
    int mb, nb, Z, Eb;
    const int *row_ptr;        // [mb+1]
    const unsigned *row_info;  // [Eb]
    const int *col_ptr;        // [nb+1]
    const unsigned *col_info;  // [Eb]
    const unsigned *colr_info; // [Eb]
    int max_dc, max_dv;
};

// Rule constants converted to the arithmetic type.
template <typename T>
struct Cfg {
    int rule, max_iters, early_exit, float_syndrome0, num_active_checks, num_refresh_vars;
    T scale, clip_tanh_arg, clip_prod, clip_post, offset;
};

struct Launch {
    const void *llr;
    int llr_dtype;
    long long frames;
    uint8_t *hard;
    int *iters;
    uint8_t *converged;
    void *posterior;
    hipStream_t stream;
};

// Device scratch memory of the kernels whose per-frame state does not fit on chip (csr_global_*,
// qc_lnms_global_*): every resident workgroup owns one slice of it.  `fixed` scratch cannot grow:
// the launcher then starts only as many workgroups as it holds slices for, and reports
// hipErrorOutOfMemory (nothing launched) when not even one slice fits.
struct Scratch {
    void *ptr = nullptr;
    size_t bytes = 0;
    bool fixed = false;
    bool exhausted = false;  // set by scratch_reserve: fixed scratch too small for one slice
};
// Make room for `grid` slices of `per_wg` bytes; may lower `grid` (fixed scratch).  The buffer is sized
// for the batch at hand, not for the largest grid the device could hold (a 40-frame call on a code
// whose slices are megabytes must not hold gigabytes); it grows geometrically, at most to
// `want_grid` slices, and never shrinks.  Growing frees and allocates, which synchronises the device:
// the old buffer is idle by then.
inline hipError_t scratch_reserve(Scratch &ws, size_t per_wg, long long want_grid, long long &grid)
{
    ws.exhausted = false;
    if (ws.fixed) {
        const long long fit = (long long)(ws.bytes / per_wg);
        if (fit < 1) { ws.exhausted = true; return hipErrorOutOfMemory; }
        if (grid > fit) grid = fit;
        return hipSuccess;
    }
    const size_t need = (size_t)grid * per_wg, cap = (size_t)(want_grid > grid ? want_grid : grid) * per_wg;
    if (ws.bytes < need) {
        size_t take = need > 2 * ws.bytes ? need : 2 * ws.bytes;
        if (take > cap) take = cap;
        if (ws.ptr) (void)hipFree(ws.ptr);
        ws.ptr = nullptr; ws.bytes = 0;
        hipError_t e = hipMalloc(&ws.ptr, take);
        if (e != hipSuccess) return e;
        ws.bytes = take;
    }
    return hipSuccess;
}

// One-time set-up of a kernel on the current device: raise its dynamic LDS limit to `lds` bytes
// and (per_cu != NULL) ask the runtime how many workgroups of `threads` threads a compute unit
// holds.  Both answers are fixed for a (kernel, device, threads, lds) combination, so they are
// cached: every launch after the first costs one map lookup instead of two runtime calls.
inline hipError_t kernel_setup(const void *kernel, int threads, size_t lds, int *per_cu)
{
    static std::mutex mu;
    static std::map<std::tuple<const void *, int, int, size_t>, int> cache;
    static std::map<std::pair<const void *, int>, size_t> lds_limit;  // largest dynamic LDS size set per (kernel, device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const auto key = std::make_tuple(kernel, dev, threads, lds);
    std::lock_guard<std::mutex> lock(mu);
    // the attribute is one value per kernel, not per launch shape: it is only ever raised, so that a
    // kernel used with a large and then a small footprint keeps the limit the large one needs
    size_t &limit = lds_limit[std::make_pair(kernel, dev)];
    if (lds > limit) {
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        limit = lds;
    }
    auto it = cache.find(key);
    if (it == cache.end()) {
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, lds);
        if (e != hipSuccess) return e;
        if (getenv("LDPC_DEBUG_OCCUPANCY"))
            fprintf(stderr, "kernel %p: %d threads, %zu B dynamic LDS: %d workgroups per CU by the runtime's count\n",
                    kernel, threads, lds, nb);
        it = cache.emplace(key, nb).first;
    }
    if (per_cu) *per_cu = it->second;
    return hipSuccess;
}

// decode_csr.hip
hipError_t launch_csr(const CsrDev &code, const ldpc_decode_config &cfg, int compute_dtype,
                      const Launch &io, int num_cus, Scratch &workspace, const char **kernel_name);
// decode_qc_r0.hip (rule 0) and decode_qc.hip, one translation unit per rule (-DLDPC_QC_RULE=k, k = 1..4)
#define LDPC_DECL_QC(k)                                                                            \
    hipError_t launch_qc_rule_##k(const QcDev &code, const ldpc_decode_config &cfg,                \
                                  int compute_dtype, const Launch &io, int num_cus,                \
                                  const char **kernel_name, bool *supported);
LDPC_DECL_QC(0) LDPC_DECL_QC(1) LDPC_DECL_QC(2) LDPC_DECL_QC(3) LDPC_DECL_QC(4) LDPC_DECL_QC(6)
#undef LDPC_DECL_QC
// decode_spec.hip: kernels specialised at compile time for the packaged codes
int spec_match(int Z, int mb, int nb, int Eb, const int *er, const int *ec, const int *es);
hipError_t launch_spec(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                       int num_cus, const char **kernel_name, bool *supported);
hipError_t launch_spec_bounded(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                               int num_cus, const char **kernel_name, bool *supported);
// decode_spec_h.hip: rule R1 in packed half precision, two frames per lane (compute_dtype LDPC_F16)
hipError_t launch_spec_half(int spec_id, const ldpc_decode_config &cfg, const Launch &io, int num_cus,
                            const char **kernel_name, bool *supported);
// decode_lnms.hip: row-layered min-sum, specialised for the packaged codes (same spec_id)
hipError_t launch_spec_lnms(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                            int num_cus, const char **kernel_name, bool *supported);
// decode_spec_spa.hip: sum-product rules R2-R4, specialised for the small packaged codes
hipError_t launch_spec_spa(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                           int num_cus, const char **kernel_name, bool *supported);
// decode_spec_r1w.hip: rules R0 and R1 with several waves per frame group, for the larger packaged
// codes; one translation unit per rule
hipError_t launch_spec_r0w(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                           int num_cus, const char **kernel_name, bool *supported);
hipError_t launch_spec_r1w(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                           int num_cus, const char **kernel_name, bool *supported);
hipError_t launch_spec_omsw(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                            int num_cus, const char **kernel_name, bool *supported);
// decode_spec_spaw.hip: rules R2-R4 with several waves per frame group, one translation unit per rule
#define LDPC_DECL_SPAW(k)                                                                          \
    hipError_t launch_spec_spaw_rule_##k(int spec_id, const ldpc_decode_config &cfg, int compute_dtype, \
                                         const Launch &io, int num_cus, const char **kernel_name, bool *supported);
LDPC_DECL_SPAW(2) LDPC_DECL_SPAW(3) LDPC_DECL_SPAW(4)
#undef LDPC_DECL_SPAW
// decode_qc_lnms.hip: the same rule for any circulant code, base graph in device tables
hipError_t launch_qc_lnms(const QcDev &code, const ldpc_decode_config &cfg, int compute_dtype, const Launch &io,
                          int num_cus, Scratch &workspace, const char **kernel_name, bool *supported);
// decode_lnms_rt.hip: the same rule compiled per NR base graph (connectivity only); lifting size, shift
// table and the number of active block rows are run-time arguments.  A handle whose circulant
// description is a prefix of the block rows of such a graph carries the kernel's tables.
struct BgHandle {
    int graph = -1;  // which base graph; -1: none
    int Z = 0, mb = 0, nb = 0;
    unsigned *d_tab = nullptr;
};
int bg_graph_match(int Z, int mb, int nb, int Eb, const int *er, const int *ec);
hipError_t bg_tables_create(int graph, int Z, int mb, const int *es, unsigned **d_tab);
hipError_t launch_bg_lnms(const BgHandle &h, int mb_act, const ldpc_decode_config &cfg, int compute_dtype,
                          const Launch &io, int num_cus, const char **kernel_name, bool *supported);
// count_errors.hip
hipError_t launch_count_errors(const uint8_t *hard, const uint8_t *ref, long long ref_stride,
                               const int *iters, const uint8_t *conv, long long frames, int n,
                               unsigned long long *counters, hipStream_t stream);
// channel.hip
hipError_t launch_channel_llr(const uint8_t *ref, long long ref_stride, long long frames, int n, double sigma2,
                              double gain, int plus_is_bit, int prefix_len, double prefix_llr,
                              unsigned long long seed, unsigned long long first_frame, double cfo, double phase0,
                              void *llr, int llr_dtype, hipStream_t stream);

}  // namespace ldpc
