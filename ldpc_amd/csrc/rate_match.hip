// 5G NR rate matching around the decoder (3GPP TS 38.212 section 5.4.2: bit selection from the circular
// buffer and bit interleaving), on the device.  The reference fakes both ends of this step with
// constants: it cuts the matrix to its first 1400 rows (/root/reference/channel.py:35) - what a
// receiver does when the last extension parity bits were never sent - and overwrites the first 80
// LLRs (bpsk.py:163, channel.py:412) - the 2 Z punctured systematic bits of a Z = 40 code.  Here the
// mapping is the standard's, for any lifting size, redundancy version and code rate:
//
//   codeword position `offset + d`  <->  circular-buffer position d, 0 <= d < n_cb   (offset = 2 Z)
//   positions [filler_begin, filler_end) of the buffer are filler bits: known zeros, never sent
//   e_k = k-th position met when walking the buffer from k0, wrapping, skipping fillers (k < E)
//   f_{i + j Qm} = e_{i E/Qm + j}: what goes over the air (Qm = 1: no interleaving)
//
// rate_match_kernel: codeword bits -> transmitted bits f.  rate_recover_kernel: LLRs of f -> LLRs of
// the codeword positions the decoder uses, [0, n_out): 0 for what was not sent (punctured prefix,
// tail beyond the last transmitted bit), the sum over all laps for positions sent more than once,
// `filler_llr` for fillers; `accumulate` adds to what the buffer holds (soft combining of
// retransmissions).  Every output is a pure gather: no atomics.  HBM-bound byte work:
// n + E bytes per frame (match), 4 (E + n_out) bytes per frame (recover).
#include "extra_kernels.h"

namespace ldpc {

struct RmIndex {
    int n_cb, k0, fa, fb, M, rank0, E, rows;  // rows = E / Qm
    int Qm;
    __device__ __forceinline__ int rank(int d) const
    {
        int in = d - fa;
        in = in < 0 ? 0 : (in > fb - fa ? fb - fa : in);
        return d - in;
    }
    __device__ __forceinline__ int unrank(int r) const { return r < fa ? r : r + (fb - fa); }
    // index in f of e_k
    __device__ __forceinline__ int air(int k) const
    {
        if (Qm == 1) return k;
        const int i = k / rows, j = k - i * rows;
        return i + j * Qm;
    }
};

static RmIndex make_index(const ldpc_rate_match_config &c)
{
    RmIndex x;
    x.n_cb = c.n_cb; x.k0 = c.k0; x.E = c.E; x.Qm = c.Qm < 1 ? 1 : c.Qm;
    x.fa = c.filler_begin < c.n_cb ? c.filler_begin : c.n_cb;
    x.fb = c.filler_end < c.n_cb ? c.filler_end : c.n_cb;
    if (x.fb < x.fa) x.fb = x.fa;
    x.M = c.n_cb - (x.fb - x.fa);
    int in = c.k0 - x.fa;
    in = in < 0 ? 0 : (in > x.fb - x.fa ? x.fb - x.fa : in);
    x.rank0 = c.k0 - in;
    x.rows = c.E / x.Qm;
    return x;
}

// One thread per transmitted bit / per codeword position; blockIdx.y owns a contiguous run of frames,
// so the index arithmetic of a position is done once and reused for every frame of the run, a
// workgroup walks through memory frame by frame, and neighbouring threads touch neighbouring bytes
// of the same frame.
__global__ void __launch_bounds__(256)
rate_match_kernel(RmIndex x, int offset, const uint8_t *__restrict__ cw, long long cw_stride, long long frames,
                  uint8_t *__restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= x.E) return;
    int r = x.rank0 + k % x.M;
    if (r >= x.M) r -= x.M;
    const int from = offset + x.unrank(r), to = x.air(k);
    const long long run = (frames + gridDim.y - 1) / gridDim.y, f0 = blockIdx.y * run;
    const long long f1 = f0 + run < frames ? f0 + run : frames;
#pragma unroll 4
    for (long long f = f0; f < f1; ++f) out[f * x.E + to] = cw[f * cw_stride + from];
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256)
rate_recover_kernel(RmIndex x, int offset, int n_out, const TI *__restrict__ llr, long long frames, float filler_llr,
                    int accumulate, TO *__restrict__ out)
{
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n_out) return;
    const int d = pos - offset;
    const bool in_buffer = d >= 0 && d < x.n_cb, filler = in_buffer && d >= x.fa && d < x.fb;
    int idx = x.E;  // first lap on which this position is sent; E = never
    if (in_buffer && !filler) {
        idx = x.rank(d) - x.rank0;
        if (idx < 0) idx += x.M;
    }
    const bool once = idx + x.M >= x.E;  // sent at most once: the usual case
    const int a0 = idx < x.E ? x.air(idx) : 0;
    const long long run = (frames + gridDim.y - 1) / gridDim.y, f0 = blockIdx.y * run;
    const long long f1 = f0 + run < frames ? f0 + run : frames;
#pragma unroll 4
    for (long long f = f0; f < f1; ++f) {
        TO v = (TO)0;
        const TI *e = llr + f * x.E;
        if (once) {
            if (idx < x.E) v = (TO)e[a0];
        } else {
            for (int k = idx; k < x.E; k += x.M) v += (TO)e[x.air(k)];
        }
        TO *o = out + f * n_out + pos;
        if (filler) *o = (TO)filler_llr;
        else *o = accumulate ? *o + v : v;
    }
}

static dim3 rm_grid(int width, long long frames)
{
    const unsigned bx = (unsigned)((width + 255) / 256);
    long long by = 16384 / (bx ? bx : 1);  // about 16 K workgroups in all
    if (by > frames) by = frames;
    if (by > 65535) by = 65535;
    if (by < 1) by = 1;
    return dim3(bx, (unsigned)by);
}

hipError_t launch_rate_match(const ldpc_rate_match_config &c, const uint8_t *cw, long long cw_stride, long long frames,
                             uint8_t *out, hipStream_t stream)
{
    hipLaunchKernelGGL(rate_match_kernel, rm_grid(c.E, frames), dim3(256), 0, stream, make_index(c), c.offset,
                       cw, cw_stride, frames, out);
    return hipGetLastError();
}

hipError_t launch_rate_recover(const ldpc_rate_match_config &c, const void *llr, int llr_dtype, long long frames,
                               double filler_llr, int accumulate, void *out, int out_dtype, hipStream_t stream)
{
    const RmIndex x = make_index(c);
    const dim3 grid = rm_grid(c.n_out, frames), block(256);
#define LDPC_RM_GO(TI, TO)                                                                         \
    hipLaunchKernelGGL((rate_recover_kernel<TI, TO>), grid, block, 0, stream, x, c.offset, c.n_out, \
                       reinterpret_cast<const TI *>(llr), frames, (float)filler_llr, accumulate, reinterpret_cast<TO *>(out))
    if (llr_dtype == LDPC_F64 && out_dtype == LDPC_F64) LDPC_RM_GO(double, double);
    else if (llr_dtype == LDPC_F64) LDPC_RM_GO(double, float);
    else if (out_dtype == LDPC_F64) LDPC_RM_GO(float, double);
    else LDPC_RM_GO(float, float);
#undef LDPC_RM_GO
    return hipGetLastError();
}

}  // namespace ldpc
