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

__global__ void __launch_bounds__(256)
rate_match_kernel(RmIndex x, int offset, const uint8_t *__restrict__ cw, long long cw_stride, long long frames,
                  uint8_t *__restrict__ out)
{
    const long long total = frames * x.E;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long f = i / x.E;
        const int k = (int)(i - f * x.E);
        int r = x.rank0 + k % x.M;
        if (r >= x.M) r -= x.M;
        out[f * x.E + x.air(k)] = cw[f * cw_stride + offset + x.unrank(r)];
    }
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256)
rate_recover_kernel(RmIndex x, int offset, int n_out, const TI *__restrict__ llr, long long frames, float filler_llr,
                    int accumulate, TO *__restrict__ out)
{
    const long long total = frames * n_out;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long f = i / n_out;
        const int pos = (int)(i - f * n_out), d = pos - offset;
        TO v = (TO)0;
        bool filler = false;
        if (d >= 0 && d < x.n_cb) {
            if (d >= x.fa && d < x.fb) {
                filler = true;
            } else {
                int idx = x.rank(d) - x.rank0;
                if (idx < 0) idx += x.M;
                const TI *e = llr + f * x.E;
                for (int k = idx; k < x.E; k += x.M) v += (TO)e[x.air(k)];
            }
        }
        if (filler) out[i] = (TO)filler_llr;
        else out[i] = accumulate ? out[i] + v : v;
    }
}

static int rm_blocks(long long total)
{
    long long b = (total + 255) / 256;
    return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

hipError_t launch_rate_match(const ldpc_rate_match_config &c, const uint8_t *cw, long long cw_stride, long long frames,
                             uint8_t *out, hipStream_t stream)
{
    hipLaunchKernelGGL(rate_match_kernel, dim3(rm_blocks(frames * c.E)), dim3(256), 0, stream, make_index(c), c.offset,
                       cw, cw_stride, frames, out);
    return hipGetLastError();
}

hipError_t launch_rate_recover(const ldpc_rate_match_config &c, const void *llr, int llr_dtype, long long frames,
                               double filler_llr, int accumulate, void *out, int out_dtype, hipStream_t stream)
{
    const RmIndex x = make_index(c);
    const dim3 grid(rm_blocks(frames * c.n_out)), block(256);
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
