// Hard-decision bit flipping, /root/reference/ldpc.py:82-135 (bitFlipDecode), for any matrix.
//
// One frame per workgroup; bits, check parities and vote totals of the frame live in LDS.  A pass:
//   1. every check forms the parity of its bits (ldpc.py:120-121); all zero -> the frame is done
//      (ldpc.py:103-109; the test uses every row, the votes only the first `mact` rows, ldpc.py:113)
//   2. every check votes on each of its bits: +1 when the parity of the check's OTHER bits is 1,
//      -1 otherwise (ldpc.py:123-128); votes are summed per bit with integer LDS atomics
//   3. a bit flips when its total exceeds half the degree of the last row visited
//      (`len(Ej) / 2` with Ej left over from the loop, ldpc.py:130-132): 2 * total > degree
#include "extra_kernels.h"

namespace ldpc {

template <typename TL>
__global__ void __launch_bounds__(512)
bitflip_kernel(CsrDev c, int max_iters, int mact, const TL *__restrict__ llr, long long frames,
               uint8_t *__restrict__ hard_out, int *__restrict__ iters_out, uint8_t *__restrict__ conv_out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, NT = blockDim.x;
    const int n = c.n, m = c.m;
    int *votes = reinterpret_cast<int *>(smem);                 // [n]
    uint8_t *bits = reinterpret_cast<uint8_t *>(votes + n);     // [n]
    uint8_t *par = bits + ((n + 15) & ~15);                     // [m]
    const int last_deg = mact > 0 ? c.row_ptr[mact] - c.row_ptr[mact - 1] : 0;

    for (long long f = blockIdx.x; f < frames; f += gridDim.x) {
        for (int i = tid; i < n; i += NT) bits[i] = llr[f * n + i] > (TL)0 ? 0 : 1;  // ldpc.py:86-90
        __syncthreads();
        int t = 0, converged = 0;
        for (; t < max_iters; ++t) {
            int bad = 0;
            for (int j = tid; j < m; j += NT) {
                unsigned p = 0;
                for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; ++e) p ^= bits[c.col_idx[e]];
                par[j] = (uint8_t)p;
                bad |= (int)p;
            }
            for (int i = tid; i < n; i += NT) votes[i] = 0;
            if (!__syncthreads_or(bad)) { converged = 1; break; }
            for (int j = tid; j < mact; j += NT) {
                const unsigned p = par[j];
                for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; ++e) {
                    const int i = c.col_idx[e];
                    atomicAdd(&votes[i], (p ^ bits[i]) ? 1 : -1);
                }
            }
            __syncthreads();
            for (int i = tid; i < n; i += NT)
                if (2 * votes[i] > last_deg) bits[i] ^= 1;
            __syncthreads();
        }
        for (int i = tid; i < n; i += NT) hard_out[f * n + i] = bits[i];
        if (tid == 0) { iters_out[f] = t; conv_out[f] = (uint8_t)converged; }
        __syncthreads();
    }
}

hipError_t launch_bitflip(const CsrDev &code, int max_iters, int num_active_checks, const Launch &io, int num_cus,
                          const char **kernel_name, bool *supported)
{
    const size_t lds = (size_t)code.n * 4 + (((size_t)code.n + 15) & ~(size_t)15) + code.m + 16;
    *supported = lds <= 160 * 1024;
    if (!*supported) return hipSuccess;
    const int mact = num_active_checks < 0 || num_active_checks > code.m ? code.m : num_active_checks;
    int nt = 64;
    while (nt < 512 && nt * 2 <= code.n) nt *= 2;
    size_t per_cu = (160 * 1024) / lds;
    if (per_cu > 8) per_cu = 8;
    long long grid = (long long)num_cus * (long long)per_cu * 4;
    if (grid > io.frames) grid = io.frames;
    if (grid < 1) grid = 1;
    *kernel_name = "bitflip";
    hipError_t e;
    if (io.llr_dtype == LDPC_F64) {
        auto k = bitflip_kernel<double>;
        e = kernel_setup(reinterpret_cast<const void *>(k), nt, lds, nullptr);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(nt), lds, io.stream, code, max_iters, mact,
                           reinterpret_cast<const double *>(io.llr), io.frames, io.hard, io.iters, io.converged);
    } else {
        auto k = bitflip_kernel<float>;
        e = kernel_setup(reinterpret_cast<const void *>(k), nt, lds, nullptr);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(nt), lds, io.stream, code, max_iters, mact,
                           reinterpret_cast<const float *>(io.llr), io.frames, io.hard, io.iters, io.converged);
    }
    return hipGetLastError();
}

}  // namespace ldpc
