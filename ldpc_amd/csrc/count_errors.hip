// Error accounting over a decoded batch (ldpc.py:166-170, 210-216; FER loop ldpc.py:644-667).
//
// counters[0] += frames, [1] += frames with converged == 0, [2] += frames that converged to a word
// other than the reference, [3] += codeword-bit errors over all n bits, [4] += sum of iters.
// One wave per frame at a time; per-workgroup partial sums, then one atomic per counter.
#include "bp_common.h"
#include "extra_kernels.h"

namespace ldpc {

template <bool WORDS>
__global__ void __launch_bounds__(256)
count_errors_kernel(const uint8_t *__restrict__ hard, const uint8_t *__restrict__ ref,
                    long long ref_stride, const int *__restrict__ iters,
                    const uint8_t *__restrict__ conv, long long frames, int n,
                    unsigned long long *__restrict__ counters)
{
    __shared__ unsigned long long part[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wave_id = (long long)blockIdx.x * 4 + wave;
    const long long nwaves = (long long)gridDim.x * 4;
    unsigned long long nonconv = 0, wrong = 0, biterr = 0, itsum = 0;
    for (long long f = wave_id; f < frames; f += nwaves) {
        const uint8_t *h = hard + f * n;
        const uint8_t *r = ref + f * ref_stride;
        unsigned e = 0;
        if (WORDS) {  // rows and strides are multiples of four bytes: four positions per load, four loads in flight
            const unsigned *h4 = reinterpret_cast<const unsigned *>(h), *r4 = reinterpret_cast<const unsigned *>(r);
            const int n4 = n >> 2;
            int i = lane;
            for (; i + 192 < n4; i += 256) {
                unsigned x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = h4[i + 64 * u] ^ r4[i + 64 * u];
#pragma unroll
                for (int u = 0; u < 4; ++u) e += __popc((((x[u] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x[u]) & 0x80808080u);  // bytes that differ
            }
            for (; i < n4; i += 64) {
                const unsigned x = h4[i] ^ r4[i];
                e += __popc((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u);
            }
        } else {
            for (int i = lane; i < n; i += 64) e += h[i] != r[i];
        }
        for (int off = 32; off > 0; off >>= 1) e += __shfl_down(e, off, 64);
        if (lane == 0) {
            const bool c = conv[f] != 0;
            nonconv += !c;
            wrong += c && e != 0;
            biterr += e;
            itsum += (unsigned long long)iters[f];
        }
    }
    if (lane == 0) { part[wave][0] = nonconv; part[wave][1] = wrong; part[wave][2] = biterr; part[wave][3] = itsum; }
    __syncthreads();
    if (threadIdx.x < 4) {
        unsigned long long s = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (s) atomicAdd(&counters[1 + threadIdx.x], s);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&counters[0], (unsigned long long)frames);
}

hipError_t launch_count_errors(const uint8_t *hard, const uint8_t *ref, long long ref_stride,
                               const int *iters, const uint8_t *conv, long long frames, int n,
                               unsigned long long *counters, hipStream_t stream)
{
    long long blocks = (frames + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    const bool words = n % 4 == 0 && ref_stride % 4 == 0 && ((reinterpret_cast<uintptr_t>(hard) | reinterpret_cast<uintptr_t>(ref)) & 3u) == 0;
    if (words)
        hipLaunchKernelGGL(count_errors_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, hard, ref,
                           ref_stride, iters, conv, frames, n, counters);
    else
        hipLaunchKernelGGL(count_errors_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, hard, ref,
                           ref_stride, iters, conv, frames, n, counters);
    return hipGetLastError();
}

// One thread per output byte: positions 8 i .. 8 i + 7 of a frame, lowest bit first; positions beyond n are 0.
template <bool WORDS>
__global__ void pack_bits_kernel(const uint8_t *__restrict__ bytes, long long stride, long long frames, int n,
                                 uint8_t *__restrict__ bits, long long bits_stride)
{
    const int per = (n + 7) / 8;
    const long long total = frames * per;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long f = i / per;
        const int b = (int)(i - f * per);
        unsigned v = 0;
        if (WORDS && 8 * b + 8 <= n) {  // rows eight-byte aligned: the eight positions in one load, their low bits gathered by a multiply
            const uint2 w = *reinterpret_cast<const uint2 *>(bytes + f * stride + 8 * b);
            constexpr unsigned M = (1u << 24) | (1u << 17) | (1u << 10) | (1u << 3);  // byte j's low bit to bit 24 + j, no two products on one bit
            v = (((w.x & 0x01010101u) * M) >> 24) | ((((w.y & 0x01010101u) * M) >> 24) << 4);
        } else {
            for (int k = 0; k < 8; ++k) {
                const int pos = 8 * b + k;
                if (pos < n) v |= (unsigned)(bytes[f * stride + pos] & 1u) << k;
            }
        }
        bits[f * bits_stride + b] = (uint8_t)v;
    }
}

hipError_t launch_pack_bits(const uint8_t *bytes, long long stride, long long frames, int n, uint8_t *bits,
                            long long bits_stride, hipStream_t stream)
{
    const long long total = frames * ((n + 7) / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    if (stride % 8 == 0 && (reinterpret_cast<uintptr_t>(bytes) & 7u) == 0)
        hipLaunchKernelGGL(pack_bits_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, bytes, stride, frames, n, bits, bits_stride);
    else
        hipLaunchKernelGGL(pack_bits_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, bytes, stride, frames, n, bits, bits_stride);
    return hipGetLastError();
}

}  // namespace ldpc
