// BPSK + white Gaussian noise + channel LLRs on the device: the step in front of the decode path
// (addNoiseBPSK, /root/reference/ldpc.py:241-266, and the LLR formulas at the top of every decode
// method: ldpc.py:143-151, bpsk.py:159-167, bpsk.py:271-277, channel.py:268-274, 410-416).
//
//   x   = 2b - 1 (bit 1 sent as +1, ldpc.py:257) or 1 - 2b (bit 0 sent as +1, the convention the
//         other decoders assume, bpsk.py:240-243)
//   y   = x + sqrt(sigma2) * N(0, 1)                       sigma2 = 1 / (2 * 10^(SNR/10)), ldpc.py:245
//   llr = gain * y            gain = 2 / sigma2 (ldpc.py:144) or 1 / sigma2 (bpsk.py:273)
//   llr[i] = prefix_llr for i < prefix_len                 the hard-coded puncture prefix, bpsk.py:163
//
// Doppler variant (ldpc_channel_llr_doppler; build-defined, the reference only plots the scale factor
// 1 - v/c, /root/reference/plotDoppler.py:14-16): the receiver follows the symbol timing but keeps the
// carrier phase it acquired at the start of the frame, so symbol i arrives rotated by
// 2 pi (cfo (i + 1/2) + phase0) with cfo = residual frequency offset in cycles per symbol, and an
// integrate-and-dump matched filter loses the factor sinc(cfo) inside each symbol:
//   y = x * sinc(cfo) * cos(2 pi (cfo (i + 1/2) + phase0)) + sqrt(sigma2) * N(0, 1)
// The LLR formula stays that of an AWGN receiver (it does not know about the rotation).
//
// The reference draws its noise from numpy's global generator; nothing can reproduce that
// stream, so the noise here is this build's own: Philox4x32-10 keyed by the caller's seed, counter
// = (absolute frame index, position / 4), two Box-Muller pairs per counter value.  A frame's noise
// therefore depends only on (seed, frame index, position) - not on the batch it is generated in
// or on the GPU that generates it, which is what lets ranks shard a sweep by frame ranges.
#include "sweep_fused.h"

namespace ldpc {

template <typename TL>
__global__ void __launch_bounds__(256)
channel_llr_kernel(const uint8_t *__restrict__ ref, long long ref_stride, long long frames, int n, ChannelDev ch,
                   TL *__restrict__ llr)
{
    const int quads = (n + 3) / 4;
    const long long total = frames * quads;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long f = i / quads;
        const int qd = (int)(i - f * quads);
        const bool whole = qd * 4 + 3 < n;
        unsigned bits = 0;  // byte j = reference bit of position 4 qd + j
        if (ref) {
            if (whole && (n & 3) == 0) bits = *reinterpret_cast<const unsigned *>(ref + f * ref_stride + qd * 4);
            else
                for (int j = 0; j < 4; ++j)
                    if (qd * 4 + j < n) bits |= (unsigned)ref[f * ref_stride + qd * 4 + j] << (8 * j);
        }
        float v[4];
        channel_quad(ch, ch.first_frame + (unsigned long long)f, qd, bits, v);
        bool stored = false;
        if constexpr (sizeof(TL) == 4) {  // one 16-byte store per thread
            if (whole && (n & 3) == 0) {
                *reinterpret_cast<float4 *>(llr + f * n + qd * 4) = make_float4(v[0], v[1], v[2], v[3]);
                stored = true;
            }
        }
        if (!stored) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (qd * 4 + j < n) llr[f * n + qd * 4 + j] = (TL)v[j];
        }
    }
}

hipError_t launch_channel_llr(const uint8_t *ref, long long ref_stride, long long frames, int n, double sigma2,
                              double gain, int plus_is_bit, int prefix_len, double prefix_llr,
                              unsigned long long seed, unsigned long long first_frame, double cfo, double phase0,
                              void *llr, int llr_dtype, hipStream_t stream)
{
    const ChannelDev ch = make_channel_dev(sigma2, gain, plus_is_bit, prefix_len, prefix_llr, seed, first_frame, cfo, phase0);
    const long long total = frames * ((n + 3) / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    if (llr_dtype == LDPC_F64)
        hipLaunchKernelGGL(channel_llr_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, stream, ref, ref_stride,
                           frames, n, ch, reinterpret_cast<double *>(llr));
    else
        hipLaunchKernelGGL(channel_llr_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, ref, ref_stride,
                           frames, n, ch, reinterpret_cast<float *>(llr));
    return hipGetLastError();
}

}  // namespace ldpc
