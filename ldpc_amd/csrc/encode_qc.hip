// Encoder for quasi-cyclic codes on the device: the step in front of the channel in the reference's
// Monte-Carlo loop (`numpy.dot(self.G, message) % 2`, /root/reference/ldpc.py:46, with G from
// pyldpc.coding_matrix, ldpc.py:369).  A dense generator of a lifted 5G matrix would be 8448 x 26112
// bits; a circulant code needs none.  The host turns the code into a SCHEDULE over block columns
// (Z-vectors of bits):
//
//     step:  x[out] = XOR over its terms (col, s) of roll(x[col], -s),   i.e.  out[z] ^= col[(z + s) mod Z]
//
// Information block columns are given; every step produces one more block column from columns that
// are already known.  Two kinds of schedule come out of ldpc_amd/encoder.py:
//   * parity structure (5G NR, 802.11n): the core parity column whose three circulants leave one
//     when the core rows are added up, then every other parity column by substitution in a row
//     where it is the only unknown;
//   * block-circulant systematic generator (CCSDS): parity column j = sum over information columns
//     i and over the taps of circulant (i, j) of the generator.
// Steps carry a level; steps of one level only read columns of earlier levels, so a workgroup
// needs one barrier per level, not per step.
//
// Thread (f, z) owns position z of every block column of frame slot f; a frame's block columns
// (codeword + temporaries) live in LDS, one byte per bit.
#include "extra_kernels.h"

namespace ldpc {

__global__ void __launch_bounds__(1024)
encode_qc_kernel(EncDev e, const uint8_t *__restrict__ msg, long long frames, uint8_t *__restrict__ cw, int F)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int Z = e.Z, tid = threadIdx.x;
    const bool lane_on = tid < F * Z;
    const int f = lane_on ? tid / Z : 0;
    const int z = lane_on ? tid - f * Z : 0;
    uint8_t *xs = smem + (size_t)f * e.n_cols * Z;
    const long long k = (long long)e.k_cols * Z, n = (long long)e.nb * Z;
    const long long groups = (frames + F - 1) / F;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long frame = grp * F + f;
        const bool active = lane_on && frame < frames;
        for (int j = 0; j < e.k_cols; ++j) {
            const uint8_t b = active ? (msg[frame * k + (long long)j * Z + z] & 1) : 0;
            if (lane_on) xs[e.info_cols[j] * Z + z] = b;
        }
        int level = -1;
        for (int s = 0; s < e.n_steps; ++s) {
            if (e.step_level[s] != level) {  // the same for every thread
                __syncthreads();
                level = e.step_level[s];
            }
            unsigned acc = 0;
            for (int t = e.step_ptr[s]; t < e.step_ptr[s + 1]; ++t) {
                const unsigned term = e.terms[t];
                int zz = z + (int)(term & 0xffffu);
                if (zz >= Z) zz -= Z;
                acc ^= xs[(term >> 16) * Z + zz];
            }
            if (lane_on) xs[e.step_out[s] * Z + z] = (uint8_t)acc;
        }
        __syncthreads();
        if (active)
            for (int c = 0; c < e.nb; ++c) cw[frame * n + (long long)c * Z + z] = xs[c * Z + z];
        __syncthreads();
    }
}

hipError_t launch_encode_qc(const EncDev &e, const uint8_t *msg, long long frames, uint8_t *cw, int num_cus,
                            hipStream_t stream, bool *supported)
{
    *supported = false;
    if (e.Z < 1 || e.Z > 1024) return hipSuccess;
    int F = e.Z >= 256 ? 1 : 256 / e.Z;
    size_t per_frame = (size_t)e.n_cols * e.Z;
    while (F > 1 && F * per_frame > 64 * 1024) --F;
    const size_t lds = F * per_frame;
    if (lds > 160 * 1024) return hipSuccess;
    *supported = true;
    const int threads = ((F * e.Z + 63) / 64) * 64;
    hipError_t err = kernel_setup(reinterpret_cast<const void *>(encode_qc_kernel), threads, lds, nullptr);
    if (err != hipSuccess) return err;
    const long long groups = (frames + F - 1) / F;
    long long per_cu = (long long)((160 * 1024) / (lds ? lds : 1));
    if (per_cu > 2048 / threads) per_cu = 2048 / threads;
    if (per_cu < 1) per_cu = 1;
    long long grid = (long long)num_cus * per_cu * 2;
    if (grid > groups) grid = groups;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(encode_qc_kernel, dim3((unsigned)grid), dim3(threads), lds, stream, e, msg, frames, cw, F);
    return hipGetLastError();
}

}  // namespace ldpc
