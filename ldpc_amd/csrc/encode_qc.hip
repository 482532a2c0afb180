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
//
// Random messages (msg == NULL): the reference draws `numpy.random.randint(0, 2, k)` per frame
// (ldpc.py:649); here message bit i of the frame with absolute index a is bit (i mod 128) of
// Philox4x32-10(counter = (a, i / 128, 1), key = seed) - a function of (seed, frame, bit) only, so
// that any split of a sweep into launches, ranks or GPUs sends the same messages.
#include "extra_kernels.h"
#include "sweep_fused.h"

namespace ldpc {

__global__ void __launch_bounds__(1024)
encode_qc_kernel(EncDev e, const uint8_t *__restrict__ msg, long long frames, uint8_t *__restrict__ cw, int F,
                 unsigned k0, unsigned k1, unsigned long long first_frame, uint8_t *__restrict__ msg_out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int Z = e.Z, tid = threadIdx.x;
    const bool lane_on = tid < F * Z;
    const int f = lane_on ? tid / Z : 0;
    const int z = lane_on ? tid - f * Z : 0;
    uint8_t *xs = smem + (size_t)f * e.n_cols * Z;
    const long long k = (long long)e.k_cols * Z, n = (long long)e.nb * Z;
    const int words = (int)((k + 127) / 128);                                        // Philox blocks per frame
    unsigned *rnd = reinterpret_cast<unsigned *>(smem + (((size_t)F * e.n_cols * Z + 15) & ~(size_t)15));  // [F][words][4]
    const long long groups = (frames + F - 1) / F;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long frame = grp * F + f;
        const bool active = lane_on && frame < frames;
        if (!msg) {
            for (int i = tid; i < F * words; i += blockDim.x) {
                const int f2 = i / words, w = i - f2 * words;
                const unsigned long long af = first_frame + (unsigned long long)(grp * F + f2);
                const Philox4 r = philox4x32_10((unsigned)af, (unsigned)(af >> 32), (unsigned)w, 1u, k0, k1);
                *reinterpret_cast<uint4 *>(rnd + (size_t)i * 4) = make_uint4(r.x, r.y, r.z, r.w);
            }
            __syncthreads();
        }
        for (int j = 0; j < e.k_cols; ++j) {
            const long long pos = (long long)j * Z + z;
            uint8_t b;
            if (msg) b = active ? (msg[frame * k + pos] & 1) : 0;
            else b = (uint8_t)((rnd[((size_t)f * words + (pos >> 7)) * 4 + ((pos >> 5) & 3)] >> (pos & 31)) & 1u);
            if (lane_on) xs[e.info_cols[j] * Z + z] = b;
            if (msg_out && active) msg_out[frame * k + pos] = b;
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

// Dense schedules of small circulants (Z <= 64: the CCSDS generator, 115 taps per row): a block
// column is a Z-bit mask - the ballot of its bits over the wave - and one step input is
//     out[z] ^= XOR over the taps s of col[(z + s) mod Z]  =  parity(popcount(col & rotl_Z(taps, z))),
// eight vector instructions per (output, input) column pair instead of about six per tap.  One frame
// per wave; masks of the block columns live in LDS (read back as broadcasts).  taps[step][column] is
// the Z-bit mask of the step's shifts for that source column (0 = not an input).
__global__ void __launch_bounds__(64)
encode_dense_kernel(EncDev e, const unsigned long long *__restrict__ taps, const uint8_t *__restrict__ msg,
                    long long frames, uint8_t *__restrict__ cw, unsigned k0, unsigned k1, unsigned long long first_frame,
                    uint8_t *__restrict__ msg_out)
{
    __shared__ unsigned long long col_mask[256];
    __shared__ __align__(16) unsigned rnd[4 * 32];
    const int Z = e.Z, z = threadIdx.x;
    const bool lane_on = z < Z;
    const unsigned long long zmask = Z >= 64 ? ~0ull : ((1ull << Z) - 1ull);
    const long long k = (long long)e.k_cols * Z, n = (long long)e.nb * Z;
    const int words = (int)((k + 127) / 128);
    for (long long frame = blockIdx.x; frame < frames; frame += gridDim.x) {
        if (!msg) {
            const unsigned long long af = first_frame + (unsigned long long)frame;
            for (int w = z; w < words; w += 64) {
                const Philox4 r = philox4x32_10((unsigned)af, (unsigned)(af >> 32), (unsigned)w, 1u, k0, k1);
                *reinterpret_cast<uint4 *>(rnd + w * 4) = make_uint4(r.x, r.y, r.z, r.w);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // other lanes read these words below
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        for (int j = 0; j < e.k_cols; ++j) {
            const long long pos = (long long)j * Z + z;
            unsigned b = 0;
            if (lane_on) b = msg ? (msg[frame * k + pos] & 1u) : ((rnd[(pos >> 7) * 4 + ((pos >> 5) & 3)] >> (pos & 31)) & 1u);
            const unsigned long long m = __ballot(b != 0);
            const int c = e.info_cols[j];
            if (z == 0) col_mask[c] = m;   // (read by every lane in the steps below: fenced after this loop)
            if (lane_on) {
                cw[frame * n + (long long)c * Z + z] = (uint8_t)b;
                if (msg_out) msg_out[frame * k + pos] = (uint8_t)b;
            }
        }
        for (int s = 0; s < e.n_steps; ++s) {
            // lane 0's column masks of the information columns / of the previous step, before any lane reads them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            unsigned acc = 0;
            for (int c = 0; c < e.n_cols; ++c) {
                const unsigned long long t = taps[(size_t)s * e.n_cols + c];  // the same for every lane
                if (t == 0ull) continue;
                unsigned long long r = (t << z) | (t >> ((Z - z) & 63));
                if (z == 0) r = t;
                r &= zmask & col_mask[c];
                acc += (unsigned)__popcll(r);
            }
            const unsigned long long m = __ballot(lane_on && (acc & 1u));
            const int c = e.step_out[s];
            if (z == 0) col_mask[c] = m;
            if (lane_on && c < e.nb) cw[frame * n + (long long)c * Z + z] = (uint8_t)(acc & 1u);
        }
    }
}

hipError_t launch_encode_qc(const EncDev &e, const uint8_t *msg, long long frames, uint8_t *cw, int num_cus,
                            hipStream_t stream, bool *supported, unsigned long long seed, unsigned long long first_frame,
                            uint8_t *msg_out, const unsigned long long *dense_taps)
{
    *supported = false;
    if (e.Z < 1 || e.Z > 1024) return hipSuccess;
    if (dense_taps) {  // dense schedule of small circulants: one frame per wave, ballot masks
        *supported = true;
        long long grid = (long long)num_cus * 32;
        if (grid > frames) grid = frames;
        hipLaunchKernelGGL(encode_dense_kernel, dim3((unsigned)grid), dim3(64), 0, stream, e, dense_taps, msg, frames, cw,
                           (unsigned)seed, (unsigned)(seed >> 32), first_frame, msg_out);
        return hipGetLastError();
    }
    int F = e.Z >= 256 ? 1 : 256 / e.Z;
    const size_t per_frame = (size_t)e.n_cols * e.Z, rnd_bytes = (((size_t)e.k_cols * e.Z + 127) / 128) * 16;
    while (F > 1 && F * (per_frame + rnd_bytes) > 64 * 1024) --F;
    const size_t lds = ((F * per_frame + 15) & ~(size_t)15) + F * rnd_bytes;
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
    hipLaunchKernelGGL(encode_qc_kernel, dim3((unsigned)grid), dim3(threads), lds, stream, e, msg, frames, cw, F,
                       (unsigned)seed, (unsigned)(seed >> 32), first_frame, msg_out);
    return hipGetLastError();
}

}  // namespace ldpc
