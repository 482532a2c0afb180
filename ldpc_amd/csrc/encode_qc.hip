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

// Lifting sizes that are multiples of four (every 5G NR size from 8 on): thread (f, q) owns positions
// 4 q .. 4 q + 3 of every block column of frame slot f as the four bytes of one word.  A term's input
// word - positions (4 q + s) mod Z and the three behind it - straddles two aligned words of the source
// column when s is not a multiple of four: both are read (a column carries a copy of its first word
// behind its last, so the second read needs no wrap) and one v_alignbyte_b32 cuts the four bytes out;
// the byte shift s mod 4 is the same for every lane.  Six vector instructions and one LDS read per
// term and FOUR positions where the kernel above spends about seven per position, and messages and
// codewords move as words (256 bytes per wave instruction instead of 64).
// Only the columns that some term reads live in LDS (5G NR base graph 1: 26 of 68 - the 42 extension
// parity columns are written once and never read); every column goes to the codeword as it is produced.
// The schedule arrives as separate read-only pointers: their words are the same for every lane and come
// by scalar loads, four terms of a step at a time and already decoded (word-aligned shift, byte offset of the
// source slot, byte shift; the steps' term lists are padded to multiples of four with terms that read an
// all-zero slot), so that four LDS reads are in flight together and the scalar unit only loads.
__global__ void __launch_bounds__(1024)
encode_qc4_kernel(int Z, int n_slots, int nb, int k_cols, int n_steps, const int *__restrict__ info_cols,
                  const int *__restrict__ info_slot, const int *__restrict__ step_out, const int *__restrict__ step_slot,
                  const int *__restrict__ step_ptr4, const int *__restrict__ step_level, const unsigned *__restrict__ terms4,
                  const uint8_t *__restrict__ msg, long long frames, uint8_t *__restrict__ cw, int F, unsigned k0,
                  unsigned k1, unsigned long long first_frame, uint8_t *__restrict__ msg_out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int Q = Z >> 2, CS = Z + 4, tid = threadIdx.x;  // CS: bytes of a column with its wrap word
    const bool lane_on = tid < F * Q;
    const int f = lane_on ? tid / Q : 0;
    const unsigned q4 = lane_on ? (unsigned)(tid - f * Q) * 4u : 0u;
    unsigned char *xs = smem + (size_t)f * (n_slots + 1) * CS;  // the slots and the zero slot
    const long long k = (long long)k_cols * Z, n = (long long)nb * Z;
    const int words = (int)((k + 127) / 128);                                       // Philox blocks per frame
    unsigned *rnd = reinterpret_cast<unsigned *>(smem + (((size_t)F * (n_slots + 1) * CS + 15) & ~(size_t)15));  // [F][words][4]
    auto put = [&](int slot, unsigned w) {  // idle lanes alias lane (0, 0) and do not write
        if (lane_on && slot >= 0) {
            *reinterpret_cast<unsigned *>(xs + slot * CS + q4) = w;
            if (q4 == 0) *reinterpret_cast<unsigned *>(xs + slot * CS + Z) = w;
        }
    };
    put(n_slots, 0u);
    const long long groups = (frames + F - 1) / F;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long frame = grp * F + f;
        const bool active = lane_on && frame < frames;
        uint8_t *out = cw + frame * n + q4;
        if (!msg) {
            for (int i = tid; i < F * words; i += blockDim.x) {
                const int f2 = i / words, w = i - f2 * words;
                const unsigned long long af = first_frame + (unsigned long long)(grp * F + f2);
                const Philox4 r = philox4x32_10((unsigned)af, (unsigned)(af >> 32), (unsigned)w, 1u, k0, k1);
                *reinterpret_cast<uint4 *>(rnd + (size_t)i * 4) = make_uint4(r.x, r.y, r.z, r.w);
            }
            __syncthreads();
        }
#pragma unroll 4
        for (int j = 0; j < k_cols; ++j) {
            const long long pos = (long long)j * Z + q4;  // a multiple of four: its four bits share a random word
            unsigned w;
            if (msg) w = active ? (*reinterpret_cast<const unsigned *>(msg + frame * k + pos) & 0x01010101u) : 0u;
            else {
                const unsigned nib = (rnd[((size_t)f * words + (pos >> 7)) * 4 + ((pos >> 5) & 3)] >> (pos & 31)) & 15u;
                w = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
            }
            put(info_slot[j], w);
            if (active) {
                *reinterpret_cast<unsigned *>(out + (long long)info_cols[j] * Z) = w;
                if (msg_out) *reinterpret_cast<unsigned *>(msg_out + frame * k + pos) = w;
            }
        }
        int level = -1;
        for (int s = 0; s < n_steps; ++s) {
            if (step_level[s] != level) {  // the same for every thread
                __syncthreads();
                level = step_level[s];
            }
            const int t1 = step_ptr4[s + 1];
            unsigned acc = 0;
            for (int t = step_ptr4[s]; t < t1; t += 4) {
                const unsigned *grp4 = terms4 + 3 * t;  // the group's twelve words, decoded on the host
                const uint4 sh = *reinterpret_cast<const uint4 *>(grp4), at = *reinterpret_cast<const uint4 *>(grp4 + 4),
                            bs = *reinterpret_cast<const uint4 *>(grp4 + 8);
                const unsigned s4[4] = {sh.x, sh.y, sh.z, sh.w}, base[4] = {at.x, at.y, at.z, at.w}, by[4] = {bs.x, bs.y, bs.z, bs.w};
                unsigned lo[4], hi[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    unsigned o = q4 + s4[u];  // aligned offset of the word that holds position (4 q + s) mod Z
                    o = o >= (unsigned)Z ? o - (unsigned)Z : o;
                    const unsigned char *src = xs + base[u] + o;
                    lo[u] = *reinterpret_cast<const unsigned *>(src);
                    hi[u] = *reinterpret_cast<const unsigned *>(src + 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc ^= __builtin_amdgcn_alignbyte(hi[u], lo[u], by[u]);
            }
            put(step_slot[s], acc);
            const int c = step_out[s];
            if (active && c < nb) *reinterpret_cast<unsigned *>(out + (long long)c * Z) = acc;
        }
        __syncthreads();  // the slots are read out before the next group overwrites them
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
    const size_t rnd_bytes = (((size_t)e.k_cols * e.Z + 127) / 128) * 16;
    const bool word_aligned = ((reinterpret_cast<uintptr_t>(msg) | reinterpret_cast<uintptr_t>(cw) | reinterpret_cast<uintptr_t>(msg_out)) & 3u) == 0;
    if (e.terms4 && word_aligned) {  // four positions per thread (encode_qc4_kernel)
        const size_t per_frame4 = (size_t)(e.n_slots4 + 1) * (e.Z + 4);
        const int Q = e.Z / 4;
        // frame slots per workgroup: the count that keeps most lanes of a compute unit busy within LDS and the wave limit
        int best_f = 0;
        double best = 0.0;
        for (int f = 1; f <= 64 && f * Q <= 1024; ++f) {
            const size_t lds = ((f * per_frame4 + 15) & ~(size_t)15) + f * rnd_bytes;
            if (lds > 160 * 1024) break;
            const int threads = ((f * Q + 63) / 64) * 64;
            long long per_cu = (long long)((160 * 1024) / lds);
            if (per_cu > 2048 / threads) per_cu = 2048 / threads;
            const double score = (double)per_cu * f * ((double)(f * Q) / threads);
            if (score > best * 1.02) { best = score; best_f = f; }
        }
        if (best_f > 0) {
            const int F4 = best_f, threads = ((F4 * Q + 63) / 64) * 64;
            const size_t lds = ((F4 * per_frame4 + 15) & ~(size_t)15) + F4 * rnd_bytes;
            *supported = true;
            hipError_t err = kernel_setup(reinterpret_cast<const void *>(encode_qc4_kernel), threads, lds, nullptr);
            if (err != hipSuccess) return err;
            const long long groups = (frames + F4 - 1) / F4;
            long long per_cu = (long long)((160 * 1024) / lds);
            if (per_cu > 2048 / threads) per_cu = 2048 / threads;
            if (per_cu < 1) per_cu = 1;
            long long grid = (long long)num_cus * per_cu * 2;
            if (grid > groups) grid = groups;
            if (grid < 1) grid = 1;
            hipLaunchKernelGGL(encode_qc4_kernel, dim3((unsigned)grid), dim3(threads), lds, stream, e.Z, e.n_slots4, e.nb, e.k_cols,
                               e.n_steps, e.info_cols, e.info_slot4, e.step_out, e.step_slot4, e.step_ptr4, e.step_level, e.terms4, msg, frames, cw, F4,
                               (unsigned)seed, (unsigned)(seed >> 32), first_frame, msg_out);
            return hipGetLastError();
        }
    }
    int F = e.Z >= 256 ? 1 : 256 / e.Z;
    const size_t per_frame = (size_t)e.n_cols * e.Z;
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
