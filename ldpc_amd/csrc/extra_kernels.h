// Launchers of the kernels beside the belief-propagation decoders (bit flipping, encoder, fused
// sweep); kept out of bp_common.h so that adding one does not rebuild every decoder.
#pragma once
#include "bp_common.h"

namespace ldpc {

// decode_lnms_h.hip: layered min-sum in packed half precision, two frames per lane (compute_dtype LDPC_F16)
hipError_t launch_spec_lnms_half(int spec_id, const ldpc_decode_config &cfg, const Launch &io, int num_cus,
                                 const char **kernel_name, bool *supported);

// bitflip.hip: ldpc.py:82-135
hipError_t launch_bitflip(const CsrDev &code, int max_iters, int num_active_checks, const Launch &io, int num_cus,
                          const char **kernel_name, bool *supported);

// encode_qc.hip: circulant-schedule encoder (replaces G @ message % 2, ldpc.py:46)
struct EncDev {
    int Z, n_cols, nb, k_cols, n_steps;
    const int *info_cols;     // [k_cols] block column of each information block
    const int *step_out;      // [n_steps] block column a step produces
    const int *step_ptr;      // [n_steps + 1] its terms
    const int *step_level;    // [n_steps] non-decreasing; a step reads columns of earlier levels only
    const unsigned *terms;    // block column << 16 | shift
    // lifting sizes that are multiples of four (the word-wide kernel; NULL / 0 otherwise).  Only columns that some
    // term reads get a place ("slot") in LDS; slot n_slots4 stays zero.  Every step's terms are padded to a
    // multiple of four with terms that read the zero slot.
    int n_slots4;
    const int *info_slot4;    // [k_cols] slot of an information column, -1 = nothing reads it
    const int *step_slot4;    // [n_steps] slot of a step's output, -1 = nothing reads it
    const int *step_ptr4;     // [n_steps + 1], multiples of four
    const unsigned *terms4;   // per group of four terms twelve words: shifts & ~3, byte offsets of the slots, shifts & 3; 16-byte aligned
};
hipError_t launch_encode_qc(const EncDev &e, const uint8_t *msg, long long frames, uint8_t *cw, int num_cus,
                            hipStream_t stream, bool *supported, unsigned long long seed, unsigned long long first_frame,
                            uint8_t *msg_out, const unsigned long long *dense_taps);
// msg == NULL: random messages keyed by (seed, first_frame + frame); dense_taps != NULL: the ballot-mask
// kernel for dense schedules with Z <= 64, taps[step][column] = Z-bit mask of the step's shifts

// rate_match.hip: 5G NR bit selection / interleaving and its inverse on LLRs (TS 38.212 5.4.2)
hipError_t launch_rate_match(const ldpc_rate_match_config &c, const uint8_t *cw, long long cw_stride, long long frames,
                             uint8_t *out, hipStream_t stream);
hipError_t launch_rate_recover(const ldpc_rate_match_config &c, const void *llr, int llr_dtype, long long frames,
                               double filler_llr, int accumulate, void *out, int out_dtype, hipStream_t stream);

// count_errors.hip: words of one byte per bit -> eight positions per byte, lowest bit first (the packed
// reference words of ldpc_sweep_batch_packed)
hipError_t launch_pack_bits(const uint8_t *bytes, long long stride, long long frames, int n, uint8_t *bits,
                            long long bits_stride, hipStream_t stream);

}  // namespace ldpc
