/*
 * ldpc_amd — C ABI of the MI355X (gfx950) batched LDPC belief-propagation decoder.
 *
 * The reference (SebPelay0/ldpc) has no FFI: its decode path is the Python method surface of
 * `class LDPCEncoder`, one frame per call.  Each entry point below names the reference code it
 * stands in for.  All pointers are plain C; `*_dev` pointers are device (HIP) addresses on the
 * current device, everything else is host memory.  No torch types cross this boundary.
 *
 * Every function returns LDPC_OK (0) or a positive error code; ldpc_last_error() gives the text
 * of the calling thread's last failure.  A frame that does not converge is DATA
 * (converged[f] == 0), never an error — the reference signals it by returning None
 * (FRAME_ERROR, /root/reference/ldpc.py:15,227).
 */
#ifndef LDPC_AMD_H
#define LDPC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDPC_OK 0
#define LDPC_ERR_ARG 1          /* bad argument (null pointer, bad size, bad enum)            */
#define LDPC_ERR_HIP 2          /* a HIP runtime call failed; text holds hipGetErrorString      */
#define LDPC_ERR_UNSUPPORTED 3  /* valid request this build cannot serve (e.g. degree > 64)     */
#define LDPC_ERR_NO_DEVICE 4    /* no gfx950 device visible                                    */

/* update rules — the five decoders present in the reference */
#define LDPC_RULE_R0 0 /* /root/reference/ldpc.py:137-227    minSumDecode(codeword, M)            */
#define LDPC_RULE_R1 1 /* /root/reference/bpsk.py:154-260    minSumDecode(codeword)               */
#define LDPC_RULE_R2 2 /* /root/reference/bpsk.py:263-387    sumProductDecodeTest(codeword)       */
#define LDPC_RULE_R3 3 /* /root/reference/channel.py:402-513 sumProductDecodeTest(codeword)       */
#define LDPC_RULE_R4 4 /* /root/reference/channel.py:260-398 sumProductDecode(codeword) + phi     */
/*
 * Row-layered normalised min-sum.  NO counterpart in the reference (it only has flooding
 * schedules); a build extension for the large 5G codes whose flooding messages do not fit on
 * chip.  Checks are visited in ascending order; each takes q = P - r_old from the current
 * posteriors P, forms r_new = scale * min-of-the-others * sign-product-of-the-others and writes
 * P = q + r_new back at once.  A variable that takes part in one check only sends that check its
 * channel LLR itself (its exact extrinsic value).  Iteration count, syndrome test at the top of
 * each iteration, LLR polarity and hard decision are those of R1.  Served for handles with a
 * circulant description (ldpc_code_attach_qc) without double-circulant blocks (all checks of a
 * block row then touch distinct variables), check degree <= 24 and posteriors that fit LDS;
 * LDPC_ERR_UNSUPPORTED otherwise.
 */
#define LDPC_RULE_LNMS 5
/*
 * Offset min-sum, flooding schedule.  NO counterpart in the reference (its only min-sum correction
 * is the factor 0.75 of /root/reference/bpsk.py:233); a build extension that sits beside R1 and
 * keeps every convention of it (LLR polarity, sign(0) := +1, ascending summation order, syndrome
 * test at the top of each iteration, iteration count).  The check-to-variable magnitude is
 *     max(scale * min_{k != i} |q_k| - offset, 0)
 * evaluated in that order; scale = 1 gives the textbook offset rule, offset = 0 gives R1.
 */
#define LDPC_RULE_OMS 6

/*
 * Hard-decision bit flipping, /root/reference/ldpc.py:82-135 bitFlipDecode(codeword): bits are
 * `0 if sample > 0 else 1`; at most max_iters (10, ldpc.py:102) passes, each preceded by the syndrome
 * test; in a pass every check among the first num_active_checks rows (int(self.m), ldpc.py:113)
 * sends each of its bits +1 when the parity of its OTHER bits is 1 and -1 otherwise, and a bit is
 * flipped when its total exceeds half the degree of the LAST row visited (`len(Ej) / 2` after the
 * loop, ldpc.py:130-132).  Only the sign of the input matters; integer arithmetic throughout.
 * iters = passes made; converged = the syndrome test passed (no test after the last pass).
 * posterior_dev must be NULL; compute_dtype is ignored.
 */
#define LDPC_RULE_BF 7

#define LDPC_F32 0
#define LDPC_F64 1
/*
 * compute_dtype only: packed half precision, two frames per lane - a reduced-precision fast mode with
 * no counterpart in the reference (which computes in float64).  Not bit-exact with anything; meant for
 * error-rate sweeps, validated statistically against the float32 kernels.  Served for rule R1 on the
 * small compiled-in codes and for LDPC_RULE_LNMS on the compiled-in codes without double-circulant
 * blocks (half-precision posteriors, check magnitudes kept as eight bits; a posterior of exactly zero
 * decides 0 there), with float LLRs and posterior_dev = NULL; LDPC_ERR_UNSUPPORTED otherwise.
 */
#define LDPC_F16 2

#define LDPC_KERNEL_AUTO 0 /* circulant kernel when the code has QC structure, else edge-list   */
#define LDPC_KERNEL_CSR 1  /* generic edge-list kernel                                          */
#define LDPC_KERNEL_QC 2   /* circulant kernel (error if the handle carries no QC structure)    */
/*
 * LDPC_RULE_LNMS only: the kernel compiled per 5G NR base graph, with lifting size, shift table and
 * number of active block rows as run-time arguments.  Chosen by LDPC_KERNEL_AUTO for every handle
 * whose circulant description is a prefix of the block rows of BG1 or BG2 (any Z up to 384, any
 * shifts) unless the exact table is compiled in; this selector forces it (error if the handle does
 * not match a base graph).
 */
#define LDPC_KERNEL_GRAPH 3

/*
 * The constants the reference edits in its source, per rule (presets: ldpc_amd/config.py):
 *   max_iters         ldpc.py:161 (10), bpsk.py:200 (30), bpsk.py:309 (50), channel.py:448 (50),
 *                     channel.py:306 (40)
 *   scale             ldpc.py:196 / bpsk.py:233 (0.75), channel.py:474 (0.85)
 *   clip_tanh_arg     bpsk.py:333 (20); 0 = no clip
 *   clip_prod         bpsk.py:337 (0.999999999), channel.py:473 (0.99999999)
 *   clip_post         bpsk.py:370 (30), channel.py:491 (50), channel.py:367 (500); 0 = no clip
 *   early_exit        1 = the reference's loop: syndrome check at the top of every iteration,
 *                     none after the last update; iters = numIterations (0..max_iters-1).
 *                     0 = fixed work: max_iters updates, iters = max_iters, converged = final
 *                     syndrome (no counterpart in the reference; for throughput runs).
 *   float_syndrome0   R0: the iteration-0 check is made on the raw LLR floats
 *                     (ldpc.py:158,165: numpy.dot(H, llr) % 2 == 0).
 *   num_active_checks R0: rows [0, k) take part in the check-node update (self.m, ldpc.py:30,175);
 *                     the syndrome always uses every row.  -1 = all.
 *                     LDPC_RULE_LNMS: the decoder sees the sub-matrix of rows [0, k) ONLY (sweeps,
 *                     syndrome test and the meaning of "degree one"): what 5G rate matching leaves
 *                     of the matrix when extension parity bits were not sent (the reference cuts
 *                     the matrix by hand, /root/reference/channel.py:35).  Variables in none of
 *                     those rows decide from their channel LLR.  k must be a whole number of block
 *                     rows of a handle that matches an NR base graph (LDPC_KERNEL_GRAPH).
 *   num_refresh_vars  R1-R4: variables [0, k) refresh their variable-to-check messages
 *                     (range(self.n), bpsk.py:244); -1 = all.
 *                     (Both bounds are served by the table-driven circulant kernels and the
 *                     edge-list kernel; the half-precision mode refuses them, layered min-sum all but the
 *                     block-row prefix above.)
 *   offset            LDPC_RULE_OMS only: the offset beta (no counterpart in the reference).
 */
typedef struct ldpc_decode_config {
    int32_t rule;
    int32_t max_iters;
    int32_t early_exit;
    int32_t float_syndrome0;
    int32_t num_active_checks;
    int32_t num_refresh_vars;
    double scale;
    double clip_tanh_arg;
    double clip_prod;
    double clip_post;
    double offset;
} ldpc_decode_config;

typedef struct ldpc_code ldpc_code; /* opaque: compiled parity-check matrix resident on one device */

/* text of the calling thread's most recent error ("" if none) */
const char *ldpc_last_error(void);
/* ABI version, bumped on any incompatible change */
int32_t ldpc_abi_version(void);
/* number of visible HIP devices; LDPC_ERR_NO_DEVICE if none */
int32_t ldpc_device_count(int32_t *count);

/*
 * Compile a parity-check matrix given as a CSR edge list (row_ptr[m+1], col_idx[E], columns
 * ascending inside each row) into device-resident layouts on the current device.
 * Replaces the dense `self.H` the reference scans with numpy.where on every check of every
 * iteration (/root/reference/ldpc.py:176, bpsk.py:214,236) after readMatrix / loadmat
 * (/root/reference/ldpc.py:360-366, ccsds.py:6-8, channel.py:34).
 */
int32_t ldpc_code_create_csr(int32_t m, int32_t n, const int32_t *row_ptr, const int32_t *col_idx,
                             ldpc_code **out);
/*
 * Attach a quasi-cyclic description to a handle: H is an mb x nb array of Z x Z blocks; base edge
 * e joins check (edge_row[e], z) with variable (edge_col[e], (z + edge_shift[e]) mod Z).
 * Edges sorted by (row, col, shift).  The description must expand to exactly the CSR matrix the
 * handle was created from (checked).  No counterpart in the reference (it never exploits the
 * circulant structure of its matrices).
 */
int32_t ldpc_code_attach_qc(ldpc_code *code, int32_t mb, int32_t nb, int32_t Z, int32_t n_base_edges,
                            const int32_t *edge_row, const int32_t *edge_col,
                            const int32_t *edge_shift);
int32_t ldpc_code_destroy(ldpc_code *code);
/* m, n, number of edges, lifting size (0 if no QC description); any out pointer may be NULL */
int32_t ldpc_code_info(const ldpc_code *code, int32_t *m, int32_t *n, int64_t *edges, int32_t *Z);

/*
 * Decode `frames` frames.  Replaces the loop body of LDPCEncoder.minSumDecode /
 * sumProductDecode(Test) (file:line per rule above) from the point where the channel LLRs exist
 * (ldpc.py:151) to the return (ldpc.py:170 / 227), for a whole batch per call.
 *
 *   llr_dev        [frames, n] row-major channel LLRs, float (llr_dtype LDPC_F32) or double
 *   compute_dtype  arithmetic type of the message passing (LDPC_F64 reproduces the reference's
 *                  float64 arithmetic bit for bit under R0/R1)
 *   hard_dev       [frames, n] uint8 hard decisions (the reference's hardDecisions)
 *   iters_dev      [frames] int32, the reference's self.numIterations
 *   converged_dev  [frames] uint8, 0 where the reference would return FRAME_ERROR
 *   posterior_dev  NULL, or [frames, n] of compute_dtype: the reference's bitNodes at exit
 *   stream         hipStream_t (NULL = default stream); the call only enqueues work
 */
int32_t ldpc_decode_batch(ldpc_code *code, const ldpc_decode_config *cfg, const void *llr_dev,
                          int32_t llr_dtype, int32_t compute_dtype, int32_t kernel, int64_t frames,
                          uint8_t *hard_dev, int32_t *iters_dev, uint8_t *converged_dev,
                          void *posterior_dev, void *stream);

/*
 * ldpc_decode_batch for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  NO counterpart in the reference.  Same
 * arguments, same results.  The kernels that keep a frame's state in global memory (csr_global_*,
 * qc_lnms_global_*: codes too large for LDS) take their per-workgroup slices from a file-scope
 * device array of ldpc_noalloc_scratch_bytes() bytes (32 MiB, held on every GPU the library runs
 * on until the process ends) instead of the handle's growable workspace, and start only as many
 * workgroups as it holds slices for.  The call allocates and frees nothing.  A code whose state
 * for one frame exceeds the array is refused with LDPC_ERR_UNSUPPORTED.  Every call on a GPU
 * shares that GPU's array: calls that overlap in time on one GPU (different streams or threads)
 * race; serialise them.  The first call per device and kernel looks up the array's address and the
 * kernel's launch limits (no allocation, but runtime queries): under HIP graph capture make one call
 * with the same code, rule and arithmetic outside the capture first.
 */
int32_t ldpc_decode_batch_noalloc(ldpc_code *code, const ldpc_decode_config *cfg, const void *llr_dev,
                                  int32_t llr_dtype, int32_t compute_dtype, int32_t kernel, int64_t frames,
                                  uint8_t *hard_dev, int32_t *iters_dev, uint8_t *converged_dev,
                                  void *posterior_dev, void *stream);
/* size in bytes of the scratch array behind ldpc_decode_batch_noalloc */
int64_t ldpc_noalloc_scratch_bytes(void);
/*
 * Device memory the handle's growable workspaces hold right now (the kernels that keep frame state in
 * global memory own one buffer per stream they were launched on, at most eight; each is sized for the
 * largest batch it has served, not for the device).  No counterpart in the reference; for tests and for
 * embedders who account for device memory.
 */
int64_t ldpc_code_workspace_bytes(const ldpc_code *code);

/*
 * Same, host buffers in and out (allocates device staging, copies, decodes, copies back,
 * synchronises).  This is what a one-frame-per-call caller such as the reference's drivers
 * (/root/reference/ldpc.py:392,654, bpsk.py:609,824) binds.
 */
int32_t ldpc_decode_batch_host(ldpc_code *code, const ldpc_decode_config *cfg, const void *llr,
                               int32_t llr_dtype, int32_t compute_dtype, int32_t kernel,
                               int64_t frames, uint8_t *hard, int32_t *iters, uint8_t *converged,
                               void *posterior);

/*
 * Error accounting over a decoded batch (/root/reference/ldpc.py:166-170,210-216 and the FER loop
 * ldpc.py:644-667): adds into counters_dev[5] (uint64, device):
 *   [0] frames  [1] frames with converged == 0 (the reference's frame errors)
 *   [2] frames that converged to a word other than ref  [3] codeword-bit errors (all n bits)
 *   [4] sum of iters.
 * ref_dev is [frames, n] uint8 (ref_stride = n) or one shared codeword (ref_stride = 0).
 */
int32_t ldpc_count_errors(const uint8_t *hard_dev, const uint8_t *ref_dev, int64_t ref_stride,
                          const int32_t *iters_dev, const uint8_t *converged_dev, int64_t frames,
                          int32_t n, uint64_t *counters_dev, void *stream);

/*
 * The step in front of the decode path, on the device: BPSK, white Gaussian noise and the channel
 * LLRs of a batch (addNoiseBPSK, /root/reference/ldpc.py:241-266, and the LLR lines at the top of
 * the decode methods: ldpc.py:143-151, bpsk.py:159-167,271-277, channel.py:268-274,410-416).
 *   x = 2b - 1 if plus_is_bit (ldpc.py:257) else 1 - 2b;  y = x + sqrt(sigma2) * N(0,1);
 *   llr = gain * y, except llr = prefix_llr at positions < prefix_len (bpsk.py:163, channel.py:412).
 * ref_dev: [frames, n] uint8 code bits (ref_stride = n), one shared codeword (ref_stride = 0) or
 * NULL for the all-zero codeword.  Noise: Philox4x32-10 keyed by `seed`, counter = (first_frame +
 * frame index, position / 4), Box-Muller in float: a frame's noise depends on its absolute index
 * only, so ranks can shard a sweep by frame ranges.  The reference's own noise stream (numpy's
 * global generator) cannot be reproduced; nothing pins these values but this build's oracle.
 */
int32_t ldpc_channel_llr(const uint8_t *ref_dev, int64_t ref_stride, int64_t frames, int32_t n,
                         double sigma2, double gain, int32_t plus_is_bit, int32_t prefix_len,
                         double prefix_llr, uint64_t seed, uint64_t first_frame, void *llr_dev,
                         int32_t llr_dtype, void *stream);

/*
 * The one collective of the multi-GPU path: sum `count` 64-bit counters (ldpc_count_errors /
 * ldpc_sweep_batch layout) over the ranks of an RCCL communicator, in place, enqueued on `stream`.
 * The reference is a single process (its FER loop adds into Python integers, ldpc.py:660-667); with
 * frames sharded over GPUs each rank counts its own frames and the sums meet here.  rccl_comm is the
 * caller's ncclComm_t; the library looks RCCL up at run time (the copy already loaded in the
 * process, else librccl.so.1) and does not link it.  rccl_comm = NULL: single rank, no-op.
 */
int32_t ldpc_allreduce_counters(void *rccl_comm, uint64_t *counters_dev, int32_t count, void *stream);

/*
 * ldpc_channel_llr under a Doppler-shifted carrier.  The reference holds no such signal path, only
 * the time-scale factor 1 - v/c of a receiver moving at speed v through a medium of wave speed c
 * (/root/reference/plotDoppler.py:5-6,14-16); the channel is this build's definition: the receiver
 * follows the symbol timing, keeps the carrier phase acquired at the start of the frame, and
 * integrates each symbol, so that
 *   y_i = x_i * sinc(cfo) * cos(2 pi (cfo * (i + 1/2) + phase0)) + sqrt(sigma2) * N(0,1)
 * with cfo = residual carrier offset in cycles per symbol (|cfo| < 1/2), e.g. (v/c) * carrier
 * frequency / symbol rate * (1 - fraction compensated), and phase0 in cycles.  LLRs are formed as in
 * ldpc_channel_llr (the receiver does not know about the rotation).  cfo = phase0 = 0 gives
 * ldpc_channel_llr exactly.  Nothing pins these values but this build's oracle.
 */
int32_t ldpc_channel_llr_doppler(const uint8_t *ref_dev, int64_t ref_stride, int64_t frames, int32_t n,
                                 double sigma2, double gain, int32_t plus_is_bit, int32_t prefix_len,
                                 double prefix_llr, uint64_t seed, uint64_t first_frame,
                                 double cfo_cycles_per_symbol, double phase0_cycles, void *llr_dev,
                                 int32_t llr_dtype, void *stream);

/*
 * Encoder for circulant codes, on the device: replaces `numpy.dot(self.G, message) % 2`
 * (/root/reference/ldpc.py:46) and the dense generator of pyldpc.coding_matrix (ldpc.py:369) for
 * codes whose generator would be too large to hold (5G NR) or is itself block-circulant (CCSDS).
 * The code is handed over as a schedule over block columns (Z-vectors of bits): information block
 * k of the message (bits [k Z, (k+1) Z)) is block column info_cols[k] of the codeword; step s, in
 * order, produces block column step_out[s] as the XOR over its terms t in
 * [step_ptr[s], step_ptr[s+1]) of block column term_col[t] rotated by term_shift[t]:
 *     out[z] ^= col[(z + shift) mod Z].
 * Block columns [0, nb) form the codeword, [nb, n_cols) are temporaries.  step_level is
 * non-decreasing; a step may only read information columns and outputs of steps of LOWER levels
 * (checked).  ldpc_amd/encoder.py derives schedules from a parity-check matrix with 5G NR / 802.11n
 * parity structure or from a block-circulant systematic generator.
 * ldpc_encode_batch: msg_dev [frames, k_cols * Z] bytes (bit 0 of each), cw_dev [frames, nb * Z]
 * bytes of 0 / 1; enqueued on `stream`.  device: HIP device that holds the tables and runs the
 * launches (the buffers must be its own); negative = the calling thread's current device.
 */
typedef struct ldpc_encoder ldpc_encoder;
int32_t ldpc_encoder_create(int32_t Z, int32_t n_cols, int32_t nb, int32_t k_cols, const int32_t *info_cols,
                            int32_t n_steps, const int32_t *step_out, const int32_t *step_ptr,
                            const int32_t *step_level, const int32_t *term_col, const int32_t *term_shift,
                            int32_t device, ldpc_encoder **out);
void ldpc_encoder_destroy(ldpc_encoder *enc);
int32_t ldpc_encode_batch(ldpc_encoder *enc, const uint8_t *msg_dev, int64_t frames, uint8_t *cw_dev, void *stream);
/*
 * The same with random messages drawn on the device, the counterpart of
 * `numpy.random.randint(0, 2, k)` per frame in the reference's loop (/root/reference/ldpc.py:649):
 * message bit i of frame f is bit (i mod 128) of Philox4x32-10 with counter (first_frame + f, i / 128, 1)
 * and key `seed` - a function of (seed, absolute frame index, bit) only, so a sweep sends the same
 * messages however it is cut into launches, ranks or GPUs.  msg_out_dev: [frames, k] bytes, or NULL.
 */
int32_t ldpc_encode_random(ldpc_encoder *enc, uint64_t seed, uint64_t first_frame, int64_t frames, uint8_t *cw_dev,
                           uint8_t *msg_out_dev, void *stream);

/*
 * 5G NR rate matching (3GPP TS 38.212 section 5.4.2: bit selection from the circular buffer, bit
 * interleaving) and its inverse on LLRs.  The reference fakes both ends with constants: the matrix
 * cut to its first 1400 rows (/root/reference/channel.py:35: the extension parity bits that were
 * never sent) and the first 80 LLRs overwritten (bpsk.py:163, channel.py:412: the 2 Z punctured
 * systematic bits of a Z = 40 code).  Codeword position offset + d is position d of the circular
 * buffer (0 <= d < n_cb, offset = 2 Z); buffer positions [filler_begin, filler_end) are filler
 * bits (known zeros, never sent); e_k is the k-th position met walking from k0, wrapping, skipping
 * fillers (k < E); f_{i + j Qm} = e_{i E/Qm + j} goes over the air (Qm = 1: no interleaving).
 * ldpc_amd/nr_rate_match.py derives the numbers (base graph and lifting size selection, k0 per
 * redundancy version, filler count, active rows of the matrix).
 *   ldpc_rate_match    cw_dev [frames, cw_stride] codeword bytes -> out_dev [frames, E] bytes (order f)
 *   ldpc_rate_recover  llr_dev [frames, E] (order f) -> out_dev [frames, n_out] LLRs of codeword
 *                      positions 0 .. n_out-1: 0 where nothing was sent, the sum over all laps where a
 *                      position was sent more than once, filler_llr at fillers; accumulate != 0 adds to
 *                      what out_dev holds (soft combining of retransmissions; fillers are set)
 */
typedef struct ldpc_rate_match_config {
    int32_t n_cb;          /* circular buffer length */
    int32_t k0;            /* starting position of the redundancy version */
    int32_t filler_begin;  /* in buffer positions */
    int32_t filler_end;
    int32_t E;             /* transmitted bits per frame */
    int32_t Qm;            /* bits per modulation symbol; 1 = no interleaving */
    int32_t offset;        /* codeword position of buffer position 0 (2 Z) */
    int32_t n_out;         /* ldpc_rate_recover: codeword positions the decoder uses */
} ldpc_rate_match_config;
int32_t ldpc_rate_match(const ldpc_rate_match_config *cfg, const uint8_t *cw_dev, int64_t cw_stride, int64_t frames,
                        uint8_t *out_dev, void *stream);
int32_t ldpc_rate_recover(const ldpc_rate_match_config *cfg, const void *llr_dev, int32_t llr_dtype, int64_t frames,
                          double filler_llr, int32_t accumulate, void *out_dev, int32_t out_dtype, void *stream);

/*
 * Channel -> decode -> error accounting of `frames` frames in ONE launch: the body of the
 * reference's Monte-Carlo loop (/root/reference/ldpc.py:644-667: encode + noise, ldpc.py:651 -> 52,
 * decode, ldpc.py:654, count, ldpc.py:660-662) without LLRs or hard decisions ever reaching device
 * memory.  The channel is that of ldpc_channel_llr / ldpc_channel_llr_doppler with the same noise
 * stream (frame f gets the noise of absolute frame first_frame + f), the counters are those of
 * ldpc_count_errors: the five numbers added to counters_dev equal, bit for bit, what the three
 * calls in a row add for the same arguments.  ref_dev: transmitted codewords [frames, n]
 * (ref_stride = n), one codeword for all frames (ref_stride = 0) or NULL (the all-zero codeword);
 * 4-byte aligned.  Served for the compiled-in codes (rules R0, R1, OMS, LNMS and the sum-product rules
 * R2, R3, R4 with their posterior clip in LDPC_F32 / LDPC_F64; rule R1 on the small codes and LNMS in
 * LDPC_F16); LDPC_ERR_UNSUPPORTED otherwise (matrices that are not compiled in, loop-bound quirks, a
 * posterior clip under a min-sum rule), and the three calls are the fallback.
 */
typedef struct ldpc_channel_config {
    double sigma2;                 /* noise variance, 1 / (2 * 10^(SNR/10)), ldpc.py:245 */
    double gain;                   /* llr = gain * y: 2 / sigma2 (ldpc.py:144) or 1 / sigma2 (bpsk.py:273) */
    double prefix_llr;             /* value of the overwritten prefix (bpsk.py:163, channel.py:412) */
    double cfo_cycles_per_symbol;  /* Doppler channel, see ldpc_channel_llr_doppler; 0 = none */
    double phase0_cycles;
    uint64_t seed;
    uint64_t first_frame;
    int32_t plus_is_bit;           /* bit sent as +1: 1 (ldpc.py:257) or 0 (bpsk.py:240-243) */
    int32_t prefix_len;
} ldpc_channel_config;
int32_t ldpc_sweep_batch(ldpc_code *code, const ldpc_decode_config *cfg, const ldpc_channel_config *channel,
                         const uint8_t *ref_dev, int64_t ref_stride, int64_t frames, int32_t compute_dtype,
                         uint64_t *counters_dev, void *stream);
/*
 * The same with the transmitted codewords as bits: position i of a frame is bit (i & 7) of byte i >> 3 of
 * its word, words ref_stride_bytes apart (>= (n + 7) / 8; 0 = one word for all frames; NULL = all zero),
 * no alignment requirement: n / 8 bytes of device memory traffic per frame instead of n.  Counters are
 * identical to ldpc_sweep_batch on the same words.  ldpc_pack_bits turns words of one byte per bit (what
 * ldpc_encode_batch writes) into this form; the low bit of each byte counts.  Neither has a counterpart
 * in the reference, whose codewords are numpy integer arrays (ldpc.py:46).
 */
int32_t ldpc_sweep_batch_packed(ldpc_code *code, const ldpc_decode_config *cfg, const ldpc_channel_config *channel,
                                const uint8_t *ref_bits_dev, int64_t ref_stride_bytes, int64_t frames,
                                int32_t compute_dtype, uint64_t *counters_dev, void *stream);
int32_t ldpc_pack_bits(const uint8_t *bytes_dev, int64_t stride, int64_t frames, int32_t n, uint8_t *bits_dev,
                       int64_t bits_stride, void *stream);

/*
 * One batch that mixes frames of several codes (e.g. CCSDS and 5G frames of one link), decoded by
 * one call: the batch is handed over as one segment per code, each with its own handle, buffers
 * and stream; every segment is enqueued on its stream, so segments on different streams run side
 * by side on the device.  The rule, arithmetic and kernel selector are shared.  No counterpart in
 * the reference (it decodes one frame of one code per call, /root/reference/bpsk.py:824,
 * channel.py:845).  Error statistics of all segments can be summed into one counter buffer by
 * calling ldpc_count_errors per segment with the same counters_dev (it adds atomically).  A segment
 * that is refused ends the call with its error; the segments before it stay enqueued.
 */
typedef struct ldpc_segment {
    ldpc_code *code;
    const void *llr_dev;   /* [frames, n of this code] */
    int64_t frames;
    uint8_t *hard_dev;
    int32_t *iters_dev;
    uint8_t *converged_dev;
    void *posterior_dev;   /* or NULL */
    void *stream;          /* hipStream_t, NULL = default stream */
} ldpc_segment;
int32_t ldpc_decode_segments(const ldpc_segment *segs, int32_t n_segs, const ldpc_decode_config *cfg,
                             int32_t llr_dtype, int32_t compute_dtype, int32_t kernel);

/* name of the kernel variant the last ldpc_decode_batch on this thread launched (for tests) */
const char *ldpc_last_kernel(void);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_AMD_H */
