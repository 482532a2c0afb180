"""``LDPCEncoder``-shaped front end: the reference's Python call surface over the HIP decoders.

The reference has one copy of ``class LDPCEncoder`` per script, with drifting APIs.  The three
classes below mirror the decode-path surface of each copy (same method names, argument meaning,
attributes set and return conventions), so a caller written against /root/reference/ldpc.py,
bpsk.py or channel.py can switch by changing its import:

    LdpcPyEncoder     ldpc.py     minSumDecode(codeword, M)  -> rule R0        ldpc.py:137-227
    BpskPyEncoder     bpsk.py     minSumDecode(codeword)     -> rule R1        bpsk.py:154-260
                                  sumProductDecodeTest(..)   -> rule R2        bpsk.py:263-387
    ChannelPyEncoder  channel.py  minSumDecode               -> rule R1        channel.py:151-257
                                  sumProductDecodeTest       -> rule R3        channel.py:402-513
                                  sumProductDecode           -> rule R4        channel.py:260-398

Conventions kept: decode methods take channel SAMPLES (the LLR formula of each method is part of
the path), read ``self.SNR`` and ``self.originalEncoded``, return the bit error rate over all n
code bits or ``FRAME_ERROR`` (None) when the iteration budget runs out, and set
``numIterations``, ``messageDecoded`` (hard decisions on success, posteriors on failure) and
``BER``.  Nothing is printed and no figure is opened.  All arithmetic runs on the GPU in float64;
there is no CPU fallback.

Not reproduced: QAM modulation / soft demodulation (needs commpy), DSSS spreading, plotting, the
``gMatrix.txt`` side effect of ``readMatrix``; random code construction uses this package's own
Gallager construction instead of ``pyldpc.make_ldpc``.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np

from . import channel, codes, config
from .decoder import Decoder

FRAME_ERROR = None


def make_regular_ldpc(n: int, d_v: int, d_c: int, seed: int = 20) -> np.ndarray:
    """Gallager's regular (d_v, d_c) construction: a band of n/d_c rows, then d_v - 1 column
    permutations of it.  Stands in for ``pyldpc.make_ldpc`` (ldpc.py:27)."""
    if n % d_c:
        raise ValueError("d_c must divide n")
    rows = n // d_c
    band = np.zeros((rows, n), np.uint8)
    for i in range(rows):
        band[i, i * d_c:(i + 1) * d_c] = 1
    rng = np.random.default_rng(seed)
    return np.concatenate([band] + [band[:, rng.permutation(n)] for _ in range(d_v - 1)], axis=0)


class _EncoderBase:
    """Shared state and helpers; subclasses map method names to rules."""

    #: True: ``self.m = n * (dV / dC)`` even for a matrix read from file (ldpc.py:30); the check-node
    #: loop then covers only the first int(self.m) rows while the syndrome uses all of them.
    M_FROM_DEGREES = False

    def __init__(self, dV, dC, n, seed=20, readDataMatrix=False, H=None, matrix_path=None,
                 compat_loop_bounds=False):
        self.dV = self.d_v = dV
        self.dC = self.d_c = dC
        self.n = n
        self.seed = seed
        self.PN = None
        if H is not None:
            Hm = np.asarray(H)
        elif matrix_path is not None:
            Hm = codes.read_parity_matrix(matrix_path)
        elif readDataMatrix:
            Hm = self._default_matrix()
        else:
            Hm = make_regular_ldpc(n, dV, dC, seed)
        self.code = codes.compile_code(Hm, name=type(self).__name__)
        self.H = self.code.dense().astype(int)
        self._G = None
        from_file = readDataMatrix or H is not None or matrix_path is not None
        # ldpc.py:30 always derives m from the degrees; bpsk.py:43 / channel.py:40 take H.shape[0]
        self.m = n * (dV / dC) if (self.M_FROM_DEGREES or not from_file) else self.H.shape[0]
        self.compat_loop_bounds = compat_loop_bounds
        self.y = None
        self.originalEncoded = None
        self.SNR = None
        self.bitEnergyRatio = None
        self.numIterations = None
        self.messageDecoded = None
        self.spread = None
        self.BER = 0
        self._decoder: Optional[Decoder] = None

    # ------------------------------------------------------------------ matrices
    def _default_matrix(self) -> np.ndarray:
        raise NotImplementedError

    @property
    def G(self) -> np.ndarray:
        """n x k generator, used as ``G @ message % 2`` like the reference's (ldpc.py:46)."""
        if self._G is None:
            Gk, _ = codes.gf2_generator(self.H)
            self._G = Gk.T.astype(int)
        return self._G

    @property
    def decoder(self) -> Decoder:
        if self._decoder is None:
            self._decoder = Decoder(self.code)
        return self._decoder

    # ------------------------------------------------------------------ encode + channel
    def encode(self, message, snr, M=2):
        """ldpc.py:42-53 / bpsk.py:84-94 (BPSK branch): codeword = G @ message % 2, returns noisy
        BPSK samples with bit 0 sent as +1 and noise variance 10^(-snr/10) (the convention of the
        third-party ``pyldpc.encode`` the reference calls; not pinned by anything in the reference)."""
        message = np.asarray(message)
        if len(message) != self.G.shape[1]:
            print("Invalid Message Length: G is " + str(self.G.shape[1]) + " message is " + str(len(message)))
            return None
        if M != 2:
            raise NotImplementedError("M-QAM modulation is outside the decode path (needs commpy)")
        self.originalEncoded = np.dot(self.G, message) % 2
        self.SNR = snr
        sigma = 10 ** (-snr / 20)
        return (-1.0) ** self.originalEncoded + sigma * np.random.randn(self.H.shape[1])

    def addNoiseBPSK(self, SNR_DB, encoded, plot=False):
        """ldpc.py:241-266: x = 2b - 1 plus noise of std sqrt(1 / (2 * 10^(SNR/10))), drawn from
        numpy's global generator exactly like the reference."""
        encoded = np.asarray(encoded)
        power = float(np.sum(encoded.astype(float) ** 2) / encoded.size)
        noise = np.sqrt(1 / (10 ** (SNR_DB / 10) * 2)) * np.random.randn(*encoded.shape)
        y = 2 * encoded - 1 + noise
        self.SNR = SNR_DB
        self.y = y
        self.bitEnergyRatio = power
        self.symbolNoiseRatio = power
        return y

    def isValidCodeword(self, decoded_codeword):
        """ldpc.py:76-80."""
        syndrome = np.dot(self.H, np.asarray(decoded_codeword).T) % 2
        return bool(np.all(syndrome == 0))

    def raw(self, codeword):
        """Uncoded bit error rate of the hard-sliced samples (ldpc.py:293-301)."""
        hard = np.where(np.asarray(codeword) < 0, 0, 1)
        return np.sum(np.array(self.originalEncoded) != hard) / len(codeword)

    # ------------------------------------------------------------------ decode
    def _config(self, base: config.DecodeConfig) -> config.DecodeConfig:
        if not self.compat_loop_bounds:
            return base
        if base.rule in (config.RULE_R0, config.RULE_BF):
            return base.with_(num_active_checks=min(int(self.m), self.H.shape[0]))
        if self.n > self.H.shape[1]:
            raise IndexError("self.n exceeds the matrix width: the reference indexes H out of bounds here")
        return base.with_(num_refresh_vars=int(self.n))

    def _decode(self, codeword, base: config.DecodeConfig):
        cfg = self._config(base)
        y = np.asarray(codeword, dtype=np.float64)
        if y.shape != (self.H.shape[1],):
            raise ValueError(f"codeword must have {self.H.shape[1]} samples")
        llr = channel.llr_from_samples(y, self.SNR, cfg)
        res = self.decoder.decode(llr, cfg, compute_dtype=np.float64, want_posterior=True)
        hard = res.hard[0]
        self.numIterations = int(res.iters[0])
        errors = int(np.sum(np.array(self.originalEncoded) != hard))
        ber = errors / y.shape[0]
        if res.converged[0]:
            self.messageDecoded = hard.astype(float) if cfg.rule == config.RULE_R0 else hard.tolist()
            return ber
        self.messageDecoded = res.posterior[0]
        self.BER = ber
        return FRAME_ERROR

    def decode_batch(self, samples, base: config.DecodeConfig, compute_dtype=np.float64):
        """Many frames of channel samples at once (same SNR): the batched form of the methods."""
        cfg = self._config(base)
        llr = channel.llr_from_samples(np.asarray(samples, dtype=np.float64), self.SNR, cfg)
        return self.decoder.decode(llr, cfg, compute_dtype=compute_dtype, want_posterior=True)


class LdpcPyEncoder(_EncoderBase):
    """Surface of /root/reference/ldpc.py's class (ldpc.py:17-356)."""

    M_FROM_DEGREES = True

    def _default_matrix(self):
        # the reference opens ./parityMatrix.txt (ldpc.py:25); the same matrix ships with this package
        if os.path.exists("parityMatrix.txt"):
            return codes.read_parity_matrix("parityMatrix.txt")
        return codes.load_packaged("wifi648").dense()

    def bitFlipDecode(self, codeword):
        """ldpc.py:82-135: hard-decision bit flipping.  Returns the bit error rate over all n bits
        whether or not a codeword was reached, and sets no attribute, as the reference does."""
        cfg = self._config(config.BF)
        y = np.asarray(codeword, dtype=np.float64)[None, :]
        res = self.decoder.decode(y, cfg)
        return float(np.sum(np.asarray(self.originalEncoded) != res.hard[0]) / self.H.shape[1])

    def minSumDecode(self, codeword, M=2):
        if M != 2:
            raise NotImplementedError("M-QAM soft demodulation is outside the decode path (needs commpy)")
        return self._decode(codeword, config.R0)


class BpskPyEncoder(_EncoderBase):
    """Surface of /root/reference/bpsk.py's class (bpsk.py:28-570)."""

    def _default_matrix(self):
        for p in ("Matrices/parityMatrix.txt", "parityMatrix.txt"):
            if os.path.exists(p):
                return codes.read_parity_matrix(p)
        return codes.load_packaged("wifi648").dense()

    def minSumDecode(self, codeword):
        return self._decode(codeword, config.R1)

    def sumProductDecodeTest(self, codeword):
        return self._decode(codeword, config.R2)


class ChannelPyEncoder(_EncoderBase):
    """Surface of /root/reference/channel.py's class (channel.py:25-607)."""

    def _default_matrix(self):
        # channel.py:34-35: the 5G matrix with its first 1400 rows kept
        if os.path.exists("5GMatrix.mat"):
            return codes.read_mat_matrix("5GMatrix.mat")[:1400]
        return codes.load_packaged("nr_bg2_z40_r1400").dense()

    def minSumDecode(self, codeword):
        return self._decode(codeword, config.R1)

    def sumProductDecodeTest(self, codeword):
        return self._decode(codeword, config.R3)

    def sumProductDecode(self, codeword):
        return self._decode(codeword, config.R4)


#: the name the reference's callers import
LDPCEncoder = LdpcPyEncoder
