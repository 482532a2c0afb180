"""LLR producers in front of the decode path (host side, numpy).

The reference forms channel LLRs inside each decode method from BPSK samples and ``self.SNR``;
the formulas differ per method and two opposite BPSK polarities are in use:

* ``addNoiseBPSK`` (/root/reference/ldpc.py:241-266): x = 2b - 1, sigma^2 = 1 / (2 * 10^(SNR/10)),
  i.e. bit 1 is sent as +1 — the polarity R0's hard decision assumes (ldpc.py:198-201);
* ``pyldpc.encode`` behind ``LDPCEncoder.encode`` (/root/reference/ldpc.py:52): x = (-1)^b, bit 0 is
  sent as +1 — the polarity R1-R4 assume (bpsk.py:240-243).  That package is not part of the
  reference tree; only the polarity matters here and the noise level is taken as in
  ``addNoiseBPSK``.

"SNR" is Es/N0-like everywhere in the reference: no code-rate term.
"""
from __future__ import annotations

import numpy as np

from .config import DecodeConfig, RULE_R1, RULE_R2, RULE_R3, RULE_R4

#: number of leading code bits whose LLR the reference overwrites (bpsk.py:163, channel.py:270,412)
PUNCTURE_PREFIX = 80


def noise_variance(snr_db) -> np.ndarray:
    """sigma^2 = 1 / (2 * 10^(SNR/10))  (ldpc.py:143)."""
    return 1 / (2 * (10 ** (np.asarray(snr_db, dtype=np.float64) / 10)))


def bpsk_awgn(codewords: np.ndarray, snr_db: float, plus_is_bit: int, rng: np.random.Generator) -> np.ndarray:
    """BPSK symbols plus white Gaussian noise of variance ``noise_variance(snr_db)``."""
    c = np.asarray(codewords, dtype=np.float64)
    x = 2.0 * c - 1.0 if plus_is_bit == 1 else 1.0 - 2.0 * c
    return x + np.sqrt(noise_variance(snr_db)) * rng.standard_normal(c.shape)


def llr_from_samples(y: np.ndarray, snr_db, cfg: DecodeConfig, compat_prefix: bool = True) -> np.ndarray:
    """Channel LLRs as the reference's decode method for ``cfg.rule`` forms them.

    R0: 2y/sigma^2 (ldpc.py:143-144).  R1: the same with the first 80 set to 1e-9 (bpsk.py:161-167).
    R2: y/sigma^2, no factor 2 and no overwritten prefix (bpsk.py:271-277).  R3/R4: 2y/sigma^2 with
    the first 80 set to 0.0 (channel.py:410-416, 268-274).  ``compat_prefix=False`` leaves the
    prefix alone (for codes that are not punctured).
    """
    y = np.asarray(y, dtype=np.float64)
    sigma2 = noise_variance(snr_db)
    if sigma2.ndim:
        sigma2 = sigma2[..., None]
    L = y / sigma2 if cfg.rule == RULE_R2 else (2 * y) / sigma2
    if compat_prefix:
        if cfg.rule == RULE_R1:
            L[..., :PUNCTURE_PREFIX] = 1e-9
        elif cfg.rule in (RULE_R3, RULE_R4):
            L[..., :PUNCTURE_PREFIX] = 0.0
    return L


#: medium and platform of the reference's Doppler plot (/root/reference/plotDoppler.py:5-6): sound in
#: water, 10 m/s of relative speed; its frequencies start at 12 kHz (plotDoppler.py:10)
WAVE_SPEED, RELATIVE_SPEED, CARRIER_HZ = 1500.0, 10.0, 12000.0


def doppler_scale(relative_speed: float = RELATIVE_SPEED, wave_speed: float = WAVE_SPEED) -> float:
    """Time-scale factor 1 - v/c of the received signal (plotDoppler.py:14-16)."""
    return 1.0 - relative_speed / wave_speed


def doppler_cfo(symbol_rate: float, relative_speed: float = RELATIVE_SPEED, wave_speed: float = WAVE_SPEED,
                carrier_hz: float = CARRIER_HZ, compensated: float = 0.0) -> float:
    """Residual carrier offset in cycles per symbol that ``ldpc_channel_llr_doppler`` takes: the
    carrier moves by (scale - 1) * carrier_hz; a receiver that removes the fraction ``compensated``
    of it is left with the rest, expressed per symbol period."""
    return (doppler_scale(relative_speed, wave_speed) - 1.0) * carrier_hz * (1.0 - compensated) / symbol_rate


def channel_parameters(snr_db: float, cfg: DecodeConfig, compat_prefix: bool = False, llr_scale_snr_db=None):
    """(noise variance, LLR gain, prefix length, prefix value) of the device channel for ``cfg.rule``:
    variance 1/(2*10^(SNR/10)) (ldpc.py:245), gain 2/sigma^2 (1/sigma^2 for R2, bpsk.py:273) taken at
    ``llr_scale_snr_db`` if the receiver's SNR estimate differs from the channel's, and the rule's
    overwritten prefix (bpsk.py:163, channel.py:412) when ``compat_prefix``."""
    sigma2 = float(noise_variance(snr_db))
    sigma2_llr = sigma2 if llr_scale_snr_db is None else float(noise_variance(llr_scale_snr_db))
    gain = (1.0 if cfg.rule == RULE_R2 else 2.0) / sigma2_llr
    plen, pval = 0, 0.0
    if compat_prefix and cfg.rule == RULE_R1:
        plen, pval = PUNCTURE_PREFIX, 1e-9
    elif compat_prefix and cfg.rule in (RULE_R3, RULE_R4):
        plen, pval = PUNCTURE_PREFIX, 0.0
    return sigma2, gain, plen, pval


def channel_llr_device(ref_ptr: int, ref_stride: int, frames: int, n: int, snr_db: float, cfg: DecodeConfig,
                       seed: int, first_frame: int, llr_ptr: int, llr_dtype=np.float32, stream: int = 0,
                       compat_prefix: bool = False, llr_scale_snr_db=None, cfo: float = 0.0,
                       phase0: float = 0.0) -> None:
    """The same on the device, for a whole batch (``ldpc_channel_llr``): BPSK with the polarity
    ``cfg`` assumes, noise of variance ``noise_variance(snr_db)``, LLR gain 2/sigma^2 (1/sigma^2 for
    R2) taken at ``llr_scale_snr_db`` if the receiver's SNR estimate differs from the channel's, and
    the rule's overwritten prefix when ``compat_prefix``.  ``ref_ptr`` = 0 sends the all-zero
    codeword.  Frame f of the call gets the noise of absolute frame ``first_frame + f``.  A non-zero
    ``cfo`` (cycles per symbol, see ``doppler_cfo``) or ``phase0`` (cycles) selects the Doppler-shifted
    channel ``ldpc_channel_llr_doppler``."""
    from . import _lib
    sigma2, gain, plen, pval = channel_parameters(snr_db, cfg, compat_prefix, llr_scale_snr_db)
    dt = 1 if np.dtype(llr_dtype) == np.float64 else 0
    if cfo or phase0:
        _lib.check(_lib.load().ldpc_channel_llr_doppler(
            ref_ptr or None, int(ref_stride), int(frames), int(n), sigma2, gain, int(cfg.plus_is_bit == 1), plen, pval,
            int(seed) & (2 ** 64 - 1), int(first_frame), float(cfo), float(phase0), llr_ptr, dt, stream or None))
        return
    _lib.check(_lib.load().ldpc_channel_llr(ref_ptr or None, int(ref_stride), int(frames), int(n), sigma2, gain,
                                            int(cfg.plus_is_bit == 1), plen, pval, int(seed) & (2 ** 64 - 1),
                                            int(first_frame), llr_ptr, dt, stream or None))
