"""Monte-Carlo harness around the batched decoder: synthetic frames on the device, decode, error
counters, and the one collective of the multi-GPU path.

Stands in for the reference's FER loop (/root/reference/ldpc.py:637-667, bpsk.py:807-836), which
draws one random message at a time, encodes it, adds noise, decodes it and counts a frame error
whenever the decoder returns None.  Here a whole batch lives in HBM and each rank owns its own
frames; ranks only meet in ``allreduce_counters``.

torch is used for device memory, random numbers and torch.distributed only.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .codes import CompiledCode
from .config import RULE_R1, DecodeConfig
from .decoder import Decoder, _dtype_code
from ._lib import LDPC_ERR_UNSUPPORTED, LdpcError

COUNTER_NAMES = ("frames", "not_converged", "converged_wrong", "bit_errors", "iterations")


@dataclass
class DeviceBatch:
    llr: "torch.Tensor"        # [B, n] float32 or float64
    ref: "torch.Tensor"        # [B, n] uint8 transmitted codewords
    hard: "torch.Tensor"       # [B, n] uint8
    iters: "torch.Tensor"      # [B] int32
    converged: "torch.Tensor"  # [B] uint8
    counters: "torch.Tensor"   # [5] int64 (the library adds into it as uint64)


def make_codewords(code: CompiledCode, G, frames: int, seed, device) -> "torch.Tensor":
    """[frames, n] uint8 codewords of random messages on the device: all-zero if ``G`` is None, else
    ``G`` is a k x n generator matrix (msg @ G mod 2, the reference's ldpc.py:46 with G transposed)
    or a callable encoder with attribute ``k``: ``encoder.DeviceEncoder`` (ldpc_encode_batch, messages
    drawn on the device) or a host function on numpy arrays."""
    import torch

    n = code.n
    if G is None:
        return torch.zeros((frames, n), dtype=torch.uint8, device=device)
    cw = torch.empty((frames, n), dtype=torch.uint8, device=device)
    if callable(G) and getattr(G, "on_device", False):  # device encoder (encoder.DeviceEncoder): messages drawn on the device
        gen = torch.Generator(device=device)
        gen.manual_seed(int(np.random.SeedSequence(seed).generate_state(1)[0]))
        step = max(1, (1 << 28) // max(n, 1))
        for a in range(0, frames, step):
            b = min(frames, a + step)
            cw[a:b] = G(torch.randint(0, 2, (b - a, G.k), generator=gen, device=device, dtype=torch.uint8))
        return cw
    if callable(G):  # structured encoder on the host
        host_rng = np.random.default_rng(np.random.SeedSequence(seed).generate_state(2))
        step = max(1, (1 << 26) // max(n, 1))
        for a in range(0, frames, step):
            b = min(frames, a + step)
            msg = host_rng.integers(0, 2, size=(b - a, G.k), dtype=np.uint8)
            cw[a:b] = torch.as_tensor(G(msg), device=device)
        return cw
    gen = torch.Generator(device=device)
    gen.manual_seed(int(np.random.SeedSequence(seed).generate_state(1)[0]))
    Gt = torch.as_tensor(np.ascontiguousarray(G), dtype=torch.float32, device=device)
    step = max(1, (1 << 28) // max(n, 1))
    for a in range(0, frames, step):
        b = min(frames, a + step)
        msg = torch.randint(0, 2, (b - a, Gt.shape[0]), generator=gen, device=device, dtype=torch.int32)
        cw[a:b] = torch.remainder(msg.to(torch.float32) @ Gt, 2.0).to(torch.uint8)
    return cw


def _empty_outputs(cw, llr, device):
    import torch

    frames, n = cw.shape
    return DeviceBatch(llr, cw,
                       torch.empty((frames, n), dtype=torch.uint8, device=device),
                       torch.empty((frames,), dtype=torch.int32, device=device),
                       torch.empty((frames,), dtype=torch.uint8, device=device),
                       torch.zeros((5,), dtype=torch.int64, device=device))


def noisy_batch(code: CompiledCode, cw, snr_db: float, cfg: DecodeConfig, seed, device, llr_dtype=None,
                llr_scale_snr_db: Optional[float] = None, first_frame: int = 0, cfo: float = 0.0,
                phase0: float = 0.0, compat_prefix: bool = False) -> DeviceBatch:
    """Channel LLRs for the codewords ``cw`` from the library's own kernel (ldpc_channel_llr): frame
    f gets the noise of absolute frame ``first_frame + f``, whatever batch or rank generates it.
    ``cfo`` / ``phase0``: the Doppler-shifted channel (ldpc_channel_llr_doppler).  ``compat_prefix``:
    the first 80 LLRs are overwritten as the reference's R1 / R3 / R4 methods do (bpsk.py:163,
    channel.py:412)."""
    import torch

    from . import channel
    llr_dtype = llr_dtype or torch.float32
    frames, n = cw.shape
    llr = torch.empty((frames, n), dtype=llr_dtype, device=device)
    key = int(np.random.SeedSequence(seed).generate_state(1, dtype=np.uint64)[0])
    if frames:
        channel.channel_llr_device(cw.data_ptr(), n, frames, n, snr_db, cfg, key, first_frame, llr.data_ptr(),
                                   np.float32 if llr_dtype == torch.float32 else np.float64,
                                   stream=torch.cuda.current_stream().cuda_stream,
                                   llr_scale_snr_db=llr_scale_snr_db, cfo=cfo, phase0=phase0,
                                   compat_prefix=compat_prefix)
    return _empty_outputs(cw, llr, device)


def make_batch(code: CompiledCode, G: Optional[np.ndarray], frames: int, snr_db: float,
               cfg: DecodeConfig, seed, device, llr_dtype=None, llr_scale_snr_db: Optional[float] = None,
               native_noise: bool = False, first_frame: int = 0) -> DeviceBatch:
    """Random codewords (``make_codewords``), BPSK with the polarity ``cfg`` expects, AWGN
    of variance 1/(2*10^(snr/10)) (ldpc.py:245), LLR = 2y/sigma^2 (y/sigma^2 for R2).  Noise comes
    from the library's channel kernel (``native_noise``) or from torch's generator."""
    import torch

    llr_dtype = llr_dtype or torch.float32
    n = code.n
    cw = make_codewords(code, G, frames, seed, device)
    if native_noise:
        return noisy_batch(code, cw, snr_db, cfg, seed, device, llr_dtype=llr_dtype,
                           llr_scale_snr_db=llr_scale_snr_db, first_frame=first_frame)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(np.random.SeedSequence(list(np.atleast_1d(seed)) + [1]).generate_state(1)[0]))
    sigma2 = 1.0 / (2.0 * 10.0 ** (snr_db / 10.0))
    sigma2_llr = sigma2 if llr_scale_snr_db is None else 1.0 / (2.0 * 10.0 ** (llr_scale_snr_db / 10.0))
    gain = (1.0 if cfg.rule == 2 else 2.0) / sigma2_llr
    llr = torch.empty((frames, n), dtype=llr_dtype, device=device)
    step = max(1, (1 << 27) // max(n, 1))
    for a in range(0, frames, step):
        b = min(frames, a + step)
        c = cw[a:b].to(torch.float64)
        x = 2.0 * c - 1.0 if cfg.plus_is_bit == 1 else 1.0 - 2.0 * c
        noise = torch.randn((b - a, n), generator=gen, device=device, dtype=torch.float64)
        llr[a:b] = ((x + (sigma2 ** 0.5) * noise) * gain).to(llr_dtype)
    return _empty_outputs(cw, llr, device)


def decode_batch(dec: Decoder, batch: DeviceBatch, cfg: DecodeConfig, compute_dtype=np.float64,
                 stream: Optional[int] = None, kernel: int = 0) -> str:
    """Enqueue one decode of the whole batch on ``stream`` (default: torch's current stream)."""
    import torch

    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    np_dtype = np.float32 if batch.llr.dtype == torch.float32 else np.float64
    return dec.decode_device(batch.llr.data_ptr(), np_dtype, batch.llr.shape[0], batch.hard.data_ptr(),
                             batch.iters.data_ptr(), batch.converged.data_ptr(), cfg,
                             compute_dtype=compute_dtype, stream=stream, kernel=kernel)


_side_streams = {}


def decode_mixed(parts, cfg: DecodeConfig, compute_dtype=np.float64, kernel: int = 0, counters=None):
    """Decode one batch that mixes frames of several codes: ``parts`` is a list of
    (Decoder, DeviceBatch), one per code; a single ``ldpc_decode_segments`` call enqueues every part
    on a stream of its own, so the codes are decoded side by side.  With ``counters`` (an int64[5]
    device tensor) the error statistics of all parts are added into it.  Returns the name of the kernel
    the last part ran on."""
    import ctypes

    import torch

    from . import _lib
    from .config import CDecodeConfig

    lib = _lib.load()
    cur = torch.cuda.current_stream()
    dev = parts[0][1].llr.device
    streams = _side_streams.setdefault(dev, [])
    while len(streams) < len(parts):
        streams.append(torch.cuda.Stream(device=dev))
    segs = (_lib.CSegment * len(parts))()
    llr_dtype = parts[0][1].llr.dtype
    for i, (dec, bt) in enumerate(parts):
        if bt.llr.dtype != llr_dtype:
            raise TypeError("all parts of a mixed batch share one LLR type")
        streams[i].wait_stream(cur)          # the LLRs were produced on the current stream
        segs[i] = _lib.CSegment(dec._h.value, bt.llr.data_ptr(), bt.llr.shape[0], bt.hard.data_ptr(), bt.iters.data_ptr(),
                                bt.converged.data_ptr(), None, streams[i].cuda_stream)
    ccfg = CDecodeConfig.from_config(cfg)
    _lib.check(lib.ldpc_decode_segments(segs, len(parts), ctypes.byref(ccfg), 0 if llr_dtype == torch.float32 else 1,
                                        _dtype_code(compute_dtype), kernel))
    for i, (dec, bt) in enumerate(parts):
        if counters is not None:
            dec.count_errors_device(bt.hard.data_ptr(), bt.ref.data_ptr(), dec.code.n, bt.iters.data_ptr(),
                                    bt.converged.data_ptr(), bt.llr.shape[0], counters.data_ptr(),
                                    stream=streams[i].cuda_stream)
        cur.wait_stream(streams[i])
    return _lib.last_kernel()


def count_errors(dec: Decoder, batch: DeviceBatch, stream: Optional[int] = None) -> None:
    """Add this batch's error statistics into ``batch.counters`` (ldpc.py:166-170, 210-216, 660-667)."""
    import torch

    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    dec.count_errors_device(batch.hard.data_ptr(), batch.ref.data_ptr(), dec.code.n, batch.iters.data_ptr(),
                            batch.converged.data_ptr(), batch.llr.shape[0], batch.counters.data_ptr(),
                            stream=stream)


def sweep_fused(dec: Decoder, cw, snr_db: float, cfg: DecodeConfig, seed, counters, compute_dtype=np.float64,
                first_frame: int = 0, llr_scale_snr_db: Optional[float] = None, cfo: float = 0.0, phase0: float = 0.0,
                compat_prefix: bool = False, stream: Optional[int] = None, packed_n: Optional[int] = None) -> str:
    """Channel, decode and error accounting of the codewords ``cw`` ([frames, n] uint8 on the device,
    as ``make_codewords`` gives them) in one launch
    (``ldpc_sweep_batch``): adds to ``counters`` exactly what ``noisy_batch`` + ``decode_batch`` +
    ``count_errors`` add for the same arguments, without LLRs or hard decisions in device memory.
    ``packed_n``: ``cw`` holds bits instead ([frames, bytes per word] from ``pack_bits``, words of
    ``packed_n`` positions; ``ldpc_sweep_batch_packed``).
    Raises ``LdpcError`` (LDPC_ERR_UNSUPPORTED) when no fused kernel serves the code and rule."""
    import ctypes

    import torch

    from . import _lib, channel
    from .config import CDecodeConfig

    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    sigma2, gain, plen, pval = channel.channel_parameters(snr_db, cfg, compat_prefix, llr_scale_snr_db)
    key = int(np.random.SeedSequence(seed).generate_state(1, dtype=np.uint64)[0])
    ch = _lib.CChannelConfig(sigma2, gain, pval, float(cfo), float(phase0), key, int(first_frame),
                             int(cfg.plus_is_bit == 1), plen)
    ccfg = CDecodeConfig.from_config(cfg)
    frames, n = cw.shape
    if packed_n is not None:
        if packed_n != dec.code.n or n < (packed_n + 7) // 8:
            raise ValueError("packed words do not match the code length")
        call = _lib.load().ldpc_sweep_batch_packed
    else:
        call = _lib.load().ldpc_sweep_batch
    _lib.check(call(dec._h, ctypes.byref(ccfg), ctypes.byref(ch), cw.data_ptr(), n, frames,
                    _dtype_code(compute_dtype), counters.data_ptr(), stream or None))
    return _lib.last_kernel()


def pack_bits(cw, stream: Optional[int] = None):
    """Codewords of one byte per bit ([frames, n] uint8 on the device) as bits, eight positions per byte,
    lowest bit first ([frames, (n + 7) // 8] uint8): the reference words of ``sweep_fused(packed_n=n)``."""
    import torch

    from . import _lib

    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    frames, n = cw.shape
    out = torch.empty((frames, (n + 7) // 8), dtype=torch.uint8, device=cw.device)
    _lib.check(_lib.load().ldpc_pack_bits(cw.data_ptr(), n, frames, n, out.data_ptr(), out.shape[1], stream or None))
    return out


def allreduce_counters(counters: "torch.Tensor") -> "torch.Tensor":
    """Sum the [.., 5] int64 counters over all ranks (RCCL on GPUs, gloo on CPU); the only
    collective of the multi-GPU path.  A no-op without an initialised process group."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if counters.is_cuda and dist.get_backend() != "nccl":   # host-memory backends (gloo rehearsals)
            host = counters.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            counters.copy_(host)
        else:
            dist.all_reduce(counters, op=dist.ReduceOp.SUM)
    return counters


def shard_range(total: int, rank: int, world: int):
    """Contiguous frame range [a, b) of ``rank`` when ``total`` frames are split over ``world`` ranks."""
    base, extra = divmod(total, world)
    a = rank * base + min(rank, extra)
    return a, a + base + (1 if rank < extra else 0)


def summarize(counters: np.ndarray, n: int) -> dict:
    c = {k: int(v) for k, v in zip(COUNTER_NAMES, np.asarray(counters).tolist())}
    frames = max(c["frames"], 1)
    c["fer"] = c["not_converged"] / frames            # the reference's FER (ldpc.py:667)
    c["ber"] = c["bit_errors"] / (frames * n)         # over all n code bits (ldpc.py:166-170)
    c["mean_iters"] = c["iterations"] / frames
    return c


def _rank_world(rank, world):
    """Rank and world size: the arguments when given, else torch.distributed's, else (0, 1)."""
    if rank is not None and world is not None:
        return int(rank), int(world)
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


#: frames per seeding unit of a sweep: the codewords of absolute frames [c * CHUNK, (c + 1) * CHUNK)
#: of an SNR point come from seed [seed, point, c], whichever rank decodes them
SWEEP_CHUNK = 1 << 14


def ber_sweep(code: CompiledCode, G: Optional[np.ndarray], cfg: DecodeConfig, snrs_db, frames_per_point: int,
              device, compute_dtype=np.float64, seed=0, dec: Optional[Decoder] = None, batch: int = 1 << 18,
              rank: Optional[int] = None, world: Optional[int] = None, reduce: bool = True,
              compat_prefix: bool = False, fused="auto", cfo: float = 0.0, phase0: float = 0.0):
    """Monte-Carlo error-rate sweep, the batched form of the reference's ``plotFrameError``
    (ldpc.py:637-667).  ``cfo`` / ``phase0``: the Doppler-shifted channel of ``noisy_batch``.  ``fused``: True = channel, decode and counting in one launch per batch
    (``ldpc_sweep_batch``; an error where no fused kernel exists), False = three launches, "auto" =
    fused where a kernel exists.  The counters are the same either way.  The ``frames_per_point`` frames of an SNR point are numbered 0, 1, ...; rank r
    of ``world`` decodes the contiguous range ``shard_range(frames_per_point, r, world)`` of them,
    counts on the device, and the counters are summed over the ranks (one all-reduce per point).
    A frame's codeword and noise depend on (seed, point, absolute frame number) only, so the summed
    counters are the same for every ``world``.  Returns one summary dict per SNR point."""
    import torch

    rank, world = _rank_world(rank, world)
    dec = dec or Decoder(code)
    per_launch = max(SWEEP_CHUNK, batch // SWEEP_CHUNK * SWEEP_CHUNK)
    if fused == "auto" and np.dtype(compute_dtype) == np.float16 and cfg.rule == RULE_R1:
        # the packed half-precision R1 decode of the small codes is so short that drawing the noise inside it costs
        # more than the LLR array's trip through device memory: three launches are the faster path (measured: 78
        # against 70 Gbit/s on the CCSDS code)
        fused = False
    out = []
    for si, snr in enumerate(snrs_db):
        total = torch.zeros((5,), dtype=torch.int64, device=device)
        a, b = shard_range(frames_per_point, rank, world)
        pos = a
        while pos < b:
            # one launch: the part of [pos, b) that lies in the next `per_launch`-aligned window
            end = min(b, (pos // per_launch + 1) * per_launch)
            done = False
            if fused:
                cw = codewords_range(code, G, pos, end, frames_per_point, [seed, si], device)
                try:
                    sweep_fused(dec, cw, float(snr), cfg, [seed, si], total, compute_dtype=compute_dtype,
                                first_frame=pos, compat_prefix=compat_prefix, cfo=cfo, phase0=phase0)
                    done = True
                except LdpcError as e:
                    if fused is True or e.code != LDPC_ERR_UNSUPPORTED:
                        raise
                    fused = False
            if not done:
                bt = make_batch_range(code, G, pos, end, frames_per_point, float(snr), cfg, seed=[seed, si], device=device,
                                      compat_prefix=compat_prefix, cfo=cfo, phase0=phase0)
                decode_batch(dec, bt, cfg, compute_dtype=compute_dtype)
                count_errors(dec, bt)
                total += bt.counters
            pos = end
        if reduce:
            allreduce_counters(total)
        torch.cuda.synchronize()
        summ = summarize(total.cpu().numpy(), code.n)
        summ["es_n0_db"] = float(snr)
        out.append(summ)
    return out


def codewords_range(code: CompiledCode, G, first: int, last: int, total: int, seed, device):
    """Codewords of the frames [first, last) of a sweep point of ``total`` frames, drawn per
    ``SWEEP_CHUNK``-aligned chunk of absolute frame numbers (seed + [chunk])."""
    import torch

    seed = list(np.atleast_1d(seed))
    if hasattr(G, "random_codewords"):   # device encoder: messages keyed by the absolute frame number, one launch
        return G.random_codewords(last - first, seed + [77], first_frame=first, device=device)
    parts = []
    for c in range(first // SWEEP_CHUNK, (last + SWEEP_CHUNK - 1) // SWEEP_CHUNK):
        c0 = c * SWEEP_CHUNK
        c1 = min(total, c0 + SWEEP_CHUNK)
        cw = make_codewords(code, G, c1 - c0, seed + [c], device)
        parts.append(cw[max(first, c0) - c0:min(last, c1) - c0])
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=0).contiguous()


def make_batch_range(code: CompiledCode, G, first: int, last: int, total: int, snr_db: float, cfg: DecodeConfig,
                     seed, device, llr_dtype=None, compat_prefix: bool = False, cfo: float = 0.0,
                     phase0: float = 0.0) -> DeviceBatch:
    """The frames [first, last) of a sweep point of ``total`` frames.  Codewords come from
    ``codewords_range``, noise from the library's channel kernel keyed by the absolute frame number:
    the result does not depend on how the point is cut into ranges."""
    seed = list(np.atleast_1d(seed))
    cw = codewords_range(code, G, first, last, total, seed, device)
    return noisy_batch(code, cw, snr_db, cfg, seed, device, llr_dtype=llr_dtype, first_frame=first,
                       compat_prefix=compat_prefix, cfo=cfo, phase0=phase0)
