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
from .config import DecodeConfig
from .decoder import Decoder

COUNTER_NAMES = ("frames", "not_converged", "converged_wrong", "bit_errors", "iterations")


@dataclass
class DeviceBatch:
    llr: "torch.Tensor"        # [B, n] float32 or float64
    ref: "torch.Tensor"        # [B, n] uint8 transmitted codewords
    hard: "torch.Tensor"       # [B, n] uint8
    iters: "torch.Tensor"      # [B] int32
    converged: "torch.Tensor"  # [B] uint8
    counters: "torch.Tensor"   # [5] int64 (the library adds into it as uint64)


def make_batch(code: CompiledCode, G: Optional[np.ndarray], frames: int, snr_db: float,
               cfg: DecodeConfig, seed, device, llr_dtype=None, llr_scale_snr_db: Optional[float] = None,
               native_noise: bool = False, first_frame: int = 0) -> DeviceBatch:
    """Random codewords (all-zero if ``G`` is None; ``G`` is a k x n generator matrix or a callable
    encoder with attribute ``k``), BPSK with the polarity ``cfg`` expects, AWGN
    of variance 1/(2*10^(snr/10)) (ldpc.py:245), LLR = 2y/sigma^2 (y/sigma^2 for R2)."""
    import torch

    llr_dtype = llr_dtype or torch.float32
    gen = torch.Generator(device=device)
    gen.manual_seed(int(np.random.SeedSequence(seed).generate_state(1)[0]))
    n = code.n
    if G is None:
        cw = torch.zeros((frames, n), dtype=torch.uint8, device=device)
    elif callable(G):  # structured encoder (host side), e.g. nr_bg1.Encoder: messages of G.k bits
        host_rng = np.random.default_rng(np.random.SeedSequence(seed).generate_state(2))
        cw = torch.empty((frames, n), dtype=torch.uint8, device=device)
        step = max(1, (1 << 26) // max(n, 1))
        for a in range(0, frames, step):
            b = min(frames, a + step)
            cw[a:b] = torch.as_tensor(G(host_rng.integers(0, 2, size=(b - a, G.k), dtype=np.uint8)), device=device)
    else:
        Gt = torch.as_tensor(np.ascontiguousarray(G), dtype=torch.float32, device=device)
        cw = torch.empty((frames, n), dtype=torch.uint8, device=device)
        step = max(1, (1 << 28) // max(n, 1))
        for a in range(0, frames, step):
            b = min(frames, a + step)
            msg = torch.randint(0, 2, (b - a, Gt.shape[0]), generator=gen, device=device, dtype=torch.int32)
            cw[a:b] = torch.remainder(msg.to(torch.float32) @ Gt, 2.0).to(torch.uint8)
    if native_noise:
        # noise and LLRs from the library's own kernel: frame f gets the noise of absolute frame
        # first_frame + f, whatever batch or rank it is generated in
        from . import channel
        llr = torch.empty((frames, n), dtype=llr_dtype, device=device)
        key = int(np.random.SeedSequence(seed).generate_state(1, dtype=np.uint64)[0])
        channel.channel_llr_device(cw.data_ptr(), n, frames, n, snr_db, cfg, key, first_frame, llr.data_ptr(),
                                   np.float32 if llr_dtype == torch.float32 else np.float64,
                                   stream=torch.cuda.current_stream().cuda_stream,
                                   llr_scale_snr_db=llr_scale_snr_db)
        return DeviceBatch(llr, cw,
                           torch.empty((frames, n), dtype=torch.uint8, device=device),
                           torch.empty((frames,), dtype=torch.int32, device=device),
                           torch.empty((frames,), dtype=torch.uint8, device=device),
                           torch.zeros((5,), dtype=torch.int64, device=device))
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
    return DeviceBatch(llr, cw,
                       torch.empty((frames, n), dtype=torch.uint8, device=device),
                       torch.empty((frames,), dtype=torch.int32, device=device),
                       torch.empty((frames,), dtype=torch.uint8, device=device),
                       torch.zeros((5,), dtype=torch.int64, device=device))


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


def count_errors(dec: Decoder, batch: DeviceBatch, stream: Optional[int] = None) -> None:
    """Add this batch's error statistics into ``batch.counters`` (ldpc.py:166-170, 210-216, 660-667)."""
    import torch

    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    dec.count_errors_device(batch.hard.data_ptr(), batch.ref.data_ptr(), dec.code.n, batch.iters.data_ptr(),
                            batch.converged.data_ptr(), batch.llr.shape[0], batch.counters.data_ptr(),
                            stream=stream)


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


def ber_sweep(code: CompiledCode, G: Optional[np.ndarray], cfg: DecodeConfig, snrs_db, frames_per_point: int,
              device, compute_dtype=np.float64, seed=0, dec: Optional[Decoder] = None, batch: int = 1 << 18):
    """Monte-Carlo error-rate sweep, the batched form of the reference's ``plotFrameError``
    (ldpc.py:637-667): per SNR point decode ``frames_per_point`` random frames on this rank, count
    on the device, sum the counters over ranks.  Returns one summary dict per SNR point."""
    import torch

    dec = dec or Decoder(code)
    out = []
    for si, snr in enumerate(snrs_db):
        total = torch.zeros((5,), dtype=torch.int64, device=device)
        left, chunk = frames_per_point, 0
        while left > 0:
            b = min(left, batch)
            bt = make_batch(code, G, b, float(snr), cfg, seed=[seed, si, chunk], device=device)
            decode_batch(dec, bt, cfg, compute_dtype=compute_dtype)
            count_errors(dec, bt)
            total += bt.counters
            left -= b
            chunk += 1
        allreduce_counters(total)
        torch.cuda.synchronize()
        summ = summarize(total.cpu().numpy(), code.n)
        summ["es_n0_db"] = float(snr)
        out.append(summ)
    return out
