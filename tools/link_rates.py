#!/usr/bin/env python3
"""Rates of the byte-work kernels around the decoder (encoder, rate matching, rate recovery, channel,
error counting) on the BG1-shaped Z = 384 table: GB/s of their algorithmic bytes, buffers allocated
outside the timed loop."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import _lib, channel, codes, config, encoder, nr_rate_match, sweep  # noqa: E402
from ldpc_amd.decoder import Decoder  # noqa: E402
import ctypes  # noqa: E402

dev = torch.device("cuda", 0)
code = codes.load_packaged("nr_bg1_z384")
enc = encoder.device_encoder(code)
frames = 16384
stream = torch.cuda.current_stream().cuda_stream
lib = _lib.load()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


msg = torch.randint(0, 2, (frames, enc.k), device=dev, dtype=torch.uint8)
cw = torch.empty((frames, enc.n), dtype=torch.uint8, device=dev)
t = timed(lambda: enc.encode_ptr(msg.data_ptr(), frames, cw.data_ptr(), stream))
print(f"encode            {frames * (enc.k + enc.n) / t / 1e9:8.1f} GB/s  {t * 1e3:.3f} ms")
for Qm in (1, 4):
    r = nr_rate_match.RateMatch(1, 384, 2 * 8448, Qm=Qm)
    cfg = r.c_config(enc.n)
    air = torch.empty((frames, r.E), dtype=torch.uint8, device=dev)
    t = timed(lambda: _lib.check(lib.ldpc_rate_match(ctypes.byref(cfg), cw.data_ptr(), enc.n, frames, air.data_ptr(), stream or None)))
    print(f"rate_match Qm={Qm}   {frames * 2 * r.E / t / 1e9:8.1f} GB/s  {t * 1e3:.3f} ms")
    llr = torch.randn((frames, r.E), device=dev, dtype=torch.float32)
    n_out = r.n_active()
    out = torch.empty((frames, n_out), device=dev, dtype=torch.float32)
    cfg2 = r.c_config(n_out)
    t = timed(lambda: _lib.check(lib.ldpc_rate_recover(ctypes.byref(cfg2), llr.data_ptr(), 0, frames, 100.0, 0, out.data_ptr(), 0, stream or None)))
    print(f"rate_recover Qm={Qm} {frames * 4 * (r.E + n_out) / t / 1e9:8.1f} GB/s  {t * 1e3:.3f} ms")
    del llr, out, air
llr = torch.empty((frames, enc.n), device=dev, dtype=torch.float32)
t = timed(lambda: channel.channel_llr_device(cw.data_ptr(), enc.n, frames, enc.n, 0.0, config.LNMS, 7, 0, llr.data_ptr(), np.float32, stream=stream))
print(f"channel           {frames * 5 * enc.n / t / 1e9:8.1f} GB/s  {t * 1e3:.3f} ms")
dec = Decoder(code)
hard = torch.zeros((frames, enc.n), dtype=torch.uint8, device=dev)
iters = torch.zeros((frames,), dtype=torch.int32, device=dev)
conv = torch.ones((frames,), dtype=torch.uint8, device=dev)
ctr = torch.zeros((5,), dtype=torch.int64, device=dev)
t = timed(lambda: dec.count_errors_device(hard.data_ptr(), cw.data_ptr(), enc.n, iters.data_ptr(), conv.data_ptr(), frames, ctr.data_ptr(), stream=stream))
print(f"count_errors      {frames * (2 * enc.n + 5) / t / 1e9:8.1f} GB/s  {t * 1e3:.3f} ms")
# the CCSDS encoder (dense generator circulants: ballot-mask kernel), random messages drawn on the device
c2 = codes.load_packaged("ccsds512")
e2 = encoder.device_encoder(c2, codes.load_packaged_generator("ccsds512"))
fr2 = 1 << 20
cw2 = torch.empty((fr2, c2.n), dtype=torch.uint8, device=dev)
t = timed(lambda: _lib.check(lib.ldpc_encode_random(e2._h, 5, 0, fr2, cw2.data_ptr(), None, stream or None)))
print(f"encode ccsds rnd  {fr2 * c2.n / t / 1e9:8.1f} GB/s  {t * 1e3:.3f} ms  {fr2 * 256 / t / 1e9:.1f} Gbit/s info")
