#!/usr/bin/env python3
"""Throughput of every (code, rule, dtype) the library serves: fixed 10 iterations (early_exit off),
LLRs resident in HBM.  Prints one line per combination; used to find the slow paths."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import codes, config, sweep  # noqa: E402
from ldpc_amd.decoder import Decoder  # noqa: E402
from ldpc_amd._lib import LdpcError  # noqa: E402

dev = torch.device("cuda", 0)
only = set(sys.argv[1:])
for name in ["ccsds512", "wifi648", "nr_bg2_z40", "nr_bg2_z40_r1400", "nr_bg1_z384"]:
    if only and name not in only:
        continue
    code = codes.load_packaged(name)
    dec = Decoder(code)
    k = code.n - code.m
    frames = max(512, min(1 << 18, (1 << 28) // (4 * code.n)))
    for rule in ["R0", "R1", "R2", "R3", "R4", "LNMS"]:
        for dtype in ("f64", "f32"):
            cfg = config.PRESETS[rule].with_(max_iters=10, early_exit=False)
            batch = sweep.make_batch(code, None, frames, -2.0, cfg, seed=[1], device=dev)
            comp = np.float64 if dtype == "f64" else np.float32
            try:
                kern = sweep.decode_batch(dec, batch, cfg, compute_dtype=comp)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 2
                for _ in range(reps):
                    sweep.decode_batch(dec, batch, cfg, compute_dtype=comp)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / reps
                print(f"{name:18s} {rule:5s} {dtype} {kern:20s} {frames:7d} frames {dt*1e3:9.2f} ms "
                      f"{frames * k / dt / 1e9:8.3f} Gbit/s", flush=True)
            except LdpcError as e:
                print(f"{name:18s} {rule:5s} {dtype} unsupported: {str(e)[:70]}", flush=True)
            del batch
