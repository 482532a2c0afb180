#!/usr/bin/env python3
"""Same workload through each kernel family (auto / edge-list / table-driven circulant)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import codes, config, sweep
from ldpc_amd.decoder import Decoder
from ldpc_amd._lib import LdpcError
dev = torch.device("cuda", 0)
for name in sys.argv[1:] or ["nr_bg2_z40", "nr_bg2_z40_r1400", "wifi648"]:
    code = codes.load_packaged(name); dec = Decoder(code); k = code.n - code.m
    frames = max(512, min(1 << 16, (1 << 27) // (4 * code.n)))
    for rule in ["R0", "R1", "R3"]:
        for dtype in ("f64", "f32"):
            cfg = config.PRESETS[rule].with_(max_iters=10, early_exit=False)
            batch = sweep.make_batch(code, None, frames, -2.0, cfg, seed=[1], device=dev)
            comp = np.float64 if dtype == "f64" else np.float32
            for kernel in (0, 1, 2):
                try:
                    kern = sweep.decode_batch(dec, batch, cfg, compute_dtype=comp, kernel=kernel)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    sweep.decode_batch(dec, batch, cfg, compute_dtype=comp, kernel=kernel)
                    torch.cuda.synchronize(); dt = time.perf_counter() - t0
                    print(f"{name:18s} {rule} {dtype} {kern:20s} {dt*1e3:9.2f} ms {frames*k/dt/1e9:8.3f} Gbit/s", flush=True)
                except LdpcError as e:
                    print(name, rule, dtype, kernel, "unsupported")
