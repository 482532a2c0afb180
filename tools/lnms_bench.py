#!/usr/bin/env python3
"""Layered min-sum throughput on the packaged codes (fixed 10 iterations and early exit)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import codes, config, sweep
from ldpc_amd.decoder import Decoder
dev = torch.device("cuda", 0)
# "qc": the table-driven layered kernel instead of the compiled-in one; "graph": the base-graph kernel (run-time shifts)
KERNEL = 2 if "qc" in sys.argv[1:] else 3 if "graph" in sys.argv[1:] else 0
dtypes = [a for a in sys.argv[1:] if a in ("f32", "f64")] or ["f32", "f64"]
names = [a for a in sys.argv[1:] if a not in ("f32", "f64", "qc", "graph")] or ["nr_bg1_z384", "nr_bg2_z40", "nr_bg2_z40_r1400", "wifi648"]
for name in names:
    code = codes.load_packaged(name); dec = Decoder(code); k = code.n - code.m
    frames = 16384 if code.n > 4096 else 1 << 18
    for dtype in dtypes:
        for early, snr in ((False, -3.0), (True, -3.0), (True, 0.0)):
            cfg = config.LNMS.with_(max_iters=10, early_exit=early)
            batch = sweep.make_batch(code, None, frames, snr, cfg, seed=[1], device=dev,
                                     llr_dtype=torch.float32)
            comp = np.float64 if dtype == "f64" else np.float32
            try:
                kern = sweep.decode_batch(dec, batch, cfg, compute_dtype=comp, kernel=KERNEL)
            except Exception as e:
                print(name, dtype, "unsupported", e); break
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                sweep.decode_batch(dec, batch, cfg, compute_dtype=comp, kernel=KERNEL)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
            sweep.count_errors(dec, batch); torch.cuda.synchronize()
            st = sweep.summarize(batch.counters.cpu().numpy(), code.n)
            print(f"{name:18s} {dtype} {kern:14s} early={int(early)} snr={snr:5.1f} {dt*1e3:9.2f} ms "
                  f"{frames*k/dt/1e9:8.3f} Gbit/s  iters={st['mean_iters']:.2f} fer={st['fer']:.4f}", flush=True)
