#!/usr/bin/env python3
"""Fixed-work layered decode time against the number of sweeps (constant part + per-sweep part)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import codes, config, sweep
from ldpc_amd.decoder import Decoder
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "nr_bg1_z384"
code = codes.load_packaged(name); dec = Decoder(code); k = code.n - code.m
frames = 16384 if code.n > 4096 else 1 << 18
base = config.LNMS.with_(early_exit=False)
batch = sweep.make_batch(code, None, frames, -3.0, base, seed=[1], device=dev, llr_dtype=torch.float32)
for it in (1, 2, 4, 10, 20):
    cfg = base.with_(max_iters=it)
    sweep.decode_batch(dec, batch, cfg, compute_dtype=np.float32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        sweep.decode_batch(dec, batch, cfg, compute_dtype=np.float32)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"{name} fixed iters={it:2d} {dt*1e3:8.3f} ms", flush=True)
