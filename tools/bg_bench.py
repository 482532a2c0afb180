#!/usr/bin/env python3
"""Base-graph layered kernel (run-time lifting size and shifts) over lifting sizes: exactly 10 sweeps, float32,
Gbit/s of information bits; tables lifted from the packaged ones with P = V mod Z."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import codes, config, sweep, nr_rate_match as rm
from ldpc_amd.decoder import Decoder
dev = torch.device("cuda", 0)
dt = np.float64 if "f64" in sys.argv[1:] else np.float32
for bg, name, kb in ((1, "nr_bg1_z384", 22), (2, "nr_bg2_z40", 10)):
    base = codes.load_packaged(name).qc
    for Z in (8, 24, 40, 64, 96, 128, 160, 208, 256, 320, 384):
        code = codes.compile_qc(rm.lift(base, Z), f"bg{bg}_z{Z}")
        dec = Decoder(code)
        frames = max(2048, min(1 << 18, (1 << 28) // code.n // 4))
        for early, snr in ((False, -2.0), (True, -2.0)):
            cfg = config.LNMS.with_(max_iters=10, early_exit=early)
            batch = sweep.make_batch(code, None, frames, snr, cfg, seed=[1], device=dev, llr_dtype=torch.float32)
            kern = sweep.decode_batch(dec, batch, cfg, compute_dtype=dt, kernel=0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                sweep.decode_batch(dec, batch, cfg, compute_dtype=dt, kernel=0)
            torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 3
            sweep.count_errors(dec, batch); torch.cuda.synchronize()
            st = sweep.summarize(batch.counters.cpu().numpy(), code.n)
            print(f"BG{bg} Z={Z:3d} n={code.n:5d} {kern:13s} early={int(early)} frames={frames:6d} {el*1e3:8.2f} ms "
                  f"{frames*kb*Z/el/1e9:7.2f} Gbit/s iters={st['mean_iters']:.2f} fer={st['fer']:.3f}", flush=True)
            del batch
