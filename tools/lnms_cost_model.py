#!/usr/bin/env python3
"""Layered kernel with early exit: time per launch against mean sweeps over a range of SNRs (BG1), to
separate the cost of a sweep from the per-frame cost (load, successful syndrome test, write-out)."""
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
rows = []
for early, snr, iters in ((False, -3.0, 10), (False, -3.0, 5), (False, -3.0, 1), (True, -3.5, 20), (True, -3.0, 20), (True, -2.0, 20),
                          (True, 0.0, 20), (True, 3.0, 20), (True, 12.0, 20)):
    cfg = config.LNMS.with_(max_iters=iters, early_exit=early)
    batch = sweep.make_batch(code, None, frames, snr, cfg, seed=[1], device=dev, llr_dtype=torch.float32)
    sweep.decode_batch(dec, batch, cfg, compute_dtype=np.float32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        sweep.decode_batch(dec, batch, cfg, compute_dtype=np.float32)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    sweep.count_errors(dec, batch); torch.cuda.synchronize()
    st = sweep.summarize(batch.counters.cpu().numpy(), code.n)
    it = batch.iters.cpu().numpy()
    grp = it.reshape(-1, 2).max(axis=1).mean() if early else iters
    print(f"{name} early={int(early)} snr={snr:5.1f} max={iters:2d} {dt*1e3:8.3f} ms  mean sweeps={st['mean_iters']:.3f} "
          f"mean of pair maxima={grp:.3f} fer={st['fer']:.4f}", flush=True)
