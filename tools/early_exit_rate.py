#!/usr/bin/env python3
"""Early-exit efficiency of a flooding rule: Gbit/s at several Es/N0 against the fixed-iteration rate.
    python tools/early_exit_rate.py CODE RULE DTYPE MAX_ITERS SNR [SNR ...]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import codes, config, sweep
from ldpc_amd.decoder import Decoder
name, rule, dtype, iters = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
snrs = [float(a) for a in sys.argv[5:]]
dev = torch.device("cuda", 0)
code = codes.load_packaged(name); dec = Decoder(code); k = code.n - code.m
G = codes.load_packaged_generator(name)
frames = 1 << 16
comp = np.float64 if dtype == "f64" else np.float32
for early, snr in [(False, snrs[0])] + [(True, s) for s in snrs]:
    cfg = config.PRESETS[rule].with_(max_iters=iters, early_exit=early)
    batch = sweep.make_batch(code, G, frames, snr, cfg, seed=[3], device=dev)
    kern = sweep.decode_batch(dec, batch, cfg, compute_dtype=comp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        sweep.decode_batch(dec, batch, cfg, compute_dtype=comp)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    sweep.count_errors(dec, batch); torch.cuda.synchronize()
    st = sweep.summarize(batch.counters.cpu().numpy(), code.n)
    print(f"{name} {rule} {dtype} {kern} early={int(early)} snr={snr:5.1f} {dt*1e3:8.2f} ms {frames*k/dt/1e9:8.3f} Gbit/s "
          f"iters={st['mean_iters']:.2f} fer={st['fer']:.4f}  Gbit-iterations/s={frames*k*(st['mean_iters'] + (0 if not early else 1))/dt/1e9:.1f}", flush=True)
