#!/usr/bin/env python3
"""Channel + decode + error accounting as one launch (ldpc_sweep_batch) against the same work as
three launches, for any compiled-in code.

    python tools/fused_bench.py CODE RULE DTYPE SNR_DB FRAMES [fixed] ...   (groups of five or six words)
    e.g.  python tools/fused_bench.py nr_bg1_z384 LNMS f16 -3 16384  nr_bg2_z40 R1 f32 -4 32768 fixed
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, torch, steps=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    import torch

    from ldpc_amd import codes, config, sweep
    from ldpc_amd.decoder import Decoder
    dev = torch.device("cuda", 0)
    words = sys.argv[1:]
    while words:
        name, rule, dtype, snr, frames = words[:5]
        words = words[5:]
        fixed = bool(words) and words[0] == "fixed"
        if fixed:
            words = words[1:]
        snr, frames = float(snr), int(frames)
        dt = {"f64": np.float64, "f32": np.float32, "f16": np.float16}[dtype]
        code = codes.load_packaged(name)
        try:
            G = codes.load_packaged_generator(name)
        except Exception:
            G = None
        dec = Decoder(code)
        cfg = config.PRESETS[rule].with_(max_iters=10, early_exit=not fixed)
        cw = sweep.make_codewords(code, G, frames, [31], dev)
        ctr = torch.zeros((5,), dtype=torch.int64, device=dev)
        kern = {}

        def fused():
            kern["fused"] = sweep.sweep_fused(dec, cw, snr, cfg, [32], ctr, compute_dtype=dt)

        def three():
            bt = sweep.noisy_batch(code, cw, snr, cfg, [32], dev)
            kern["three"] = sweep.decode_batch(dec, bt, cfg, compute_dtype=dt)
            sweep.count_errors(dec, bt)

        def decode_only(bt=sweep.noisy_batch(code, cw, snr, cfg, [32], dev)):
            sweep.decode_batch(dec, bt, cfg, compute_dtype=dt)

        t1, t3, td = timed(fused, torch), timed(three, torch), timed(decode_only, torch)
        k = code.n - code.m
        print(f"{name} {rule} {dtype} {snr:+.1f} dB {frames} frames {'fixed 10' if fixed else 'early exit'}: "
              f"fused {t1 * 1e3:.3f} ms ({frames * k / t1 / 1e9:.2f} Gbit/s, {kern['fused']}), "
              f"three launches {t3 * 1e3:.3f} ms ({frames * k / t3 / 1e9:.2f} Gbit/s, {kern['three']}), "
              f"decode alone {td * 1e3:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
