#!/usr/bin/env python3
"""Layered min-sum in packed half precision against the float32 kernel on the same frames:
error rates, agreement of flags / words / iteration counts, and throughput."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import codes, config, sweep
from ldpc_amd.decoder import Decoder
dev = torch.device("cuda", 0)
cases = {"nr_bg1_z384": (16384, [-4.0, -3.75, -3.0, 0.0]), "nr_bg2_z40": (1 << 16, [-6.5, -6.0, -5.0, -3.0]),
         "nr_bg2_z40_r1400": (1 << 16, [-5.0, -4.5, -3.0]), "wifi648": (1 << 17, [-1.5, -1.0, 0.0])}
names = [a for a in sys.argv[1:]] or list(cases)
for name in names:
    frames, snrs = cases[name]
    code = codes.load_packaged(name); dec = Decoder(code); k = code.n - code.m
    G = codes.load_packaged_generator(name) if name in ("wifi648", "nr_bg1_z384") else None
    for early in (True, False):
        for snr in snrs:
            cfg = config.LNMS.with_(max_iters=10, early_exit=early)
            bt = sweep.make_batch(code, G, frames + 1, snr, cfg, seed=[3, int(snr * 4) + 40], device=dev, native_noise=True)
            out = {}
            for dt in (np.float32, np.float16):
                kern = sweep.decode_batch(dec, bt, cfg, compute_dtype=dt)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(3):
                    sweep.decode_batch(dec, bt, cfg, compute_dtype=dt)
                torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 3
                bt.counters.zero_(); sweep.count_errors(dec, bt); torch.cuda.synchronize()
                st = sweep.summarize(bt.counters.cpu().numpy(), code.n)
                out[dt] = (st, bt.converged.clone(), bt.hard.clone(), bt.iters.clone(), kern, dtm)
            s32, c32, h32, i32, k32, t32 = out[np.float32]
            s16, c16, h16, i16, k16, t16 = out[np.float16]
            both = c16.bool() & c32.bool()
            cw_ok = code.is_codeword(h16[torch.nonzero(c16.bool())[:64, 0]].cpu().numpy()).all() if c16.any() else True
            print(f"{name:17s} early={int(early)} snr={snr:5.2f} | f32 {k32} {(frames+1)*k/t32/1e9:6.2f} Gbit/s fer={s32['fer']:.4f} "
                  f"ber={s32['ber']:.2e} it={s32['mean_iters']:.2f} cwrong={s32['converged_wrong']} | f16 {k16} {(frames+1)*k/t16/1e9:6.2f} Gbit/s "
                  f"fer={s16['fer']:.4f} ber={s16['ber']:.2e} it={s16['mean_iters']:.2f} cwrong={s16['converged_wrong']} | flag differs "
                  f"{(c16 != c32).float().mean().item():.4f} word differs (both conv) {(h16[both] != h32[both]).any(dim=1).float().mean().item() if both.any() else 0:.5f} "
                  f"converged are codewords {bool(cw_ok)}", flush=True)
