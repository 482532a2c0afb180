#!/usr/bin/env python3
"""Error-rate curves of the mixed CCSDS + 5G sum-product workload on one GPU: the sum-product rule R3 (channel.py:402-513) on
the CCSDS code and on the reference's 5G matrix under the Doppler-shifted BPSK channel (carrier offset of
plotDoppler.py:14-16 after 99.9 % compensation), every Monte-Carlo step one fused launch (channel + decode +
counting), float32 beside float64 on the same codewords and noise.  Prints one JSON object.

    python tools/config5_sweep.py [log2 frames per point on the small code, default 24] [rule, default R3]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import channel, codes, config, encoder, sweep  # noqa: E402

dev = torch.device("cuda", 0)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cfo = channel.doppler_cfo(4000.0, compensated=0.999)
rule = sys.argv[2] if len(sys.argv) > 2 else "R3"   # R2: the tanh rule of bpsk.py:263-387 (inner clip, no scale)
cfg = config.PRESETS[rule]
out = {"cfo_cycles_per_symbol": cfo, "rule": rule, "max_iters": cfg.max_iters}
cases = [("ccsds512", np.float32, [-1.0, -0.5, 0.0, 0.5, 1.0, 1.5, 2.0], 1 << lg),
         ("ccsds512", np.float64, [-1.0, 0.0, 1.0], 1 << (lg - 3)),
         ("nr_bg2_z40", np.float32, [-7.0, -6.5, -6.0, -5.5, -5.0, -4.5], 1 << (lg - 4)),
         ("nr_bg2_z40", np.float64, [-6.5, -5.5], 1 << (lg - 7))]
for name, dtype, snrs, frames in cases:
    code = codes.load_packaged(name)
    G = encoder.device_encoder(code, codes.load_packaged_generator(name) if name == "ccsds512" else None)
    t0 = time.perf_counter()
    res = sweep.ber_sweep(code, G, cfg, snrs, frames, dev, dtype, seed=2027, fused=True, cfo=cfo,
                          batch=1 << (19 if code.n < 1024 else 15))
    dt = time.perf_counter() - t0
    key = f"{name}/{rule}/{np.dtype(dtype).name}"
    out[key] = {"frames_per_point": frames, "seconds": dt,
                "points": [{k: r[k] for k in ("es_n0_db", "frames", "not_converged", "converged_wrong", "bit_errors", "fer", "ber", "mean_iters")} for r in res]}
    print(key, f"{dt:.1f} s", [(r["es_n0_db"], r["not_converged"], r["converged_wrong"], r["bit_errors"]) for r in res], file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
