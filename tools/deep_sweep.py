#!/usr/bin/env python3
"""Error-rate curves far below what the reference's one-frame-at-a-time loop (ldpc.py:637-667, about
a second per frame) can reach: the fused sweep (channel + decode + counting in one launch) on the
CCSDS code under the reference-pinned rule R1, and the layered rule on the BG1-shaped table.
Prints one JSON object; frames per point and the wall-clock time are part of it.

    python tools/deep_sweep.py [log2 frames per point, default 28]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import codes, config, encoder, sweep  # noqa: E402

dev = torch.device("cuda", 0)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
out = {}
cases = [("ccsds512", "R1", np.float32, [0.0, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0], 1 << lg, 30),
         ("ccsds512", "R1", np.float16, [0.0, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0], 1 << lg, 30),   # packed half precision, the same frames
         ("ccsds512", "R1", np.float64, [1.0, 2.0], 1 << (lg - 2), 30),
         ("nr_bg1_z384", "LNMS", np.float32, [-4.0, -3.75, -3.5, -3.25], 1 << (lg - 7), 20)]
for name, rule, dtype, snrs, frames, iters in cases:
    code = codes.load_packaged(name)
    G = encoder.device_encoder(code, codes.load_packaged_generator(name) if name == "ccsds512" else None)
    cfg = config.PRESETS[rule].with_(max_iters=iters)
    t0 = time.perf_counter()
    res = sweep.ber_sweep(code, G, cfg, snrs, frames, dev, dtype, seed=2026, fused=True, batch=1 << (20 if code.n < 4096 else 15))
    dt = time.perf_counter() - t0
    key = f"{name}/{rule}/{np.dtype(dtype).name}"
    out[key] = {"frames_per_point": frames, "max_iters": iters, "seconds": dt,
                "points": [{k: r[k] for k in ("es_n0_db", "frames", "not_converged", "converged_wrong", "bit_errors", "fer", "ber", "mean_iters")} for r in res]}
    print(key, f"{dt:.1f} s", [(r["es_n0_db"], r["not_converged"], r["converged_wrong"], r["bit_errors"]) for r in res], file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
