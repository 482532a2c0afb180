#!/usr/bin/env python3
"""Frame and bit error rates of the layered rule (LNMS, 10 sweeps) beside the reference's flooding
rule R1 (bpsk.py:154-260) at 10 and 20 iterations, on the reference's own matrices; float64,
2^17 frames per point, same frames for the three decoders.  Prints JSON."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import codes, config, sweep  # noqa: E402
from ldpc_amd.decoder import Decoder  # noqa: E402

dev = torch.device("cuda", 0)
POINTS = {"wifi648": [-1.5, -1.0, -0.5, 0.0], "nr_bg2_z40": [-6.5, -6.0, -5.5, -5.0]}
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17
out = {}
for name, snrs in POINTS.items():
    code = codes.load_packaged(name)
    G = codes.load_packaged_generator(name) if name == "wifi648" else None
    dec = Decoder(code)
    for label, cfg in (("R1_10", config.R1.with_(max_iters=10)), ("R1_20", config.R1.with_(max_iters=20)),
                       ("LNMS_10", config.LNMS.with_(max_iters=10)), ("LNMS_5", config.LNMS.with_(max_iters=5))):
        res = sweep.ber_sweep(code, G, cfg, snrs, frames, dev, np.float64, seed=17, dec=dec)
        out[f"{name}/{label}"] = [{k: r[k] for k in ("es_n0_db", "frames", "not_converged", "converged_wrong", "bit_errors", "fer", "ber", "mean_iters")} for r in res]
        print(name, label, [(r["es_n0_db"], r["not_converged"], r["bit_errors"]) for r in res], file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
