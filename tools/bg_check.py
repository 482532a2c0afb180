#!/usr/bin/env python3
"""Base-graph layered kernel in float against the C restatement on a few BG1-shaped codes (development aid:
works with a library built for BG1 / float only)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import codes, config, nr_bg1, nr_rate_match as rm
from ldpc_amd.decoder import Decoder
from oracle import bp_oracle
bp_oracle.build_c_oracle()
ok = True
for Z, rows, snr in ((384, 46, -0.8), (384, 12, 3.0), (96, 26, 1.0), (7, 46, 1.0)):
    qc = rm.active_submatrix(rm.lift(nr_bg1.build(Z=384, seed=Z + rows), Z), rows)
    code = codes.compile_qc(qc, "t")
    rng = np.random.default_rng(Z)
    s2 = 1 / (2 * 10 ** (snr / 10))
    L = ((2 * (1.0 + np.sqrt(s2) * rng.standard_normal((33, code.n)))) / s2).astype(np.float32)
    for cfg in (config.LNMS, config.LNMS.with_(early_exit=False, max_iters=4)):
        eh, ei, ec, ep = bp_oracle.c_decode_lnms(code, cfg, L.astype(np.float64), use_float=True)
        r = Decoder(code).decode(L, cfg, compute_dtype=np.float32, want_posterior=True)
        same = (np.array_equal(r.hard, eh) and np.array_equal(r.iters, ei) and np.array_equal(r.converged, ec)
                and np.array_equal(r.posterior.astype(np.float64), ep))
        ok &= same
        print(f"Z={Z} rows={rows} early={cfg.early_exit} {r.kernel}: {'identical' if same else 'DIFFERENT'} "
              f"(converged {int(ec.sum())}/33, mean iters {ei.mean():.2f})", flush=True)
sys.exit(0 if ok else 1)
