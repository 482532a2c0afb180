"""Manual check of codes registered with `python -m ldpc_amd.add_code` (not collected by pytest: the
library has to be rebuilt around it).  On a GPU box, after registering and rebuilding:

    python tests/manual_user_code_check.py NAME [NAME ...]

Every rule, early exit and fixed iterations, in double against the CPU restatement (min-sum rules
bit for bit, sum-product within 1e-4 relative), and the float32 kernels' convergence flags.
Last run on two synthetic tables made for the purpose - Z = 48, 12 x 24, 134 edges and Z = 20,
5 x 10, 32 edges with three frames per wave, each with a double-circulant block, a variable in one
check only and a variable in none: all rules identical to the restatement.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ldpc_amd import codes, config
from ldpc_amd.decoder import Decoder
from ldpc_amd._lib import LdpcError
from oracle import bp_oracle
bp_oracle.build_c_oracle()
fails = 0
for name in sys.argv[1:]:
    code = codes.load_packaged(name)
    dec = Decoder(code)
    for rule in ("R0", "R1", "R2", "R3", "R4", "LNMS"):
        for early in (True, False):
            cfg = config.PRESETS[rule].with_(max_iters=8, early_exit=early)
            rng = np.random.default_rng([5, cfg.rule])
            B = 301
            snr = 1.0
            sigma2 = 1.0 / (2.0 * 10 ** (snr / 10))
            cw = np.zeros((B, code.n))
            x = 2.0 * cw - 1.0 if cfg.plus_is_bit == 1 else 1.0 - 2.0 * cw
            L = bp_oracle.llr_from_samples(x + np.sqrt(sigma2) * rng.standard_normal(x.shape), snr, cfg.rule)
            try:
                res = dec.decode(L, cfg, compute_dtype=np.float64, want_posterior=True)
            except LdpcError as e:
                print(name, rule, early, "unsupported:", str(e)[:60]); continue
            eh, ei, ec, ep = bp_oracle.c_decode(code, cfg, L)
            if cfg.rule in (0, 1, 5):
                ok = np.array_equal(res.converged, ec) and np.array_equal(res.iters, ei) and np.array_equal(res.hard, eh) and np.array_equal(res.posterior, ep)
            else:
                same = (res.converged == ec) & (res.iters == ei)
                scale = np.maximum(np.abs(ep[same]), 1.0)
                ok = same.mean() >= 0.95 and np.array_equal(res.hard[same & ec], eh[same & ec]) and float(np.max(np.abs(res.posterior[same] - ep[same]) / scale, initial=0)) <= 1e-4
            fails += not ok
            print(name, rule, "early" if early else "fixed", res.kernel, "conv", float(ec.mean()), "OK" if ok else "MISMATCH", flush=True)
            res32 = dec.decode(L.astype(np.float32), cfg, compute_dtype=np.float32)
            agree = (res32.converged == ec).mean()
            print("   float32:", res32.kernel, "flags agree", round(float(agree), 3), flush=True)
print("fails", fails)
sys.exit(1 if fails else 0)
