"""Measure how often float32 decoding differs from float64 decoding of the same frames, and how
often the device's float64 sum-product kernels differ from the numpy restatement (the one that is
bit-identical to the reference), per (rule, code) at a waterfall SNR.  Shared by
tests/test_gpu_precision.py and tools/precision_rates.py."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import codes, config, sweep  # noqa: E402
from ldpc_amd.decoder import Decoder  # noqa: E402
from oracle import bp_oracle  # noqa: E402

CASES = [("R0", "ccsds512", 0.5), ("R1", "ccsds512", -0.75), ("R1", "wifi648", -1.0), ("R1", "nr_bg2_z40", -5.5),
         ("OMS", "wifi648", -1.0), ("R2", "wifi648", -1.25), ("R3", "ccsds512", -1.0), ("R3", "nr_bg2_z40_r1400", -4.5),
         ("R4", "nr_bg2_z40_r1400", -4.5), ("LNMS", "nr_bg2_z40", -6.0), ("LNMS", "nr_bg1_z384", -3.75)]


def run(dec, batch, cfg, dtype, dev):
    post = torch.empty(batch.llr.shape, dtype=torch.float64 if dtype == np.float64 else torch.float32, device=dev)
    dec.decode_device(batch.llr.data_ptr(), np.float32, batch.llr.shape[0], batch.hard.data_ptr(), batch.iters.data_ptr(),
                      batch.converged.data_ptr(), cfg, compute_dtype=dtype, posterior_ptr=post.data_ptr(),
                      stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return batch.hard.clone(), batch.iters.clone(), batch.converged.clone(), post


def measure(cases=CASES, want=1 << 17, verbose=False):
    dev = torch.device("cuda", 0)
    out = {}
    for rule, name, snr in cases:
        code = codes.load_packaged(name)
        G = codes.load_packaged_generator(name) if name in ("ccsds512", "wifi648", "nr_bg1_z384") else None
        cfg = config.PRESETS[rule]
        frames = want if code.n <= 1024 else (want // 4 if code.n <= 4096 else 8192)
        dec = Decoder(code)
        tot = dict(frames=0, conv_diff=0, iters_diff=0, hard_diff=0, iters_diff_gt1=0, conv64=0)
        rel_by_iter = {}
        step = min(frames, 1 << 15 if code.n <= 4096 else 4096)
        for c in range(0, frames, step):
            b = min(step, frames - c)
            batch = sweep.make_batch(code, G, b, snr, cfg, seed=[41, c], device=dev, native_noise=True, first_frame=c)
            h64, i64, c64, p64 = run(dec, batch, cfg, np.float64, dev)
            h32, i32, c32, p32 = run(dec, batch, cfg, np.float32, dev)
            dconv = c64 != c32
            diters = i64 != i32
            dhard = (h64 != h32).any(dim=1)
            tot["frames"] += b
            tot["conv_diff"] += int(dconv.sum())
            tot["iters_diff"] += int(diters.sum())
            tot["iters_diff_gt1"] += int(((i64 - i32).abs() > 1).sum())
            tot["hard_diff"] += int((dhard & (c64 != 0) & (c32 != 0)).sum())   # both converged, different word
            tot["conv64"] += int(c64.sum())
            # posterior error of frames that converged alike, by iteration count
            ok = (c64 != 0) & ~dconv & ~diters
            if rule != "R0" or True:
                rel = ((p64 - p32.double()).abs() / p64.abs().clamp(min=1.0)).amax(dim=1)
                for it in torch.unique(i64[ok]).tolist():
                    m = ok & (i64 == it)
                    rel_by_iter[it] = max(rel_by_iter.get(it, 0.0), float(rel[m].max()))
        rec = {k: v for k, v in tot.items()}
        rec.update(es_n0_db=snr, fer_f64=1 - tot["conv64"] / tot["frames"],
                   max_rel_posterior_error_by_iters={str(k): v for k, v in sorted(rel_by_iter.items())})
        # the sum-product rules in float64: device (own tanh/atanh/exp/log) against numpy (the reference's)
        if rule in ("R2", "R3", "R4"):
            k = 1536 if code.n <= 1024 else 384
            batch = sweep.make_batch(code, G, k, snr, cfg, seed=[43], device=dev, native_noise=True)
            h64, i64, c64, p64 = run(dec, batch, cfg, np.float64, dev)
            L = batch.llr.cpu().numpy().astype(np.float64)
            nh, ni, nc, npost = bp_oracle.numpy_decode(code, cfg, L)
            rec["f64_vs_numpy"] = dict(frames=k, iters_diff=int((ni != i64.cpu().numpy()).sum()),
                                       conv_diff=int((nc != c64.cpu().numpy().astype(bool)).sum()),
                                       hard_diff=int((nh != h64.cpu().numpy()).any(axis=1).sum()),
                                       max_rel_posterior=float(np.max(np.abs(npost - p64.cpu().numpy()) / np.maximum(np.abs(npost), 1.0))))
        out[f"{rule}/{name}"] = rec
        if verbose:
            print(rule, name, json.dumps(rec), file=sys.stderr, flush=True)
    return out
