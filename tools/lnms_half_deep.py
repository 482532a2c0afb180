#!/usr/bin/env python3
"""Low error rates of the packed half-precision layered kernel against float32 on the same frames:
does the reduced precision (half-precision posteriors, eight-bit check magnitudes) show an error floor
or undetected errors where float32 has none?  Many frames per point, a few SNRs past the waterfall."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import codes, config, sweep
from ldpc_amd.decoder import Decoder
dev = torch.device("cuda", 0)
cases = [("wifi648", 1 << 21, 2, [0.5, 1.0, 1.5, 2.0]), ("nr_bg2_z40", 1 << 19, 2, [-4.5, -4.0, -3.5]),
         ("nr_bg1_z384", 1 << 15, 2, [-3.6, -3.5, -3.25])]
tag = ""
if len(sys.argv) > 2 and sys.argv[1] == "bg1":   # python tools/lnms_half_deep.py bg1 CHUNKS: the long code only, CHUNKS x 2^15 frames per point
    cases = [("nr_bg1_z384", 1 << 15, int(sys.argv[2]), [-3.6, -3.5, -3.25])]
    tag = "_bg1"
out = {}
for name, frames, chunks, snrs in cases:
    code = codes.load_packaged(name); dec = Decoder(code)
    G = codes.load_packaged_generator(name) if name in ("wifi648", "nr_bg1_z384") else None
    cfg = config.LNMS.with_(max_iters=10, early_exit=True)
    for snr in snrs:
        tot = {np.float32: np.zeros(5, np.int64), np.float16: np.zeros(5, np.int64)}
        differ = 0
        for ch in range(chunks):
            bt = sweep.make_batch(code, G, frames, snr, cfg, seed=[77, int(snr * 100) + 1000, ch], device=dev, native_noise=True)
            words = {}
            for dt in tot:
                sweep.decode_batch(dec, bt, cfg, compute_dtype=dt)
                bt.counters.zero_(); sweep.count_errors(dec, bt); torch.cuda.synchronize()
                tot[dt] += bt.counters.cpu().numpy()
                words[dt] = (bt.converged.clone(), bt.hard.clone())
            both = words[np.float32][0].bool() & words[np.float16][0].bool()
            differ += int((words[np.float32][1][both] != words[np.float16][1][both]).any(dim=1).sum())
            del bt, words
        s32, s16 = sweep.summarize(tot[np.float32], code.n), sweep.summarize(tot[np.float16], code.n)
        rec = {"frames": int(s32["frames"]), "f32": {k: s32[k] for k in ("not_converged", "converged_wrong", "bit_errors", "mean_iters")},
               "f16x2": {k: s16[k] for k in ("not_converged", "converged_wrong", "bit_errors", "mean_iters")},
               "both_converged_to_different_words": differ}
        out[f"{name}@{snr}"] = rec
        print(name, snr, json.dumps(rec), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"lnms_half_deep{tag}.json"), "w"), indent=1)
