#!/usr/bin/env python3
"""Low error rates of the layered kernels on NR BG1 Z = 384 through the fused sweep (random messages encoded
on the device, channel + decode + accounting in one launch, transmitted words as bits): float32 and packed
half precision on the same codewords and the same noise, many frames per point past the waterfall.  Does
the reduced precision show an error floor or undetected errors where float32 has none?

    python tools/lnms_fused_deep.py OUT.json LOG2FRAMES@SNR_DB ...      e.g.  out.json 22@-3.5 24@-3.25
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
from ldpc_amd.decoder import Decoder  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    out_name = sys.argv[1]
    points = [(int(a.split("@")[0]), float(a.split("@")[1])) for a in sys.argv[2:]]
    code = codes.load_packaged("nr_bg1_z384")
    G = encoder.device_encoder(code, None)   # messages drawn and encoded on the device
    dec = Decoder(code)
    cfg = config.LNMS.with_(max_iters=10, early_exit=True)
    chunk = 1 << 15
    out = {}
    for log2, snr in points:
        chunks = max(1, (1 << log2) // chunk)
        tot = {"f32": torch.zeros((5,), dtype=torch.int64, device=dev), "f16x2": torch.zeros((5,), dtype=torch.int64, device=dev)}
        t0 = time.perf_counter()
        for c in range(chunks):
            cw = sweep.make_codewords(code, G, chunk, [91, int(snr * 100) + 1000, c], dev)
            bits = sweep.pack_bits(cw)
            del cw
            for name, dt in (("f32", np.float32), ("f16x2", np.float16)):
                sweep.sweep_fused(dec, bits, snr, cfg, [92, int(snr * 100) + 1000], tot[name], compute_dtype=dt,
                                  first_frame=c * chunk, packed_n=code.n)
            del bits
            if c % 64 == 63:
                torch.cuda.synchronize()
                print(f"  Es/N0 {snr}: {(c + 1) * chunk} frames, {time.perf_counter() - t0:.0f} s", flush=True)
        torch.cuda.synchronize()
        rec = {"frames": chunks * chunk, "seconds": round(time.perf_counter() - t0, 1)}
        for name in tot:
            s = sweep.summarize(tot[name].cpu().numpy(), code.n)
            rec[name] = {k: s[k] for k in ("not_converged", "converged_wrong", "bit_errors", "mean_iters", "fer")}
        out[f"nr_bg1_z384@{snr}"] = rec
        print(snr, json.dumps(rec), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", out_name), "w"), indent=1)


if __name__ == "__main__":
    main()
