import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/ldpc_amd") else os.getcwd())
from ldpc_amd import codes, config, sweep
from ldpc_amd.decoder import Decoder
dev = torch.device("cuda", 0)
for name in ("ccsds512", "wifi648"):
    code = codes.load_packaged(name); dec = Decoder(code); k = code.n - code.m
    frames = 1 << 19
    for early, snr in ((False, 1.0), (True, 1.0)):
        cfg = config.R1.with_(max_iters=10, early_exit=early)
        bt = sweep.make_batch(code, None, frames, snr, cfg, seed=[1], device=dev)
        for dt in (np.float32, np.float16):
            kern = sweep.decode_batch(dec, bt, cfg, compute_dtype=dt); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): sweep.decode_batch(dec, bt, cfg, compute_dtype=dt)
            torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 3
            print(f"{name} early={int(early)} {kern:16s} {dtm*1e3:8.2f} ms {frames*k/dtm/1e9:8.2f} Gbit/s", flush=True)
