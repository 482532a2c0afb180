#!/usr/bin/env python3
"""Rate of the host-buffer entry point (ldpc_decode_batch_host): allocation, PCIe both ways, decode."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import codes, config
from ldpc_amd.decoder import Decoder
code = codes.load_packaged("ccsds512"); dec = Decoder(code)
rng = np.random.default_rng(0)
for frames in (65536, 1 << 19):
    L = (4.0 * (-1.0 + 0.8 * rng.standard_normal((frames, code.n)))).astype(np.float32)
    dec.decode(L[:1024], config.R0)
    t0 = time.perf_counter(); r = dec.decode(L, config.R0); dt = time.perf_counter() - t0
    print(f"host entry point, {frames} CCSDS frames (f32 LLRs, rule R0 f64): {dt*1e3:.1f} ms, "
          f"{frames*256/dt/1e9:.2f} Gbit/s info, {frames*(5*code.n+5)/dt/1e9:.2f} GB/s over the boundary, kernel {r.kernel}")

# one frame per call through the reference-shaped class (what a script of the reference does)
from ldpc_amd import shim
enc = shim.BpskPyEncoder(4, 8, 512, H=code.dense())
enc.originalEncoded = np.zeros(code.n, dtype=int)
enc.SNR = 2.0
y = -1.0 + 0.56 * rng.standard_normal((2000, code.n))
enc.minSumDecode(y[0])
t0 = time.perf_counter()
for i in range(len(y)):
    enc.minSumDecode(y[i])
dt = time.perf_counter() - t0
print(f"frame-at-a-time through the reference-shaped class (R1, float64, posteriors back): {dt/len(y)*1e6:.0f} us per call")
