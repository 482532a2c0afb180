#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — golden vectors of the reference's bit-flipping decoder
(/root/reference/ldpc.py:82-135, bitFlipDecode).  Runs only where /root/reference is mounted: the
method is taken from the reference's class definition (oracle/gen_golden.py's extraction) and run
UNMODIFIED; a subclass that adds no behaviour records the bits and the pass count at every
syndrome test.  Writes tests/golden/BF_<code>.npz: channel samples, transmitted codewords, the
method's return value (bit error rate), whether and after how many passes a test succeeded, the
bits seen at the last test, and for the first frames the bits at every test.

    python oracle/gen_golden_bitflip.py
"""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden as gg  # noqa: E402
from ldpc_amd import codes, shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
#: (label, matrix, generator or None, Es/N0 dB, frames, m passed as self.m)
CASES = [("ccsds512", 5.0, 24, None), ("wifi648", 6.0, 16, None), ("wifi648", 5.0, 8, 100)]


def main():
    base, plt = gg.reference_class("ldpc.py")

    class Probed(base.__mro__[1]):
        def isValidCodeword(self, x):
            loc = sys._getframe(1).f_locals
            if "bitNodes" in loc and "numIterations" in loc and "bitNodeMessages" in loc:
                self._bf.append((int(loc["numIterations"]), np.array(loc["bitNodes"], dtype=np.uint8)))
            return super().isValidCodeword(x)

    for ci, (name, snr, frames, m_attr) in enumerate(CASES):
        cc = codes.load_packaged(name)
        G = codes.load_packaged_generator(name)
        rng = np.random.default_rng([82, ci])
        cw = codes.encode(G, rng.integers(0, 2, size=(frames, G.shape[0]))).astype(np.uint8)
        sigma2 = 1.0 / (2.0 * 10.0 ** (snr / 10.0))
        y = ((1.0 - 2.0 * cw) + np.sqrt(sigma2) * rng.standard_normal(cw.shape)).astype(np.float32).astype(np.float64)
        ret = np.zeros(frames)
        conv = np.zeros(frames, bool)
        iters = np.zeros(frames, np.int32)
        last = np.zeros((frames, cc.n), np.uint8)
        traces = []
        for f in range(frames):
            o = object.__new__(Probed)
            o.H = cc.dense().astype(int)
            o.m = cc.m if m_attr is None else m_attr
            o.n = cc.n
            o.originalEncoded = cw[f].astype(int)
            o._bf = []
            with contextlib.redirect_stdout(io.StringIO()):
                ret[f] = o.bitFlipDecode(y[f].copy())
            it, bits = o._bf[-1]
            # the method returns from inside the loop exactly when the last test passed
            conv[f] = bool(cc.is_codeword(bits)) and len(o._bf) == it + 1
            iters[f] = it if conv[f] else 10
            last[f] = bits
            if f < 4:
                traces.append(np.stack([b for _, b in o._bf] + [np.full(cc.n, 255, np.uint8)] * (10 - len(o._bf))))
        path = os.path.join(OUT, f"BF_{name}_m{m_attr or 'all'}.npz")
        np.savez_compressed(path, rule="BF", code=name, y=y, codeword=cw, ret=ret, converged=conv, iters=iters,
                            last_tested_bits=last, trace=np.stack(traces), m_attr=-1 if m_attr is None else m_attr,
                            snr=snr)
        print(path, "converged", int(conv.sum()), "of", frames, "iters", iters.tolist())


if __name__ == "__main__":
    main()
