#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — golden-vector generator.  Runs only where /root/reference is mounted;
nothing under tests/, bench.py or the product imports it at run time.

The reference scripts cannot be imported whole (missing third-party modules, experiments that run
at import).  This script parses each script's text with ``ast``, takes only the ``LDPCEncoder``
class definition, executes that definition in a numpy-only namespace, and drives the UNMODIFIED
decode methods on seeded inputs.  Outputs (return value, ``numIterations``, hard decisions,
posteriors, and for a few frames the posterior at every top-of-loop syndrome check) are written
as small ``.npz`` fixtures under tests/golden/.  No reference source text is stored.

Decode rules (the names used across this repository):

    R0  ldpc.py    minSumDecode(codeword, 2)        ldpc.py:137-227
    R1  bpsk.py    minSumDecode(codeword)           bpsk.py:154-260   (channel.py:151-257 identical)
    R2  bpsk.py    sumProductDecodeTest(codeword)   bpsk.py:263-387
    R3  channel.py sumProductDecodeTest(codeword)   channel.py:402-513
    R4  channel.py sumProductDecode(codeword)       channel.py:260-398

    python oracle/gen_golden.py [--jobs 8] [--only R0,R1] [--quick]
"""
import argparse
import ast
import contextlib
import io
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import codes  # noqa: E402

REF = os.environ.get("LDPC_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

RULES = {
    # rule: (file, method, extra positional args, +1 carries bit, iteration cap)
    "R0": ("ldpc.py", "minSumDecode", (2,), 1, 10),
    "R1": ("bpsk.py", "minSumDecode", (), 0, 30),
    "R1c": ("channel.py", "minSumDecode", (), 0, 30),
    "R2": ("bpsk.py", "sumProductDecodeTest", (), 0, 50),
    "R3": ("channel.py", "sumProductDecodeTest", (), 0, 50),
    "R4": ("channel.py", "sumProductDecode", (), 0, 40),
}

# (rule, code, [true Es/N0 dB points], frames per point, frames whose per-iteration trace is kept)
CASES = [
    ("R0", "ccsds512", [-1.0, 0.0, 1.0, 2.5], 32, 2),
    ("R0", "wifi648", [2.0, 6.0], 16, 1),
    ("R0", "nr_bg2_z40", [-1.0, 4.0], 8, 1),
    ("R0", "nr_bg2_z40_r1400", [0.0, 5.0], 8, 1),
    ("R1", "ccsds512", [-1.0, 1.0, 2.5, 4.0], 24, 2),
    ("R1", "wifi648", [-1.0, 1.0, 2.5, 4.0], 24, 2),
    ("R1", "nr_bg2_z40", [-6.0, -4.5, -3.0, -1.0], 10, 1),
    ("R1", "nr_bg2_z40_r1400", [-3.5, -1.0, 1.0], 10, 1),
    ("R1c", "wifi648", [2.5], 4, 0),
    ("R2", "ccsds512", [-1.5, 0.0, 1.0, 3.0], 12, 1),
    ("R2", "wifi648", [-1.5, 0.5, 3.0], 12, 1),
    ("R2", "nr_bg2_z40", [-5.5, -2.0], 6, 1),
    ("R2", "nr_bg2_z40_r1400", [-3.5, 0.0], 6, 1),
    ("R3", "ccsds512", [-1.0, 1.0, 3.0], 12, 1),
    ("R3", "wifi648", [-1.0, 1.0, 3.0], 12, 1),
    ("R3", "nr_bg2_z40", [-6.0, -3.0], 6, 1),
    ("R3", "nr_bg2_z40_r1400", [-4.0, -1.0], 6, 1),
    ("R4", "ccsds512", [-1.0, 1.0, 3.0], 12, 1),
    ("R4", "wifi648", [-1.0, 1.0, 3.0], 12, 1),
    ("R4", "nr_bg2_z40", [-6.0, -3.0], 6, 1),
    ("R4", "nr_bg2_z40_r1400", [-4.0, -1.0], 6, 1),
]

_classes = {}


def reference_class(fname):
    """``LDPCEncoder`` from one reference script, compiled from its class definition alone."""
    if fname in _classes:
        return _classes[fname]
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    path = os.path.join(REF, fname)
    with open(path, "r") as f:
        tree = ast.parse(f.read(), filename=path)
    cls = [node for node in tree.body if isinstance(node, ast.ClassDef) and node.name == "LDPCEncoder"]
    assert len(cls) == 1, path
    module = ast.Module(body=cls, type_ignores=[])
    ns = {"numpy": np, "np": np, "plt": plt, "FRAME_ERROR": None, "modulate": None,
          "pyldpc": None, "time": time, "__name__": "reference_" + fname[:-3]}
    exec(compile(module, path, "exec"), ns)
    base = ns["LDPCEncoder"]

    class Probed(base):
        """Adds no behaviour: records the caller's state at each syndrome check."""

        def isValidCodeword(self, x):
            loc = sys._getframe(1).f_locals
            if "bitNodes" in loc and "numIterations" in loc:
                self._probe.append((int(loc["numIterations"]),
                                    np.array(loc["bitNodes"], dtype=float),
                                    np.array(loc["hardDecisions"], dtype=float)))
            return base.isValidCodeword(self, x)

    _classes[fname] = (Probed, plt)
    return _classes[fname]


def make_inputs(rule, code_name, snr_true, nframes, seed):
    """Channel samples, stored LDPCEncoder.SNR values and transmitted codewords.

    First half of the frames: ``SNR`` attribute 0.0 so the reference's LLR formula multiplies by an
    exact power of two; the samples are pre-scaled by the true 1/(2 sigma^2)... and rounded to
    float32, so float32 and float64 decoders see identical LLRs.  Second half: ``SNR`` attribute =
    true SNR (LLRs are then general float64 values).
    """
    plus_bit = RULES[rule][3]
    cc = codes.load_packaged(code_name)
    G = codes.load_packaged_generator(code_name)
    rng = np.random.default_rng(seed)
    msgs = rng.integers(0, 2, size=(nframes, G.shape[0]))
    cw = codes.encode(G, msgs)
    assert cc.is_codeword(cw).all()
    sigma2 = 1.0 / (2.0 * 10.0 ** (snr_true / 10.0))
    x = (2.0 * cw - 1.0) if plus_bit == 1 else (1.0 - 2.0 * cw)
    y = x + np.sqrt(sigma2) * rng.standard_normal(cw.shape)
    snr_attr = np.zeros(nframes)
    half = nframes // 2
    # exact-scale half: the decoder computes 4*y' (or 2*y' for R2); choose y' = y * (0.5/sigma2)
    y[:half] *= 0.5 / sigma2
    snr_attr[half:] = snr_true
    y = y.astype(np.float32).astype(np.float64)
    return y, snr_attr, cw.astype(np.uint8)


def run_frame(task):
    rule, code_name, y, snr_attr, cw, keep_trace = task
    fname, method, extra, _, cap = RULES[rule]
    cls, plt = reference_class(fname)
    cc = codes.load_packaged(code_name)
    o = object.__new__(cls)
    o.H = cc.dense().astype(int)
    o.m = o.H.shape[0]
    o.n = o.H.shape[1]
    o.SNR = float(snr_attr)
    o.originalEncoded = cw.astype(int)
    o.bitEnergyRatio = None
    o.numIterations = None
    o.messageDecoded = None
    o.BER = 0
    o._probe = []
    with contextlib.redirect_stdout(io.StringIO()):
        ret = getattr(o, method)(y.copy(), *extra)
    plt.close("all")
    n = o.n
    converged = ret is not None
    last_it, last_post, last_hard = o._probe[-1]
    if converged:
        assert last_it == o.numIterations
        posterior, hard = last_post, last_hard
        assert np.array_equal(np.asarray(o.messageDecoded, dtype=float), hard)
    else:
        posterior = np.asarray(o.messageDecoded, dtype=float)
        hard = np.full(n, np.nan)  # filled by the caller's rule below
        assert o.numIterations == cap - 1
    trace = None
    if keep_trace:
        trace = np.stack([p for (_, p, _) in o._probe])
    return dict(ret=np.nan if ret is None else float(ret), iters=int(o.numIterations),
                converged=converged, posterior=posterior, hard=hard, trace=trace,
                checks=len(o._probe))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--only", default="")
    ap.add_argument("--quick", action="store_true", help="2 frames per point (smoke run)")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    os.makedirs(OUT, exist_ok=True)

    tasks, index = [], []
    for ci, (rule, code_name, snrs, nframes, ntrace) in enumerate(CASES):
        if only and rule not in only:
            continue
        if args.quick:
            nframes = 2
        for si, snr in enumerate(snrs):
            y, snr_attr, cw = make_inputs(rule, code_name, snr, nframes, seed=[20250509, ci, si])
            for f in range(nframes):
                keep = si == 0 and (f < ntrace or 0 <= f - nframes // 2 < ntrace)  # first of each half
                tasks.append((rule, code_name, y[f], snr_attr[f], cw[f], keep))
                index.append((ci, si, f))
    print(f"{len(tasks)} frames to decode with the reference on {args.jobs} workers", flush=True)
    t0 = time.time()
    # longest first so the pool drains evenly
    order = sorted(range(len(tasks)), key=lambda i: -codes.load_packaged(tasks[i][1]).num_edges
                   * RULES[tasks[i][0]][4])
    results = [None] * len(tasks)
    with Pool(args.jobs) as pool:
        for done, (i, res) in enumerate(pool.imap_unordered(_indexed, [(i, tasks[i]) for i in order])):
            results[i] = res
            if done % 25 == 0:
                print(f"  {done}/{len(tasks)}  {time.time() - t0:.0f}s", flush=True)

    # regroup per (case, snr)
    groups = {}
    for (ci, si, f), task, res in zip(index, tasks, results):
        groups.setdefault((ci, si), []).append((f, task, res))
    for (ci, si), items in sorted(groups.items()):
        rule, code_name, snrs, _, _ = CASES[ci]
        items.sort(key=lambda t: t[0])
        plus_bit = RULES[rule][3]
        post = np.stack([r["posterior"] for _, _, r in items])
        hard = np.stack([r["hard"] for _, _, r in items])
        conv = np.array([r["converged"] for _, _, r in items])
        # failure path: hard decisions from the final posterior, by the rule's own polarity
        # (ldpc.py:198-201 / bpsk.py:240-243); only used to fill rows the reference left soft
        fail_hard = np.where(post < 0, 0, 1) if plus_bit == 1 else np.where(post > 0, 0, 1)
        hard = np.where(conv[:, None], hard, fail_hard)
        traces = {f"trace_{f}": r["trace"] for f, _, r in items if r["trace"] is not None}
        name = f"{rule}_{code_name}_snr{snrs[si]:+.1f}.npz".replace("+", "p").replace("-", "m")
        np.savez_compressed(
            os.path.join(OUT, name),
            rule=rule, code=code_name, snr_true=snrs[si],
            y=np.stack([t[2] for _, t, _ in items]).astype(np.float32),
            snr_attr=np.array([t[3] for _, t, _ in items]),
            codeword=np.stack([t[4] for _, t, _ in items]),
            ret=np.array([r["ret"] for _, _, r in items]),
            iters=np.array([r["iters"] for _, _, r in items], np.int32),
            converged=conv,
            hard=hard.astype(np.float64),
            posterior=post,
            **traces)
        print(f"{name}: {len(items)} frames, {int(conv.sum())} converged, "
              f"iters {[r['iters'] for _, _, r in items]}", flush=True)
    print(f"done in {time.time() - t0:.0f}s")


def _indexed(arg):
    i, task = arg
    return i, run_frame(task)


if __name__ == "__main__":
    main()
