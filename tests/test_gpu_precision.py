"""How far float32 decoding is from float64 decoding, and the device's float64 sum-product kernels
from the numpy restatement that is bit-identical to the reference: measured on >= 10^5 frames per
(rule, code) for the short codes (32 768 for the 5G matrices, 8 192 for BG1) at a waterfall SNR, and
bounded at about three times the measured rates.

What the bounds say: with float64 arithmetic the min-sum rules reproduce the reference bit for bit
(golden vectors, tests/test_gpu_parity.py) and the sum-product rules reproduce its decisions and
iteration counts.  With float32 arithmetic NO rule is bit-exact, but two frames decoded to a word by
both arithmetic types never got different words, fewer than one frame in a thousand changes its
iteration count or its convergence flag under the min-sum and tanh rules (one in seventy under the
phi rule R4), and posteriors of frames that converge within six iterations agree to 1e-4 under the
min-sum rules.  Posteriors of the sum-product rules are NOT comparable in float32: the product clip
0.999999999 / 0.99999999 (bpsk.py:337, channel.py:473) is not a float32 value, so messages saturate
at 17.3 instead of 21.4 / 19.1."""
import pytest

pytestmark = pytest.mark.gpu

#: rule -> (iteration-count disagreement rate, convergence-flag disagreement rate) allowed, f32 vs f64
BOUNDS = {"R0": (1e-3, 5e-4), "R1": (1e-3, 5e-4), "OMS": (1e-3, 5e-4), "LNMS": (1e-3, 5e-4),
          "R2": (1e-3, 5e-4), "R3": (1e-3, 5e-4), "R4": (1.5e-2, 8e-3)}


def test_float32_and_device_transcendental_disagreement_rates():
    import precision_rates
    res = precision_rates.measure()
    assert len(res) == len(precision_rates.CASES)
    for key, r in res.items():
        rule = key.split("/")[0]
        n = r["frames"]
        it_bound, conv_bound = BOUNDS[rule]
        assert 0.005 < r["fer_f64"] < 0.6, (key, r["fer_f64"])            # the batch straddles the waterfall
        assert r["iters_diff"] <= max(2, it_bound * n), (key, r["iters_diff"], n)
        assert r["conv_diff"] <= max(2, conv_bound * n), (key, r["conv_diff"], n)
        assert r["hard_diff"] == 0, key                                    # converged alike -> the same word
        if rule in ("R0", "R1", "OMS", "LNMS"):
            quick = [v for k, v in r["max_rel_posterior_error_by_iters"].items() if int(k) <= 6]
            if rule == "LNMS" and "bg2" in key:
                assert max(quick) <= 5e-3, (key, max(quick))               # 23 messages per variable: rounding adds up
            else:
                assert max(quick) <= 1e-4, (key, max(quick))
        if "f64_vs_numpy" in r:
            v = r["f64_vs_numpy"]
            assert v["iters_diff"] == 0 and v["conv_diff"] == 0 and v["hard_diff"] == 0, (key, v)
            assert v["max_rel_posterior"] <= (1e-3 if rule == "R4" else 1e-4), (key, v)


def test_bg1_layered_float_against_double_on_131072_frames():
    """The BG1 Z = 384 half of the metric is quoted in float32 (the layered rule has no reference
    counterpart whose float64 it would have to reproduce; in double the kernel holds one frame per
    compute unit instead of two).  What that costs in results, on 2^17 frames at the waterfall
    (Es/N0 -3.75 dB): the two arithmetic types never decode a frame to two different words, and at most
    a handful of frames change their convergence flag or sweep count."""
    import numpy as np
    import torch

    import precision_rates
    from ldpc_amd import codes, config, sweep
    from ldpc_amd.decoder import Decoder
    dev = torch.device("cuda", 0)
    code = codes.load_packaged("nr_bg1_z384")
    G = codes.load_packaged_generator("nr_bg1_z384")
    dec = Decoder(code)
    cfg = config.LNMS
    frames, step = 1 << 17, 8192
    tot = dict(conv_diff=0, iters_diff=0, hard_diff=0, conv64=0, wrong=0)
    for c in range(0, frames, step):
        batch = sweep.make_batch(code, G, step, -3.75, cfg, seed=[47, c], device=dev, native_noise=True, first_frame=c)
        res = {}
        for dt in (np.float64, np.float32):
            kern = sweep.decode_batch(dec, batch, cfg, compute_dtype=dt)
            torch.cuda.synchronize()
            res[dt] = (batch.hard.clone(), batch.iters.clone(), batch.converged.clone())
        assert kern == "spec_lnms_f32"
        (h64, i64, c64), (h32, i32, c32) = res[np.float64], res[np.float32]
        tot["conv_diff"] += int((c64 != c32).sum())
        tot["iters_diff"] += int((i64 != i32).sum())
        tot["hard_diff"] += int(((h64 != h32).any(dim=1) & (c64 != 0) & (c32 != 0)).sum())
        tot["conv64"] += int(c64.sum())
        tot["wrong"] += int(((h64 != batch.ref).any(dim=1) & (c64 != 0)).sum())
        del batch
    fer = 1 - tot["conv64"] / frames
    print("BG1 float32 against float64,", frames, "frames:", tot, "FER", fer)
    assert 0.005 < fer < 0.3, fer
    assert tot["hard_diff"] == 0 and tot["wrong"] == 0
    assert tot["conv_diff"] <= 3 and tot["iters_diff"] <= 16, tot
