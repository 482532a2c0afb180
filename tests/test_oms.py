"""Offset min-sum (LDPC_RULE_OMS): a build extension with no counterpart in the reference, so parity
is against this build's own two oracles (numpy and C), and its decoding performance is checked
against the reference-pinned normalised rule R1 (bpsk.py:154-260)."""
import numpy as np
import pytest

from ldpc_amd import codes, config
from oracle import bp_oracle


def _batch(code, G, B, snr, rng):
    cw = codes.encode(G, rng.integers(0, 2, size=(B, G.shape[0]))) if G is not None else np.zeros((B, code.n), np.uint8)
    sigma2 = 1.0 / (2.0 * 10 ** (snr / 10))
    y = (1.0 - 2.0 * cw) + np.sqrt(sigma2) * rng.standard_normal(cw.shape)
    return cw, (2.0 * y / sigma2).astype(np.float32).astype(np.float64)


def test_oracles_agree_and_reduce_to_r1():
    bp_oracle.build_c_oracle()
    code = codes.load_packaged("wifi648")
    G = codes.load_packaged_generator("wifi648")
    rng = np.random.default_rng(3)
    cw, L = _batch(code, G, 96, -1.0, rng)
    cfg = config.OMS
    nh, ni, nc, npost = bp_oracle.numpy_decode(code, cfg, L)
    ch, ci, cc, cpost = bp_oracle.c_decode(code, cfg, L)
    assert np.array_equal(nh, ch) and np.array_equal(ni, ci) and np.array_equal(nc, cc) and np.array_equal(npost, cpost)
    assert 0 < cc.sum() < len(cc)
    assert np.array_equal(ch[cc], cw[cc])                       # what converges here is the sent word
    # offset 0 with R1's factor IS R1: max(0.75 m - 0, 0) = 0.75 m
    as_r1 = config.OMS.with_(scale=0.75, offset=0.0)
    a = bp_oracle.c_decode(code, as_r1, L)
    b = bp_oracle.c_decode(code, config.R1.with_(max_iters=as_r1.max_iters), L)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    # a large offset wipes every message out: posteriors stay the channel LLRs
    flat = bp_oracle.c_decode(code, config.OMS.with_(offset=1e9, max_iters=3), L)
    assert np.array_equal(flat[3], L) and not flat[2].all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ccsds512", "wifi648", "nr_bg2_z40", "nr_bg2_z40_r1400"])
def test_gpu_oms_f64_bit_exact_all_kernels(name, c_oracle):
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged(name)
    G = codes.load_packaged_generator(name) if name in ("ccsds512", "wifi648") else None
    rng = np.random.default_rng([6, len(name)])
    snr = {"ccsds512": -0.5, "wifi648": -1.0, "nr_bg2_z40": -5.5, "nr_bg2_z40_r1400": -4.5}[name]
    cw, L = _batch(code, G, 1024 if code.n <= 648 else 384, snr, rng)
    dec = Decoder(code)
    seen = set()
    for cfg in (config.OMS, config.OMS.with_(scale=0.9, offset=0.25, max_iters=12, early_exit=False)):
        eh, ei, ec, ep = c_oracle.c_decode(code, cfg, L)
        for kernel in (0, 1, 2):
            r = dec.decode(L, cfg, compute_dtype=np.float64, want_posterior=True, kernel=kernel)
            seen.add(r.kernel)
            assert np.array_equal(r.converged, ec) and np.array_equal(r.iters, ei), r.kernel
            assert np.array_equal(r.hard, eh) and np.array_equal(r.posterior, ep), r.kernel
        assert 0 < ec.sum() < len(ec)
    assert any(k.startswith("spec_oms") for k in seen) and any(k.startswith("qc") for k in seen) \
        and any(k.startswith("csr") for k in seen), seen


@pytest.mark.gpu
def test_gpu_oms_f32_and_frame_error_rate_against_r1():
    """float32 OMS agrees with float64 on nearly every frame; OMS (beta 0.5 on LLRs 2y/sigma^2) is
    at least as good a decoder as the reference's normalised rule R1 on the 802.11n code: measured
    over 2^17 frames at Es/N0 -0.75 dB and 20 iterations, frame error rate 0.0060 against R1's 0.0191
    (binomial noise is a few percent of either)."""
    import torch
    from ldpc_amd import sweep
    from ldpc_amd.decoder import Decoder
    dev = torch.device("cuda", 0)
    code = codes.load_packaged("wifi648")
    G = codes.load_packaged_generator("wifi648")
    dec = Decoder(code)
    fer = {}
    for rule in ("R1", "OMS"):
        cfg = config.PRESETS[rule].with_(max_iters=20)
        out = sweep.ber_sweep(code, G, cfg, [-0.75], 1 << 17, dev, np.float64, seed=5, dec=dec)[0]
        out32 = sweep.ber_sweep(code, G, cfg, [-0.75], 1 << 17, dev, np.float32, seed=5, dec=dec)[0]
        fer[rule] = out["fer"]
        assert abs(out32["not_converged"] - out["not_converged"]) <= 0.01 * (1 << 17)
        assert out["converged_wrong"] <= 50    # undetected errors: frames that converge to another codeword
    assert 0.002 < fer["R1"] < 0.5 and fer["R1"] / 10 < fer["OMS"] < 1.2 * fer["R1"], fer
