"""Packed half-precision fast mode (compute_dtype LDPC_F16: rule R1, two frames per lane).  It has no
counterpart in the reference, which computes in float64, and is bit-exact with nothing: it is
validated statistically, against the float32 kernel of the reference-pinned rule R1
(bpsk.py:154-260) on the SAME frames - frame and bit error rates inside a band around float32's,
every frame it declares converged is a codeword, and nearly all frames end with the same flag."""
import numpy as np
import pytest

from ldpc_amd import _lib, codes, config

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,snrs", [("ccsds512", [-1.0, -0.25, 0.5]), ("wifi648", [-1.25, -0.75, -0.25])])
def test_half_precision_error_rates_track_float32(name, snrs):
    import torch
    from ldpc_amd import sweep
    from ldpc_amd.decoder import Decoder
    dev = torch.device("cuda", 0)
    code = codes.load_packaged(name)
    G = codes.load_packaged_generator(name)
    dec = Decoder(code)
    frames = (1 << 17) + 3            # odd: the last pair holds one frame
    cfg = config.R1.with_(max_iters=20)
    for si, snr in enumerate(snrs):
        bt = sweep.make_batch(code, G, frames, snr, cfg, seed=[12, si], device=dev, native_noise=True)
        out = {}
        for dt in (np.float32, np.float16):
            k = sweep.decode_batch(dec, bt, cfg, compute_dtype=dt)
            bt.counters.zero_()
            sweep.count_errors(dec, bt)
            torch.cuda.synchronize()
            out[dt] = (sweep.summarize(bt.counters.cpu().numpy(), code.n), bt.converged.clone(), bt.hard.clone(), bt.iters.clone(), k)
        s32, c32, h32, i32, k32 = out[np.float32]
        s16, c16, h16, i16, k16 = out[np.float16]
        assert k16 == "spec_r1_f16x2" and k32.startswith("spec_r1")
        assert s16["frames"] == frames
        # every frame the half-precision kernel calls converged IS a codeword: its syndrome is exact
        conv = c16.bool()
        some = torch.nonzero(conv)[:200, 0].cpu().numpy()
        assert code.is_codeword(h16[some].cpu().numpy()).all()
        # error rates: within 15 % of float32's plus binomial noise
        slack = 4 * np.sqrt(max(s32["not_converged"], 1))
        assert abs(s16["not_converged"] - s32["not_converged"]) <= 0.15 * s32["not_converged"] + slack, (snr, s16, s32)
        assert s16["ber"] <= 1.2 * s32["ber"] + 1e-6
        assert s16["converged_wrong"] <= s32["converged_wrong"] + 20
        # frame by frame: the flag agrees on all but a few percent of the frames near the waterfall
        assert (c16 != c32).float().mean().item() <= 0.02
        both = conv & c32.bool()
        assert (h16[both] != h32[both]).any(dim=1).float().mean().item() <= 1e-3
        assert abs(s16["mean_iters"] - s32["mean_iters"]) <= 0.3
        if si == 1:
            assert 0.005 < s32["fer"] < 0.6


def test_half_precision_edges():
    import torch
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged("ccsds512")
    dec = Decoder(code)
    # noiseless all-zero word, one frame: converged at the first test
    r = dec.decode(np.full((1, code.n), 4.0, np.float32), config.R1, compute_dtype=np.float16)
    assert r.kernel == "spec_r1_f16x2" and r.converged[0] and r.iters[0] == 0 and not r.hard.any()
    # fixed work: exactly max_iters updates, the flag is the final syndrome
    rng = np.random.default_rng(0)
    L = (8.0 + 4.0 * rng.standard_normal((77, code.n))).astype(np.float32)     # LLRs of a 3 dB channel
    r = dec.decode(L, config.R1.with_(max_iters=7, early_exit=False), compute_dtype=np.float16)
    assert (r.iters == 7).all() and r.converged.mean() > 0.9
    assert code.is_codeword(r.hard[r.converged]).all()
    r32 = dec.decode(L, config.R1.with_(max_iters=7, early_exit=False), compute_dtype=np.float32)
    assert abs(int(r.converged.sum()) - int(r32.converged.sum())) <= 3
    # huge channel values are clamped, not turned into infinities
    r = dec.decode(np.full((3, code.n), 1e30, np.float32), config.R1, compute_dtype=np.float16)
    assert r.converged.all() and not r.hard.any()
    # what the mode does not serve is refused, not silently decoded in another arithmetic
    for kw, cfg in ((dict(want_posterior=True), config.R1), ({}, config.R0), ({}, config.R3), (dict(kernel=1), config.R1)):
        with pytest.raises(_lib.LdpcError) as ei:
            dec.decode(L, cfg, compute_dtype=np.float16, **kw)
        assert ei.value.code == _lib.LDPC_ERR_UNSUPPORTED
    big = Decoder(codes.load_packaged("nr_bg2_z40"))
    with pytest.raises(_lib.LdpcError):
        big.decode(np.zeros((2, 2080), np.float32), config.R1, compute_dtype=np.float16)
