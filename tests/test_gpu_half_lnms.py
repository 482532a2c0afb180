"""Row-layered min-sum in packed half precision (LDPC_RULE_LNMS with compute_dtype LDPC_F16: two
frames per lane, half-precision posteriors, check magnitudes kept as eight bits).  Neither the rule
nor the arithmetic exists in the reference (flooding schedules in float64 only): the mode is
bit-exact with nothing and is validated statistically, against the float32 layered kernel on the
SAME frames - frame and bit error rates inside a band around float32's, every frame declared
converged is a codeword, nearly all frames end with the same flag and, where both converge, with
the same word.  The float32 kernel itself is tied to this build's CPU restatement bit for bit
(tests/test_layered.py) and to the reference-pinned flooding rule statistically
(tests/test_gpu_ber.py)."""
import numpy as np
import pytest

from ldpc_amd import _lib, codes, config

pytestmark = pytest.mark.gpu


def clopper_pearson(k, n, conf=0.999):
    from scipy.stats import beta
    a = 1 - conf
    lo = 0.0 if k == 0 else beta.ppf(a / 2, k, n - k + 1)
    hi = 1.0 if k == n else beta.ppf(1 - a / 2, k + 1, n - k)
    return lo, hi

CASES = [("nr_bg1_z384", 1 << 13, [-4.0, -3.75, -3.0]),
         ("nr_bg2_z40", 1 << 16, [-6.0, -5.0, -3.0]),
         ("nr_bg2_z40_r1400", 1 << 16, [-5.0, -4.5, -3.0]),
         ("wifi648", 1 << 17, [-1.5, -1.0, 0.0])]


@pytest.mark.parametrize("name,frames,snrs", CASES)
@pytest.mark.parametrize("early", [True, False])
def test_half_precision_layered_error_rates_track_float32(name, frames, snrs, early):
    import torch
    from ldpc_amd import sweep
    from ldpc_amd.decoder import Decoder
    dev = torch.device("cuda", 0)
    code = codes.load_packaged(name)
    G = codes.load_packaged_generator(name) if name in ("wifi648", "nr_bg1_z384") else None
    dec = Decoder(code)
    frames += 3                       # odd: the last pair holds one frame, the last workgroup is ragged
    cfg = config.LNMS.with_(max_iters=10, early_exit=early)
    for si, snr in enumerate(snrs):
        bt = sweep.make_batch(code, G, frames, snr, cfg, seed=[21, si], device=dev, native_noise=True)
        out = {}
        for dt in (np.float32, np.float16):
            bt.hard.fill_(7)
            bt.iters.fill_(-5)
            k = sweep.decode_batch(dec, bt, cfg, compute_dtype=dt)
            bt.counters.zero_()
            sweep.count_errors(dec, bt)
            torch.cuda.synchronize()
            out[dt] = (sweep.summarize(bt.counters.cpu().numpy(), code.n), bt.converged.clone(), bt.hard.clone(), bt.iters.clone(), k)
        s32, c32, h32, i32, k32 = out[np.float32]
        s16, c16, h16, i16, k16 = out[np.float16]
        assert k16 == "spec_lnms_f16x2" and k32 == "spec_lnms_f32"
        assert s16["frames"] == frames
        assert int(h16.max()) <= 1 and int(i16.min()) >= 0          # every frame of every pair was written
        if not early:
            assert (i16 == 10).all()
        # every frame the half-precision kernel calls converged IS a codeword: its syndrome is exact
        conv = c16.bool()
        some = torch.nonzero(conv)[:200, 0].cpu().numpy()
        if len(some):
            assert code.is_codeword(h16[some].cpu().numpy()).all()
        # error rates: within 15 % of float32's plus binomial noise
        slack = 5 * np.sqrt(max(s32["not_converged"], 1))
        assert abs(s16["not_converged"] - s32["not_converged"]) <= 0.15 * s32["not_converged"] + slack, (snr, s16, s32)
        assert s16["ber"] <= 1.2 * s32["ber"] + 1e-6, (snr, s16["ber"], s32["ber"])
        assert s16["converged_wrong"] <= 1.05 * s32["converged_wrong"] + 30
        # frame by frame: flags agree except near the waterfall, converged frames decode to the same word
        assert (c16 != c32).float().mean().item() <= 0.16, snr
        both = conv & c32.bool()
        if both.any():
            assert (h16[both] != h32[both]).any(dim=1).float().mean().item() <= 3e-3
        assert abs(s16["mean_iters"] - s32["mean_iters"]) <= 0.3


def test_half_precision_layered_edges():
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged("nr_bg2_z40")
    dec = Decoder(code)
    # noiseless all-zero word, one frame (half a pair): converged at the first test
    r = dec.decode(np.full((1, code.n), 4.0, np.float32), config.LNMS, compute_dtype=np.float16)
    assert r.kernel == "spec_lnms_f16x2" and r.converged[0] and r.iters[0] == 0 and not r.hard.any()
    # huge channel values are clamped, not turned into infinities
    r = dec.decode(np.full((3, code.n), 1e30, np.float32), config.LNMS, compute_dtype=np.float16)
    assert r.converged.all() and not r.hard.any()
    # the two frames of a pair do not see each other: decoding frame by frame gives the batch's results
    rng = np.random.default_rng(5)
    L = (2.0 * 10 ** (-0.45) * (1.0 + 10 ** 0.225 * rng.standard_normal((9, code.n)))).astype(np.float32)
    cfg = config.LNMS.with_(max_iters=10)
    whole = dec.decode(L, cfg, compute_dtype=np.float16)
    for i in range(L.shape[0]):
        one = dec.decode(L[i:i + 1], cfg, compute_dtype=np.float16)
        assert np.array_equal(one.hard[0], whole.hard[i]) and one.iters[0] == whole.iters[i] and one.converged[0] == whole.converged[i]
    # determinism
    again = dec.decode(L, cfg, compute_dtype=np.float16)
    assert np.array_equal(again.hard, whole.hard) and np.array_equal(again.iters, whole.iters)
    # what the mode does not serve is refused, not silently decoded in another arithmetic
    for kw, c in ((dict(want_posterior=True), cfg), (dict(kernel=2), cfg), ({}, config.R0)):
        with pytest.raises(_lib.LdpcError) as ei:
            dec.decode(L, c, compute_dtype=np.float16, **kw)
        assert ei.value.code == _lib.LDPC_ERR_UNSUPPORTED
    # double-circulant blocks (CCSDS) have no layered kernel in any arithmetic
    with pytest.raises(_lib.LdpcError):
        Decoder(codes.load_packaged("ccsds512")).decode(np.zeros((2, 512), np.float32), cfg, compute_dtype=np.float16)


@pytest.mark.parametrize("name,snrs", [("wifi648", [-1.5, -1.0, -0.5, 0.0]), ("nr_bg2_z40", [-6.5, -6.0, -5.5, -5.0])])
def test_half_precision_layered_anchored_to_reference_pinned_flooding_rule(name, snrs):
    """The same anchor as the float layered kernel has (tests/test_gpu_ber.py), applied to the packed
    half-precision mode directly: on the reference's own matrices, same frames, ten sweeps in packed
    half precision are never worse than ten iterations of the reference-pinned flooding rule R1
    (bpsk.py:154-260, float64) and lie between R1 at twenty iterations and 1.8 times that."""
    import torch
    from ldpc_amd import sweep
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged(name)
    G = codes.load_packaged_generator(name) if name == "wifi648" else None
    dev = torch.device("cuda", 0)
    dec = Decoder(code)
    frames = 1 << 17
    run = lambda cfg, dt: sweep.ber_sweep(code, G, cfg, snrs, frames, dev, dt, seed=17, dec=dec, fused=False)
    r10, r20 = run(config.R1.with_(max_iters=10), np.float64), run(config.R1.with_(max_iters=20), np.float64)
    h10 = run(config.LNMS.with_(max_iters=10), np.float16)
    for a10, a20, b10 in zip(r10, r20, h10):
        lo20, hi20 = clopper_pearson(a20["not_converged"], frames)
        assert b10["frames"] == frames
        assert b10["not_converged"] <= a10["not_converged"], (name, a10["es_n0_db"])
        assert 0.9 * lo20 <= b10["fer"] <= 1.8 * hi20, (name, a10["es_n0_db"], b10["fer"], lo20, hi20)
        assert b10["ber"] <= a10["ber"] and b10["ber"] <= 1.8 * a20["ber"] + 1e-6
