"""The multi-wave compile-time kernels (several waves share a frame group): which codes take them,
ragged and tiny batches, the fixed-iteration mode and the iteration-0 float syndrome of R0, all
against the CPU restatement.  Min-sum rules in double must agree bit for bit; sum-product within
1e-4 relative on the posteriors."""
import numpy as np
import pytest

from ldpc_amd import codes, config
from oracle import bp_oracle

pytestmark = pytest.mark.gpu

_dec = {}


def decoder_for(name):
    from ldpc_amd.decoder import Decoder
    if name not in _dec:
        _dec[name] = Decoder(codes.load_packaged(name))
    return _dec[name]


def noisy_llrs(name, cfg, B, snr, seed):
    code = codes.load_packaged(name)
    G = codes.load_packaged_generator(name)
    rng = np.random.default_rng([seed, cfg.rule, B])
    cw = codes.encode(G, rng.integers(0, 2, size=(B, G.shape[0])))
    sigma2 = 1.0 / (2.0 * 10 ** (snr / 10))
    x = 2.0 * cw - 1.0 if cfg.plus_is_bit == 1 else 1.0 - 2.0 * cw
    y = x + np.sqrt(sigma2) * rng.standard_normal(x.shape)
    return code, bp_oracle.llr_from_samples(y, snr, cfg.rule)


EXPECT = [  # code, rule, kernel that must serve it in double
    ("wifi648", "R0", "spec_r0w_f64"), ("wifi648", "R1", "spec_r1m_f64"), ("wifi648", "R3", "spec_spaw_f64"),
    ("nr_bg2_z40", "R0", "spec_r0w_f64"), ("nr_bg2_z40", "R1", "spec_r1m_f64"),
    ("nr_bg2_z40", "R2", "spec_spaw_f64"), ("nr_bg2_z40", "R3", "spec_spaw_f64"), ("nr_bg2_z40", "R4", "spec_spaw_f64"),
    ("nr_bg2_z40_r1400", "R0", "spec_r0w_f64"), ("nr_bg2_z40_r1400", "R1", "spec_r1m_f64"),
    ("nr_bg2_z40_r1400", "R3", "spec_spaw_f64"),
]
SNR = {"wifi648": -1.0, "nr_bg2_z40": -6.0, "nr_bg2_z40_r1400": -5.0}


@pytest.mark.parametrize("early", [True, False], ids=["early_exit", "fixed"])
@pytest.mark.parametrize("name,rule,kernel", EXPECT, ids=[f"{n}-{r}" for n, r, _ in EXPECT])
def test_ragged_batches_match_c_oracle(name, rule, kernel, early, c_oracle):
    cfg = config.PRESETS[rule].with_(max_iters=8, early_exit=early)
    dec = decoder_for(name)
    for B in (1, 3, 131):   # fewer frames than slots, an odd count, more groups than one pass of a small grid
        code, L = noisy_llrs(name, cfg, B, SNR[name], 17)
        eh, ei, ec, ep = c_oracle.c_decode(code, cfg, L)
        res = dec.decode(L, cfg, compute_dtype=np.float64, want_posterior=True)
        assert res.kernel == kernel
        if cfg.rule <= 1:
            assert np.array_equal(res.converged, ec) and np.array_equal(res.iters, ei)
            assert np.array_equal(res.hard, eh) and np.array_equal(res.posterior, ep)
        else:
            # device transcendentals against libm: a frame on the edge of converging may take a step more
            # (against numpy, which the reference uses, no frame in 1536 differed: tests/test_gpu_precision.py)
            same = (res.converged == ec) & (res.iters == ei)
            assert (~same).sum() <= (1 if B > 100 else 0), int((~same).sum())
            assert np.array_equal(res.hard[same & ec], eh[same & ec])
            scale = np.maximum(np.abs(ep[same]), 1.0)
            assert float(np.max(np.abs(res.posterior[same] - ep[same]) / scale, initial=0.0)) <= 1e-4


def test_float32_variants_are_served_and_converge():
    """float32 arithmetic on the same kernels: at a comfortable SNR every frame must decode to the
    transmitted word (extrinsic rules)."""
    for name, rule, kernel in [("nr_bg2_z40", "R0", "spec_r0w_f32"), ("nr_bg2_z40", "R1", "spec_r1w_f32"),
                               ("nr_bg2_z40", "R3", "spec_spaw_f32"), ("nr_bg2_z40_r1400", "R1", "spec_r1w_f32"),
                               ("wifi648", "R1", "spec_r1w_f32")]:
        cfg = config.PRESETS[rule]
        code = codes.load_packaged(name)
        G = codes.load_packaged_generator(name)
        rng = np.random.default_rng([23, cfg.rule])
        B = 257
        cw = codes.encode(G, rng.integers(0, 2, size=(B, G.shape[0])))
        snr = 4.0 if name == "wifi648" else 0.0
        sigma2 = 1.0 / (2.0 * 10 ** (snr / 10))
        x = 2.0 * cw - 1.0 if cfg.plus_is_bit == 1 else 1.0 - 2.0 * cw
        L = bp_oracle.llr_from_samples(x + np.sqrt(sigma2) * rng.standard_normal(x.shape), snr, cfg.rule)
        res = decoder_for(name).decode(L.astype(np.float32), cfg, compute_dtype=np.float32)
        assert res.kernel == kernel
        if rule == "R0":
            # the reference's R0 rule never converges on this matrix (its own golden vectors show
            # the same): compare with the double kernel, whose results equal the CPU restatement's
            ref = decoder_for(name).decode(L.astype(np.float32), cfg, compute_dtype=np.float64)
            assert np.array_equal(res.converged, ref.converged) and np.array_equal(res.iters, ref.iters)
            assert (res.hard == ref.hard).mean() >= 0.99
            continue
        assert res.converged.all(), (name, rule, res.converged.mean())
        checked = np.diff(code.col_ptr) > 0   # the truncated matrix leaves some variables in no check
        assert np.array_equal(res.hard[:, checked], cw.astype(np.uint8)[:, checked])


@pytest.mark.parametrize("name,snr", [("nr_bg2_z40", -5.5), ("nr_bg2_z40_r1400", -4.5)])
def test_in_place_kernel_many_frames_per_workgroup(name, snr, c_oracle):
    """The double-precision R1 / OMS kernel with one number per edge (two workgroups per compute unit, every
    workgroup walking many frames): bit-identical to the CPU restatement on a batch across the waterfall,
    with the reference's 30 iterations and in fixed-work mode, posteriors included."""
    for rule in ("R1", "OMS"):
        for cfg in (config.PRESETS[rule], config.PRESETS[rule].with_(max_iters=7, early_exit=False)):
            code, L = noisy_llrs(name, cfg, 4099, snr, 41)
            eh, ei, ec, ep = c_oracle.c_decode(code, cfg, L)
            res = decoder_for(name).decode(L, cfg, compute_dtype=np.float64, want_posterior=True)
            assert res.kernel == ("spec_r1m_f64" if rule == "R1" else "spec_omsm_f64")
            assert np.array_equal(res.iters, ei) and np.array_equal(res.converged, ec)
            assert np.array_equal(res.hard, eh) and np.array_equal(res.posterior, ep)
            if cfg.early_exit:
                assert 0.02 < 1.0 - ec.mean() < 0.9, ec.mean()   # the batch straddles the waterfall


def test_empty_batch():
    cfg = config.PRESETS["R1"]
    code = codes.load_packaged("nr_bg2_z40")
    res = decoder_for("nr_bg2_z40").decode(np.zeros((0, code.n)), cfg)
    assert res.hard.shape == (0, code.n) and res.iters.shape == (0,)


def test_slot_queues_many_frames_per_slot(c_oracle):
    """Early exit with several slots per workgroup: every slot walks its own queue of frames.  With
    far more frames than slots each queue is long; results must land on the right frames."""
    # sum-product in double on 802.11n (two slots): against the CPU restatement
    cfg = config.PRESETS["R3"].with_(max_iters=12)
    code, L = noisy_llrs("wifi648", cfg, 6001, 0.5, 31)
    eh, ei, ec, ep = c_oracle.c_decode(code, cfg, L)
    res = decoder_for("wifi648").decode(L, cfg, compute_dtype=np.float64, want_posterior=True)
    assert res.kernel == "spec_spaw_f64"
    same = (res.converged == ec) & (res.iters == ei)
    assert (~same).sum() <= 6, int((~same).sum())   # one frame in a thousand against libm's transcendentals
    assert np.array_equal(res.hard[same & ec], eh[same & ec])
    scale = np.maximum(np.abs(ep[same]), 1.0)
    assert float(np.max(np.abs(res.posterior[same] - ep[same]) / scale)) <= 1e-4
    assert 0.3 < ec.mean() < 1.0 and ei[ec].min() < ei[ec].max()   # frames really end at different times

    # float32 on the BG2-shaped matrix (three slots over two lane groups): every frame is a different
    # random codeword, so a result written to the wrong frame cannot go unnoticed; iteration counts
    # against the double kernel, which runs one frame per workgroup
    name = "nr_bg2_z40"
    for rule in ("R1", "R3"):
        cfg = config.PRESETS[rule]
        code = codes.load_packaged(name)
        G = codes.load_packaged_generator(name)
        rng = np.random.default_rng([37, cfg.rule])
        B = 20011
        cw = codes.encode(G, rng.integers(0, 2, size=(B, G.shape[0])))
        snr = -4.0
        sigma2 = 1.0 / (2.0 * 10 ** (snr / 10))
        x = 2.0 * cw - 1.0 if cfg.plus_is_bit == 1 else 1.0 - 2.0 * cw
        L = bp_oracle.llr_from_samples(x + np.sqrt(sigma2) * rng.standard_normal(x.shape), snr, cfg.rule).astype(np.float32)
        dec = decoder_for(name)
        r32 = dec.decode(L, cfg, compute_dtype=np.float32)
        r64 = dec.decode(L, cfg, compute_dtype=np.float64)
        assert r32.kernel in ("spec_r1w_f32", "spec_spaw_f32")
        assert r64.converged.mean() > 0.98 and r64.iters.min() < r64.iters.max()
        assert (r32.converged == r64.converged).mean() >= 0.995
        both = r32.converged & r64.converged
        assert np.array_equal(r32.hard[both], r64.hard[both])
        # (a few frames converge to another codeword at this SNR: undetected errors)
        assert (r64.hard[r64.converged] == cw.astype(np.uint8)[r64.converged]).all(axis=1).mean() >= 0.999
        assert (np.abs(r32.iters[both] - r64.iters[both]) <= 1).mean() >= 0.99
