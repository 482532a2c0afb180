"""ldpc_sweep_batch: channel -> decode -> error accounting in one launch (the body of the
reference's Monte-Carlo loop, /root/reference/ldpc.py:644-667) against the same three steps as
three launches (ldpc_channel_llr, ldpc_decode_batch, ldpc_count_errors) on the same seed: the five
counters must be equal, not close - both paths draw the same noise and run the same arithmetic."""
import numpy as np
import pytest

from ldpc_amd import _lib, codes, config

pytestmark = pytest.mark.gpu

#: (code, rule, Es/N0 dB, frames): operating points where some frames fail and some converge
CASES = [("ccsds512", "R0", 0.5, 40_000), ("ccsds512", "R1", -0.75, 40_000), ("ccsds512", "OMS", -0.75, 20_000),
         ("wifi648", "R0", 6.0, 20_001), ("wifi648", "R1", -1.0, 20_001), ("nr_bg2_z40", "LNMS", -6.0, 3_001),
         ("nr_bg2_z40_r1400", "LNMS", -4.5, 3_000), ("nr_bg1_z384", "LNMS", -3.75, 1_025), ("wifi648", "LNMS", -1.0, 10_000),
         # the multi-wave flooding kernels (three Z = 40 frames per workgroup; LLRs staged in a scratch slice)
         ("nr_bg2_z40", "R1", -5.5, 6_001), ("nr_bg2_z40_r1400", "R1", -4.0, 6_000), ("nr_bg2_z40", "R0", -2.0, 4_001),
         ("nr_bg2_z40", "OMS", -5.5, 3_002)]


def _both(code, G, cfg, snr, frames, dtype, dev, **kw):
    import torch
    from ldpc_amd import sweep
    from ldpc_amd.decoder import Decoder
    dec = Decoder(code)
    cw = sweep.make_codewords(code, G, frames, [21], dev)
    fused = torch.zeros((5,), dtype=torch.int64, device=dev)
    kernel = sweep.sweep_fused(dec, cw, snr, cfg, [22], fused, compute_dtype=dtype, first_frame=7, **kw)
    bt = sweep.noisy_batch(code, cw, snr, cfg, [22], dev, first_frame=7, **kw)
    sweep.decode_batch(dec, bt, cfg, compute_dtype=dtype)
    sweep.count_errors(dec, bt)
    torch.cuda.synchronize()
    return kernel, fused.cpu().numpy(), bt.counters.cpu().numpy()


@pytest.mark.parametrize("name,rule,snr,frames", CASES, ids=[f"{c[0]}-{c[1]}" for c in CASES])
def test_fused_counters_equal_three_launches(name, rule, snr, frames):
    import torch
    dev = torch.device("cuda", 0)
    code = codes.load_packaged(name)
    G = codes.load_packaged_generator(name) if name in ("ccsds512", "wifi648") else None
    for dtype in (np.float64, np.float32):
        for cfg in (config.PRESETS[rule], config.PRESETS[rule].with_(max_iters=6, early_exit=False)):
            kernel, fused, three = _both(code, G, cfg, snr, frames, dtype, dev)
            assert kernel.startswith("fused_spec_"), kernel
            assert fused.tolist() == three.tolist(), (kernel, cfg.early_exit, fused, three)
            assert fused[0] == frames
            if name.startswith("nr_bg2") and rule != "LNMS":
                assert kernel.startswith(("fused_spec_r0w_", "fused_spec_r1w_", "fused_spec_omsw_", "fused_spec_r1m_", "fused_spec_omsm_")), kernel
            if cfg.early_exit and rule != "R0":   # R0 is a weak rule: it never converges on the 802.11n code
                assert 0 < fused[1] < frames, fused        # the batch straddles the waterfall


#: sum-product rules (code, rule, Es/N0 dB, frames, kernel family): the single-wave kernel on the small codes, the
#: multi-wave one on 802.11n in double and on the 5G matrices
SPA_CASES = [("ccsds512", "R2", -0.5, 20_001, "spa"), ("ccsds512", "R3", -0.5, 20_000, "spa"), ("ccsds512", "R4", -0.5, 10_001, "spa"),
             ("wifi648", "R3", -1.0, 10_001, "spa/spaw"), ("wifi648", "R2", -1.0, 5_000, "spa/spaw"),
             ("nr_bg2_z40", "R3", -6.0, 3_001, "spaw"), ("nr_bg2_z40", "R2", -6.0, 2_000, "spaw"),
             ("nr_bg2_z40", "R4", -6.0, 1_501, "spaw"), ("nr_bg2_z40_r1400", "R3", -4.5, 2_002, "spaw")]


@pytest.mark.parametrize("name,rule,snr,frames,family", SPA_CASES, ids=[f"{c[0]}-{c[1]}" for c in SPA_CASES])
def test_fused_sum_product_counters_equal_three_launches(name, rule, snr, frames, family):
    """Rules R2, R3, R4 (bpsk.py:263-387, channel.py:402-513, 260-398) with their posterior clips: one launch
    against three, both arithmetic types, early exit and fixed work, with and without a carrier offset
    (the channel of the mixed CCSDS + 5G sum-product workload)."""
    import torch
    from ldpc_amd import channel
    dev = torch.device("cuda", 0)
    code = codes.load_packaged(name)
    G = codes.load_packaged_generator(name) if name in ("ccsds512", "wifi648") else None
    doppler = dict(cfo=channel.doppler_cfo(4000.0, compensated=0.999), phase0=0.05)
    for dtype in (np.float64, np.float32):
        for cfg, kw in ((config.PRESETS[rule].with_(max_iters=12), {}), (config.PRESETS[rule].with_(max_iters=5, early_exit=False), {}),
                        (config.PRESETS[rule].with_(max_iters=12), doppler)):
            n_fr = frames if dtype is np.float32 or not kw else frames // 2
            kernel, fused, three = _both(code, G, cfg, snr, n_fr, dtype, dev, **kw)
            want = {"spa": ("fused_spec_spa_",), "spaw": ("fused_spec_spaw_",), "spa/spaw": ("fused_spec_spa_", "fused_spec_spaw_")}[family]
            assert kernel.startswith(want), kernel
            assert kernel.endswith("f64" if dtype is np.float64 else "f32"), kernel
            assert fused.tolist() == three.tolist(), (kernel, cfg.early_exit, kw, fused, three)
            assert fused[0] == n_fr
            if cfg.early_exit:
                assert 0 < fused[1] < n_fr, fused        # the batch straddles the waterfall


def test_fused_sum_product_packed_words_and_ragged_tail():
    """Packed reference words and frame counts that leave slots of the last workgroup idle."""
    import torch
    from ldpc_amd import sweep
    from ldpc_amd.decoder import Decoder
    dev = torch.device("cuda", 0)
    for name, frames in (("ccsds512", 1), ("ccsds512", 257), ("nr_bg2_z40", 1), ("nr_bg2_z40", 1_000)):
        code = codes.load_packaged(name)
        G = codes.load_packaged_generator(name) if name == "ccsds512" else None
        dec = Decoder(code)
        cfg = config.R3.with_(max_iters=10)
        cw = sweep.make_codewords(code, G, frames, [61], dev)
        a = torch.zeros((5,), dtype=torch.int64, device=dev)
        b = torch.zeros((5,), dtype=torch.int64, device=dev)
        snr = -0.5 if name == "ccsds512" else -6.0
        ka = sweep.sweep_fused(dec, cw, snr, cfg, [62], a, compute_dtype=np.float32, first_frame=11)
        kb = sweep.sweep_fused(dec, sweep.pack_bits(cw), snr, cfg, [62], b, compute_dtype=np.float32, first_frame=11, packed_n=code.n)
        bt = sweep.noisy_batch(code, cw, snr, cfg, [62], dev, first_frame=11)
        sweep.decode_batch(dec, bt, cfg, compute_dtype=np.float32)
        sweep.count_errors(dec, bt)
        torch.cuda.synchronize()
        assert ka == kb and ka.startswith("fused_spec_spa"), (ka, kb)
        assert a.cpu().numpy().tolist() == b.cpu().numpy().tolist() == bt.counters.cpu().numpy().tolist(), (name, frames, a, b, bt.counters)
        assert a[0] == frames


@pytest.mark.parametrize("frames", [40_001, 7])
def test_fused_half_precision_counters_equal_three_launches(frames):
    """Rule R1 in packed half precision (two frames per lane): the fused launch stages both frames of
    a lane's pair; counters equal the three launches, odd frame counts included."""
    import torch
    dev = torch.device("cuda", 0)
    code = codes.load_packaged("ccsds512")
    G = codes.load_packaged_generator("ccsds512")
    for cfg in (config.R1, config.R1.with_(max_iters=6, early_exit=False)):
        kernel, fused, three = _both(code, G, cfg, -0.75, frames, np.float16, dev)
        assert kernel == "fused_spec_r1_f16x2", kernel
        assert fused.tolist() == three.tolist(), (cfg.early_exit, fused, three)
        assert fused[0] == frames
        if cfg.early_exit and frames > 1000:
            assert 0 < fused[1] < frames, fused


@pytest.mark.parametrize("name,snr,frames", [("nr_bg1_z384", -3.75, 1_027), ("nr_bg2_z40", -6.0, 3_001),
                                             ("nr_bg2_z40_r1400", -4.5, 2_999), ("wifi648", -1.0, 10_003), ("nr_bg1_z384", -3.75, 3)])
def test_fused_half_precision_layered_counters_equal_three_launches(name, snr, frames):
    """The layered rule in packed half precision: the workgroup stages the LLRs of its frame pairs in a
    scratch slice and counts a frame's errors at the test it passes (its partner sweeps on); counters
    equal the three launches, odd frame counts and all-zero words included."""
    import torch
    dev = torch.device("cuda", 0)
    code = codes.load_packaged(name)
    G = codes.load_packaged_generator(name) if name == "wifi648" else None
    for cfg in (config.LNMS, config.LNMS.with_(max_iters=6, early_exit=False)):
        kernel, fused, three = _both(code, G, cfg, snr, frames, np.float16, dev)
        assert kernel == "fused_spec_lnms_f16x2", kernel
        assert fused.tolist() == three.tolist(), (cfg.early_exit, fused, three)
        assert fused[0] == frames
        if cfg.early_exit and frames > 1000:
            assert 0 < fused[1] < frames, fused


@pytest.mark.parametrize("name,rule,dtype,snr,frames", [
    ("ccsds512", "R0", np.float64, 0.5, 20_001), ("ccsds512", "R1", np.float16, -0.75, 20_001),
    ("wifi648", "R1", np.float32, -1.0, 10_001), ("nr_bg2_z40", "R1", np.float64, -5.5, 3_001),
    ("nr_bg2_z40", "R1", np.float32, -5.5, 3_001), ("nr_bg2_z40", "R0", np.float64, -2.0, 2_001),
    ("nr_bg1_z384", "LNMS", np.float32, -3.75, 515), ("nr_bg1_z384", "LNMS", np.float16, -3.75, 515),
    ("wifi648", "LNMS", np.float64, -1.0, 5_001)])
def test_fused_with_packed_reference_words(name, rule, dtype, snr, frames):
    """Transmitted words as bits (ldpc_sweep_batch_packed, n / 8 bytes per frame): the counters of the
    byte-per-bit call on the same words, through every kind of fused kernel; and one packed word for all
    frames against the same word repeated."""
    import torch
    from ldpc_amd import sweep
    from ldpc_amd.decoder import Decoder
    dev = torch.device("cuda", 0)
    code = codes.load_packaged(name)
    G = codes.load_packaged_generator(name) if name in ("ccsds512", "wifi648", "nr_bg1_z384") else None
    dec = Decoder(code)
    cfg = config.PRESETS[rule]
    cw = sweep.make_codewords(code, G, frames, [51], dev)
    bits = sweep.pack_bits(cw)
    assert bits.shape == (frames, (code.n + 7) // 8)
    want = np.packbits(cw[:64].cpu().numpy(), axis=1, bitorder="little")
    assert np.array_equal(bits[:64].cpu().numpy(), want)
    a = torch.zeros((5,), dtype=torch.int64, device=dev)
    b = torch.zeros((5,), dtype=torch.int64, device=dev)
    ka = sweep.sweep_fused(dec, cw, snr, cfg, [52], a, compute_dtype=dtype, first_frame=3)
    kb = sweep.sweep_fused(dec, bits, snr, cfg, [52], b, compute_dtype=dtype, first_frame=3, packed_n=code.n)
    torch.cuda.synchronize()
    assert ka == kb and ka.startswith("fused_spec_")
    assert a.cpu().numpy().tolist() == b.cpu().numpy().tolist() and a[0] == frames and 0 < a[1] <= frames, (a, b)   # (R0 never converges on the 5G matrix)
    # one word for all frames: stride 0
    one = cw[:1].contiguous()
    rep = one.expand(1000, code.n).contiguous()
    c = torch.zeros((5,), dtype=torch.int64, device=dev)
    d = torch.zeros((5,), dtype=torch.int64, device=dev)
    sweep.sweep_fused(dec, rep, snr, cfg, [53], c, compute_dtype=dtype)
    import ctypes
    from ldpc_amd import channel
    from ldpc_amd.config import CDecodeConfig
    sigma2, gain, plen, pval = channel.channel_parameters(snr, cfg)
    key = int(np.random.SeedSequence([53]).generate_state(1, dtype=np.uint64)[0])
    ch = _lib.CChannelConfig(sigma2, gain, pval, 0.0, 0.0, key, 0, int(cfg.plus_is_bit == 1), plen)
    ccfg = CDecodeConfig.from_config(cfg)
    onebits = sweep.pack_bits(one)
    _lib.check(_lib.load().ldpc_sweep_batch_packed(dec._h, ctypes.byref(ccfg), ctypes.byref(ch), onebits.data_ptr(), 0, 1000,
                                                   {np.float64: 1, np.float32: 0, np.float16: 2}[dtype], d.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream or None))
    torch.cuda.synchronize()
    assert c.cpu().numpy().tolist() == d.cpu().numpy().tolist() and c[0] == 1000


def test_pack_bits_ragged_word_length():
    """Words whose length is not a multiple of eight, bytes whose upper bits are set: the low bit counts,
    the unused bits of the last byte are zero."""
    import ctypes
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    # rows 3 bytes apart from dense (a byte at a time), and rows a multiple of eight bytes apart (eight positions per load,
    # the last byte of a ragged word a byte at a time)
    for frames, n, stride in ((1, 1, 4), (5, 13, 16), (300, 1001, 1004), (7, 1001, 1008), (64, 520, 520), (3, 8, 8), (9, 2077, 2080)):
        raw = rng.integers(0, 256, size=(frames, stride), dtype=np.uint8)
        src = torch.as_tensor(raw, device=dev)
        out = torch.full((frames, (n + 7) // 8 + 2), 0xAA, dtype=torch.uint8, device=dev)
        _lib.check(_lib.load().ldpc_pack_bits(src.data_ptr(), stride, frames, n, out.data_ptr(), out.shape[1],
                                              torch.cuda.current_stream().cuda_stream or None))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        want = np.packbits(raw[:, :n] & 1, axis=1, bitorder="little")
        assert np.array_equal(got[:, :want.shape[1]], want)
        assert (got[:, want.shape[1]:] == 0xAA).all()                            # nothing written beyond a word
    with pytest.raises(_lib.LdpcError):
        _lib.check(_lib.load().ldpc_pack_bits(src.data_ptr(), 5, 2, 8, out.data_ptr(), 1, None))   # stride shorter than a word


def test_fused_with_prefix_doppler_and_all_zero_word():
    import torch
    from ldpc_amd import channel, sweep
    from ldpc_amd.decoder import Decoder
    dev = torch.device("cuda", 0)
    code = codes.load_packaged("ccsds512")
    G = codes.load_packaged_generator("ccsds512")
    # the reference's overwritten prefix (bpsk.py:163) and a carrier offset
    for kw in (dict(compat_prefix=True), dict(cfo=channel.doppler_cfo(4000.0, compensated=0.999), phase0=0.05)):
        kernel, fused, three = _both(code, G, config.R1, 1.0, 9_999, np.float64, dev, **kw)
        assert fused.tolist() == three.tolist(), (kw, fused, three)
    # all-zero codeword: ref_dev = NULL against an explicit array of zeros
    dec = Decoder(code)
    zeros = torch.zeros((5000, code.n), dtype=torch.uint8, device=dev)
    a = torch.zeros((5,), dtype=torch.int64, device=dev)
    b = torch.zeros((5,), dtype=torch.int64, device=dev)
    sweep.sweep_fused(dec, zeros, 0.0, config.R1, [3], a, compute_dtype=np.float32)
    import ctypes
    from ldpc_amd.config import CDecodeConfig
    sigma2, gain, plen, pval = channel.channel_parameters(0.0, config.R1)
    key = int(np.random.SeedSequence([3]).generate_state(1, dtype=np.uint64)[0])
    ch = _lib.CChannelConfig(sigma2, gain, pval, 0.0, 0.0, key, 0, 0, plen)
    ccfg = CDecodeConfig.from_config(config.R1)
    _lib.check(_lib.load().ldpc_sweep_batch(dec._h, ctypes.byref(ccfg), ctypes.byref(ch), None, 0, 5000, 0,
                                            b.data_ptr(), torch.cuda.current_stream().cuda_stream or None))
    torch.cuda.synchronize()
    assert a.cpu().numpy().tolist() == b.cpu().numpy().tolist() and a[0] == 5000 and a[1] > 0


def test_ber_sweep_fused_equals_unfused_and_unsupported_falls_back():
    import torch
    from ldpc_amd import sweep
    dev = torch.device("cuda", 0)
    code = codes.load_packaged("ccsds512")
    G = codes.load_packaged_generator("ccsds512")
    snrs = [0.0, 1.0]
    a = sweep.ber_sweep(code, G, config.R0, snrs, 100_000, dev, np.float64, seed=4, fused=True)
    b = sweep.ber_sweep(code, G, config.R0, snrs, 100_000, dev, np.float64, seed=4, fused=False)
    assert a == b
    # the sum-product rules through the sweep: fused by default, same counters as three launches
    c = sweep.ber_sweep(code, G, config.R3, [1.0], 5_000, dev, np.float32, seed=4)
    d = sweep.ber_sweep(code, G, config.R3, [1.0], 5_000, dev, np.float32, seed=4, fused=False)
    assert c == d
    from ldpc_amd import channel
    cfo = channel.doppler_cfo(4000.0, compensated=0.999)   # the Doppler-shifted channel through the sweep
    e = sweep.ber_sweep(code, G, config.R3, [1.0], 5_000, dev, np.float32, seed=4, cfo=cfo, phase0=0.1)
    f = sweep.ber_sweep(code, G, config.R3, [1.0], 5_000, dev, np.float32, seed=4, cfo=cfo, phase0=0.1, fused=False)
    assert e == f and e != c
    # a loop bound of the reference (channel.py:646) has no fused kernel: "auto" falls back, True raises
    bounded = config.R3.with_(num_refresh_vars=300)
    c = sweep.ber_sweep(code, G, bounded, [1.0], 5_000, dev, np.float32, seed=4)
    d = sweep.ber_sweep(code, G, bounded, [1.0], 5_000, dev, np.float32, seed=4, fused=False)
    assert c == d
    with pytest.raises(_lib.LdpcError) as ei:
        sweep.ber_sweep(code, G, bounded, [1.0], 5_000, dev, np.float32, seed=4, fused=True)
    assert ei.value.code == _lib.LDPC_ERR_UNSUPPORTED
    # a matrix that is not compiled in
    other = codes.compile_code(np.roll(code.dense(), 1, axis=1), "rolled")
    with pytest.raises(_lib.LdpcError):
        sweep.ber_sweep(other, None, config.R0, [1.0], 1_000, dev, np.float32, seed=4, fused=True)
