"""Sum-product decoding of batches that mix CCSDS and 5G frames under the Doppler-shifted BPSK
channel.  The reference holds neither the channel (only the scale factor of plotDoppler.py:14-16)
nor a batched decode, so the channel is pinned by this build's numpy oracle alone (PARITY UNPINNED),
while the decode rules are the reference's R2 / R3 (bpsk.py:263-387, channel.py:402-513), checked
per code against the CPU restatement and against single-code decodes of the same frames."""
import numpy as np
import pytest

from ldpc_amd import channel, codes, config
from oracle import bp_oracle
from oracle import channel_oracle as co


def test_doppler_helpers_and_oracle():
    assert channel.doppler_scale() == 1.0 - 10.0 / 1500.0                      # plotDoppler.py:5-6,15
    assert np.isclose(channel.doppler_cfo(4000.0), -(10.0 / 1500.0) * 12000.0 / 4000.0)
    assert np.isclose(channel.doppler_cfo(4000.0, compensated=0.999), -2e-5)
    # no offset: the plain channel; no noise: the closed form
    a = co.channel_llr(None, 3, 200, 0.5, 4.0, False, 0, 0.0, seed=9)
    b = co.channel_llr(None, 3, 200, 0.5, 4.0, False, 0, 0.0, seed=9, cfo=0.0, phase0=0.0)
    assert np.array_equal(a, b)
    cfo, ph = 0.003, 0.1
    y = co.channel_llr(None, 1, 300, 0.0, 1.0, False, 0, 0.0, seed=9, cfo=cfo, phase0=ph)[0]
    i = np.arange(300)
    want = np.sinc(cfo) * np.cos(2 * np.pi * (cfo * (i + 0.5) + ph))
    assert np.allclose(y, want, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("n,dtype", [(512, np.float32), (2080, np.float64), (26112, np.float32)])
def test_device_doppler_channel_matches_oracle(n, dtype):
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(n)
    frames = 33
    ref = rng.integers(0, 2, size=(frames, n), dtype=np.uint8)
    ref_dev = torch.as_tensor(ref, device=dev)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    for cfg, cfo, ph in ((config.R3, -2e-5, 0.0), (config.R2, 1.3e-3, 0.21), (config.R0, 0.0, -0.4)):
        llr = torch.empty((frames, n), dtype=tdt, device=dev)
        channel.channel_llr_device(ref_dev.data_ptr(), n, frames, n, 2.0, cfg, seed=77, first_frame=5,
                                   llr_ptr=llr.data_ptr(), llr_dtype=dtype, cfo=cfo, phase0=ph)
        torch.cuda.synchronize()
        s2 = float(channel.noise_variance(2.0))
        gain = (1.0 if cfg.rule == 2 else 2.0) / s2
        want = co.channel_llr(ref, frames, n, s2, gain, cfg.plus_is_bit == 1, 0, 0.0, 77, 5, cfo=cfo, phase0=ph)
        got = llr.cpu().numpy().astype(np.float64)
        # the phase is formed in float32: a few 1e-6 cycles at position 26 000, times 2 pi, times the gain
        assert np.allclose(got, want, rtol=2e-5, atol=4e-4 * gain)
    with pytest.raises(Exception):
        channel.channel_llr_device(0, 0, 1, n, 2.0, config.R3, 1, 0, llr.data_ptr(), dtype, cfo=0.7)


@pytest.mark.gpu
@pytest.mark.parametrize("rule", ["R2", "R3"])
def test_mixed_ccsds_and_5g_batch(rule, c_oracle):
    import torch
    from ldpc_amd import sweep
    from ldpc_amd.decoder import Decoder
    dev = torch.device("cuda", 0)
    cfg = config.PRESETS[rule].with_(max_iters=25)
    cfo = channel.doppler_cfo(4000.0, compensated=0.999)
    spec = [("ccsds512", 3000, 0.0), ("nr_bg2_z40_r1400", 700, -4.5)]
    parts, singles = [], []
    counters = torch.zeros((5,), dtype=torch.int64, device=dev)
    for name, frames, snr in spec:
        code = codes.load_packaged(name)
        G = codes.load_packaged_generator(name) if name == "ccsds512" else None
        dec = Decoder(code)
        cw = sweep.make_codewords(code, G, frames, [3, len(name)], dev)
        bt = sweep.noisy_batch(code, cw, snr, cfg, [4, len(name)], dev, llr_dtype=torch.float32, cfo=cfo)
        parts.append((dec, bt))
    sweep.decode_mixed(parts, cfg, compute_dtype=np.float64, counters=counters)
    torch.cuda.synchronize()
    total = np.zeros(5, np.int64)
    for (dec, bt), (name, frames, snr) in zip(parts, spec):
        # the same frames through the single-code entry point: identical bytes
        alone = sweep._empty_outputs(bt.ref, bt.llr, dev)
        sweep.decode_batch(dec, alone, cfg, compute_dtype=np.float64)
        torch.cuda.synchronize()
        assert torch.equal(alone.hard, bt.hard) and torch.equal(alone.iters, bt.iters) and torch.equal(alone.converged, bt.converged)
        hard, iters, conv = bt.hard.cpu().numpy(), bt.iters.cpu().numpy(), bt.converged.cpu().numpy().astype(bool)
        ref = bt.ref.cpu().numpy()
        total += [frames, int((~conv).sum()), int((conv & (hard != ref).any(axis=1)).sum()), int((hard != ref).sum()),
                  int(iters.sum())]
        # per code against the CPU restatement (libm transcendentals there, polynomial ones on the device:
        # a frame at a decision boundary may take one iteration more or less)
        k = min(frames, 400)
        eh, ei, ec, _ = c_oracle.c_decode(dec.code, cfg, bt.llr[:k].cpu().numpy().astype(np.float64), want_posterior=False)
        same = (ei == iters[:k]) & (ec == conv[:k]) & (eh == hard[:k]).all(axis=1)
        assert same.mean() >= 0.99, (name, same.mean())
        assert conv.mean() > 0.5, (name, conv.mean())
        # ... and strictly against the numpy restatement, the one pinned to the reference's own methods
        # (numpy's tanh / arctanh, as bpsk.py:325-339 and channel.py:471-474 call them): flags, iteration
        # counts and words identical on every frame
        k = 256
        nh, ni, nc, _ = bp_oracle.numpy_decode(dec.code, cfg, bt.llr[:k].cpu().numpy().astype(np.float64))
        assert np.array_equal(ni, iters[:k]) and np.array_equal(nc, conv[:k]) and np.array_equal(nh, hard[:k]), name
    assert counters.cpu().numpy().tolist() == total.tolist()
    assert total[1] > 0 and total[3] > 0


@pytest.mark.gpu
def test_uncompensated_doppler_breaks_the_link():
    """With a tenth of a percent of the shift left the carrier turns by a half cycle over a
    CCSDS frame and the decoder fails; with a thousandth of a percent it decodes as on AWGN."""
    import torch
    from ldpc_amd import sweep
    from ldpc_amd.decoder import Decoder
    dev = torch.device("cuda", 0)
    code = codes.load_packaged("ccsds512")
    G = codes.load_packaged_generator("ccsds512")
    dec = Decoder(code)
    cfg = config.R3
    fer = []
    for comp in (1.0, 0.99999, 0.95):
        cw = sweep.make_codewords(code, G, 20000, [8], dev)
        bt = sweep.noisy_batch(code, cw, 3.0, cfg, [9], dev, cfo=channel.doppler_cfo(4000.0, compensated=comp))
        sweep.decode_batch(dec, bt, cfg, compute_dtype=np.float32)
        sweep.count_errors(dec, bt)
        torch.cuda.synchronize()
        fer.append(sweep.summarize(bt.counters.cpu().numpy(), code.n)["fer"])
    assert fer[0] < 0.02 and abs(fer[1] - fer[0]) < 0.01 and fer[2] > 0.5, fer
