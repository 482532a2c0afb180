"""The device channel kernel (BPSK + AWGN + LLR, ``ldpc_channel_llr``) against its numpy restatement.
The reference's noise comes from numpy's global generator and cannot be reproduced: these values
are pinned by this build's oracle only, and the oracle's integer stream by the published
Philox4x32-10 known-answer vectors."""
import numpy as np
import pytest

from ldpc_amd import channel, config
from oracle import channel_oracle as co


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 with 10 rounds
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        got = co.philox4x32_10(*[[c] for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == out


def test_oracle_noise_is_standard_normal_and_sharding_invariant():
    z = co.standard_normals(4000, 512, seed=99)
    assert abs(z.mean()) < 3e-3 and abs(z.std() - 1) < 3e-3
    assert abs((z ** 4).mean() - 3.0) < 0.05                       # kurtosis of a Gaussian
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.06
    part = co.standard_normals(100, 512, seed=99, first_frame=1234)
    assert np.array_equal(part, co.standard_normals(2000, 512, seed=99)[1234:1334])
    assert not np.array_equal(z[:100], co.standard_normals(100, 512, seed=100))


@pytest.mark.gpu
@pytest.mark.parametrize("n,dtype", [(512, np.float32), (648, np.float64), (101, np.float32), (2080, np.float32)])
def test_device_channel_matches_oracle(n, dtype):
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(n)
    frames = 257
    ref = rng.integers(0, 2, size=(frames, n), dtype=np.uint8)
    ref_dev = torch.as_tensor(ref, device=dev)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    for cfg, prefix in ((config.R0, False), (config.R1, True), (config.R2, False), (config.R3, True)):
        llr = torch.empty((frames, n), dtype=tdt, device=dev)
        channel.channel_llr_device(ref_dev.data_ptr(), n, frames, n, 1.5, cfg, seed=0xC0FFEE1234, first_frame=77,
                                   llr_ptr=llr.data_ptr(), llr_dtype=dtype, compat_prefix=prefix)
        torch.cuda.synchronize()
        s2 = float(channel.noise_variance(1.5))
        gain = (1.0 if cfg.rule == 2 else 2.0) / s2
        plen, pval = (80, 1e-9) if (prefix and cfg.rule == 1) else ((80, 0.0) if prefix else (0, 0.0))
        want = co.channel_llr(ref, frames, n, s2, gain, cfg.plus_is_bit == 1, min(plen, n), pval, 0xC0FFEE1234, 77)
        got = llr.cpu().numpy().astype(np.float64)
        # float32 log / sincos differ by a few ulp between the device library and numpy
        assert np.allclose(got, want, rtol=2e-5, atol=2e-5 * gain)
        if plen:
            assert np.all(got[:, :min(plen, n)] == np.float32(pval))
    # all-zero codeword (NULL reference) and one shared codeword (stride 0)
    llr0 = torch.empty((frames, n), dtype=tdt, device=dev)
    channel.channel_llr_device(0, 0, frames, n, 1.5, config.R1, 5, 0, llr0.data_ptr(), dtype)
    llr1 = torch.empty((frames, n), dtype=tdt, device=dev)
    zero = torch.zeros((n,), dtype=torch.uint8, device=dev)
    channel.channel_llr_device(zero.data_ptr(), 0, frames, n, 1.5, config.R1, 5, 0, llr1.data_ptr(), dtype)
    torch.cuda.synchronize()
    assert torch.equal(llr0, llr1)


@pytest.mark.gpu
def test_device_channel_is_independent_of_batching_and_feeds_the_decoder():
    import torch
    from ldpc_amd import codes, sweep
    from ldpc_amd.decoder import Decoder
    dev = torch.device("cuda", 0)
    code = codes.load_packaged("ccsds512")
    G = codes.load_packaged_generator("ccsds512")
    cfg = config.R0
    whole = sweep.make_batch(code, None, 4096, 2.0, cfg, seed=[4], device=dev, native_noise=True)
    part = sweep.make_batch(code, None, 1000, 2.0, cfg, seed=[4], device=dev, native_noise=True, first_frame=3000)
    torch.cuda.synchronize()
    assert torch.equal(part.llr, whole.llr[3000:4000])
    # statistics of a big batch, then a decode: FER at 2 dB is tiny for R0 on this code
    big = sweep.make_batch(code, G, 1 << 17, 2.0, cfg, seed=[5], device=dev, native_noise=True)
    s2 = float(channel.noise_variance(2.0))
    y = big.llr.double() * (s2 / 2.0) - (2.0 * big.ref.double() - 1.0)          # the noise samples
    assert abs(float(y.mean())) < 2e-3 and abs(float(y.var()) / s2 - 1.0) < 5e-3
    dec = Decoder(code)
    sweep.decode_batch(dec, big, cfg, compute_dtype=np.float64)
    sweep.count_errors(dec, big)
    torch.cuda.synchronize()
    st = sweep.summarize(big.counters.cpu().numpy(), code.n)
    assert st["fer"] < 2e-3 and st["converged_wrong"] == 0 and 2.0 < st["mean_iters"] < 3.2
