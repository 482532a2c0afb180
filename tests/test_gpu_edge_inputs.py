"""Inputs at the edges of the arithmetic: channel values of exactly -0.0 / +0.0 through every kernel
family (numpy.sign(-0.0) is 0 and `-0.0 < 0` is false in the reference, ldpc.py:186,198-201, bpsk.py:223-229,240-243;
a kernel that reads sign BITS must not see -0.0 as negative), and liftings so small that a workgroup
holds more than eight frame slots."""
import numpy as np
import pytest

from ldpc_amd import _lib, codes, config
from ldpc_amd import nr_rate_match as rm
from oracle import bp_oracle

pytestmark = pytest.mark.gpu


def zero_salted(n, frames, snr_db, seed, plus_is_bit):
    rng = np.random.default_rng(seed)
    s2 = 1 / (2 * 10 ** (snr_db / 10))
    y = (-1.0 if plus_is_bit else 1.0) + np.sqrt(s2) * rng.standard_normal((frames, n))   # all-zero codeword
    L = ((2 * y) / s2).astype(np.float32).astype(np.float64)
    L[rng.random(L.shape) < 0.03] = -0.0
    L[rng.random(L.shape) < 0.03] = 0.0
    L[0, :] = -0.0          # a whole frame of negative zeros
    L[1, :] = 0.0
    assert np.signbit(L).any() and (L == 0).sum() > frames
    return L


@pytest.mark.parametrize("name,snr", [("ccsds512", 1.0), ("wifi648", 1.0), ("nr_bg2_z40", -4.0)])
@pytest.mark.parametrize("rule", ["R0", "R1", "OMS", "R3"])
def test_negative_zero_llrs_flooding(name, snr, rule, c_oracle):
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged(name)
    dec = Decoder(code)
    cfg = config.PRESETS[rule].with_(max_iters=8)
    L = zero_salted(code.n, 40, snr, 5, cfg.plus_is_bit)
    want = c_oracle.c_decode(code, cfg, L)
    if rule in ("R0", "R1"):     # the numpy restatement is the one pinned to the reference's own methods
        ref = bp_oracle.numpy_decode(code, cfg, L[:12])
        assert np.array_equal(ref[0], want[0][:12]) and np.array_equal(ref[1], want[1][:12]) and np.array_equal(ref[3], want[3][:12])
    for kernel in (_lib.KERNEL_AUTO, _lib.KERNEL_QC, _lib.KERNEL_CSR):
        for dt in (np.float64, np.float32):
            res = dec.decode(L.astype(dt), cfg, compute_dtype=np.float64, want_posterior=True, kernel=kernel)
            assert np.array_equal(res.iters, want[1]) and np.array_equal(res.converged, want[2]), (res.kernel, dt)
            assert np.array_equal(res.hard, want[0]), (res.kernel, dt)
            if rule == "R3":
                assert np.allclose(res.posterior, want[3], rtol=1e-4, atol=1e-9), res.kernel
            else:
                assert np.array_equal(res.posterior, want[3]), (res.kernel, dt)
    # float arithmetic: same decisions on the frames that are all zeros (nothing to round there)
    res = dec.decode(L.astype(np.float32), cfg, compute_dtype=np.float32)
    assert np.array_equal(res.hard[:2], want[0][:2]) and np.array_equal(res.iters[:2], want[1][:2])


@pytest.mark.parametrize("name,snr", [("wifi648", 1.0), ("nr_bg2_z40", -4.0), ("nr_bg1_z384", -1.0)])
def test_negative_zero_llrs_layered(name, snr, c_oracle):
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged(name)
    dec = Decoder(code)
    cfg = config.LNMS.with_(max_iters=6)
    L = zero_salted(code.n, 12 if name == "nr_bg1_z384" else 40, snr, 6, cfg.plus_is_bit)
    kernels = [_lib.KERNEL_AUTO, _lib.KERNEL_QC] + ([_lib.KERNEL_GRAPH] if name.startswith("nr_") else [])
    for use_float, dt in ((False, np.float64), (True, np.float32)):
        want = c_oracle.c_decode_lnms(code, cfg, L, use_float=use_float)
        for kernel in kernels:
            if name == "nr_bg1_z384" and kernel == _lib.KERNEL_QC and dt == np.float64:
                continue        # its posteriors in double exceed LDS
            res = dec.decode(L.astype(dt), cfg, compute_dtype=dt, want_posterior=True, kernel=kernel)
            assert np.array_equal(res.iters, want[1]) and np.array_equal(res.converged, want[2]), res.kernel
            assert np.array_equal(res.hard, want[0]) and np.array_equal(np.asarray(res.posterior, np.float64), want[3]), res.kernel


@pytest.mark.parametrize("Z", [4, 5, 7])
def test_tiny_liftings_put_many_frames_in_a_workgroup(Z, c_oracle):
    """BG2 connectivity at Z = 4, 5, 7: the table-driven circulant kernels pack 128 / Z frame slots into a
    workgroup (more than the eight the compiled-in kernels hold), the base-graph kernel sixteen."""
    from ldpc_amd.decoder import Decoder
    base = codes.load_packaged("nr_bg2_z40").qc
    rng = np.random.default_rng(Z)
    shifts = np.where(base.edge_col >= 14, 0, rng.integers(0, 384, len(base.edge_row))).astype(np.int32)
    qc = rm.lift(codes.QCStructure(384, base.mb, base.nb, base.edge_row, base.edge_col, shifts), Z)
    code = codes.compile_qc(qc, f"bg2_z{Z}")
    dec = Decoder(code)
    s2 = 1 / (2 * 10 ** (0.5 / 10))
    L = ((2 * (1.0 + np.sqrt(s2) * rng.standard_normal((203, code.n)))) / s2).astype(np.float32).astype(np.float64)
    for rule in ("R0", "R1", "OMS"):
        cfg = config.PRESETS[rule].with_(max_iters=10)
        Lr = -L if cfg.plus_is_bit else L
        want = c_oracle.c_decode(code, cfg, Lr)
        for kernel in (_lib.KERNEL_AUTO, _lib.KERNEL_QC):
            res = dec.decode(Lr, cfg, compute_dtype=np.float64, want_posterior=True, kernel=kernel)
            assert res.kernel.startswith("qc"), res.kernel
            assert np.array_equal(res.iters, want[1]) and np.array_equal(res.hard, want[0]) and np.array_equal(res.posterior, want[3]), (rule, res.kernel)
    cfg = config.LNMS.with_(max_iters=10)
    for use_float, dt in ((False, np.float64), (True, np.float32)):
        want = c_oracle.c_decode_lnms(code, cfg, L, use_float=use_float)
        for kernel, prefix in ((_lib.KERNEL_AUTO, "bg_lnms"), (_lib.KERNEL_QC, "qc_lnms")):
            res = dec.decode(L.astype(dt), cfg, compute_dtype=dt, want_posterior=True, kernel=kernel)
            assert res.kernel.startswith(prefix), res.kernel
            assert np.array_equal(res.iters, want[1]) and np.array_equal(res.converged, want[2])
            assert np.array_equal(res.hard, want[0]) and np.array_equal(np.asarray(res.posterior, np.float64), want[3]), res.kernel
    assert want[2].sum() > 0


@pytest.mark.parametrize("name,snr", [("ccsds512", 0.0), ("wifi648", 0.0), ("nr_bg2_z40", -5.0)])
@pytest.mark.parametrize("rule", ["R2", "R3"])
def test_saturated_and_infinite_llrs_tanh_rules(name, snr, rule, c_oracle):
    """Channel values of +-1e30, +-infinity and +-1e4 (known or filler positions, saturated front ends) under the
    tanh rules: numpy's tanh saturates at +-1 and the product clip takes over (bpsk.py:333-339, channel.py:471-474);
    every kernel family gives the C restatement's decisions, iteration counts and (finite) posteriors."""
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged(name)
    dec = Decoder(code)
    cfg = config.PRESETS[rule].with_(max_iters=10)
    rng = np.random.default_rng(17)
    s2 = 1 / (2 * 10 ** (snr / 10))
    sign = -1.0 if cfg.plus_is_bit else 1.0
    frames = 48
    L = (2 * (sign + np.sqrt(s2) * rng.standard_normal((frames, code.n))) / s2).astype(np.float32).astype(np.float64)
    for value, share in ((1e30, 0.02), (np.inf, 0.02), (1e4, 0.03)):
        pick = rng.random(L.shape) < share
        L[pick] = sign * value                     # known bits of the all-zero word
    wrong = rng.random(L.shape) < 0.002
    L[wrong] = -sign * 1e30                        # and a few saturated the wrong way
    L[0, :] = sign * np.inf
    L[1, : code.n // 2] = sign * 1e30
    want = c_oracle.c_decode(code, cfg, L)
    assert not np.isnan(want[3]).any() and 0 < want[2].sum()   # (a frame that passes the first test keeps its infinite LLRs as posteriors)
    for kernel in (_lib.KERNEL_AUTO, _lib.KERNEL_QC, _lib.KERNEL_CSR):
        res = dec.decode(L, cfg, compute_dtype=np.float64, want_posterior=True, kernel=kernel)
        assert np.array_equal(res.iters, want[1]) and np.array_equal(res.converged, want[2]), res.kernel
        assert np.array_equal(res.hard, want[0]), res.kernel
        assert not np.isnan(res.posterior).any() and np.allclose(res.posterior, want[3], rtol=1e-4, atol=1e-9), res.kernel
    res = dec.decode(L.astype(np.float32), cfg, compute_dtype=np.float32, want_posterior=True)
    assert not np.isnan(res.posterior).any()
    # float arithmetic saturates the product clip one float ulp below 1 (messages up to 17 instead of 21): frames
    # at the edge may end differently, frames both arithmetic types decode end in the same word
    both = res.converged.astype(bool) & want[2].astype(bool)
    assert both[0]   # the all-infinite frame; the wrong-way saturations fail most of the others in either arithmetic
    assert np.array_equal(res.hard[both], want[0][both])
