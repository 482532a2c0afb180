"""The two CPU oracles against each other on cases the golden vectors do not reach: random
irregular matrices (degree-0 variables, degree-1 checks), the reference's loop-bound quirks, the
fixed-work mode, and the iteration-0 float check of ldpc.py."""
import numpy as np
import pytest

from ldpc_amd import codes, config
from oracle import bp_oracle


def random_code(seed, m=20, n=40, p=0.15):
    rng = np.random.default_rng(seed)
    H = (rng.random((m, n)) < p).astype(np.uint8)
    H[:, 3] = 0                      # a degree-0 variable (channel.py:35 leaves 280 of them)
    H[5] = 0
    H[5, 7] = 1                      # a degree-1 check
    H[0, :2] = 1
    return codes.compile_code(H, f"rand{seed}", find_qc=False)


def llrs(code, seed, frames=24, scale=3.0):
    rng = np.random.default_rng(seed)
    return scale * (rng.standard_normal((frames, code.n)) - 0.6)


@pytest.mark.parametrize("rule", ["R0", "R1", "R2", "R3", "R4"])
@pytest.mark.parametrize("seed", [1, 2])
def test_numpy_and_c_oracles_agree(rule, seed, c_oracle):
    code = random_code(seed)
    cfg = config.PRESETS[rule].with_(max_iters=12)
    L = llrs(code, seed)
    nh, ni, nc, npost = bp_oracle.numpy_decode(code, cfg, L)
    ch, ci, cc, cpost = c_oracle.c_decode(code, cfg, L)
    assert np.array_equal(nc, cc) and np.array_equal(ni, ci) and np.array_equal(nh, ch)
    if cfg.rule <= 1:
        assert np.array_equal(npost, cpost)
    else:
        np.testing.assert_allclose(npost, cpost, rtol=1e-6, atol=1e-9)


def test_loop_bound_quirks(c_oracle):
    code = random_code(4)
    L = llrs(code, 4)
    r0 = config.R0.with_(num_active_checks=8)          # self.m smaller than H.shape[0], ldpc.py:30,175
    a = bp_oracle.numpy_decode(code, r0, L)
    b = c_oracle.c_decode(code, r0, L)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    full = c_oracle.c_decode(code, config.R0, L)
    assert not np.array_equal(full[3], b[3])            # the quirk changes the posteriors
    r1 = config.R1.with_(num_refresh_vars=25, max_iters=8)   # range(self.n) short of H.shape[1], bpsk.py:244
    a = bp_oracle.numpy_decode(code, r1, L)
    b = c_oracle.c_decode(code, r1, L)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_fixed_work_mode(c_oracle):
    code = codes.load_packaged("ccsds512")
    rng = np.random.default_rng(9)
    L = 4.0 * (-1.0 + 0.7 * rng.standard_normal((16, code.n)))
    for rule in ("R0", "R1"):
        cfg = config.PRESETS[rule].with_(max_iters=6, early_exit=False)
        h, it, cv, post = c_oracle.c_decode(code, cfg, L)
        assert (it == 6).all()
        assert np.array_equal(cv, code.is_codeword(h))
        h2, it2, cv2, post2 = bp_oracle.numpy_decode(code, cfg, L)
        assert np.array_equal(h, h2) and np.array_equal(cv, cv2) and np.array_equal(post, post2)


def test_r0_iteration_zero_check_is_made_on_the_llr_floats(c_oracle):
    """ldpc.py:158,165: the first validity test sees the raw LLRs.  All-zero LLRs pass it (every
    row sum is 0), a noiseless codeword does not (numIterations is then 1, never 0)."""
    code = codes.load_packaged("ccsds512")
    L = np.zeros((2, code.n))
    L[1] = -4.1                                           # all-zero codeword, sent as -1 under R0's polarity
    for dec in (bp_oracle.numpy_decode, c_oracle.c_decode):
        h, it, cv, _ = dec(code, config.R0, L)
        assert list(it) == [0, 1] and cv.all()
        assert (h[0] == 0).all() and (h[1] == 0).all()
    h, it, cv, _ = c_oracle.c_decode(code, config.R0.with_(float_syndrome0=False), L)
    # with a proper hard-decision test at iteration 0 both frames stop at once: zero LLRs slice to the
    # all-ones word (a codeword here: every row of H has even weight), -4.1 to the all-zero word
    assert list(it) == [0, 0] and (h[0] == 1).all() and (h[1] == 0).all()


def test_llr_formulas():
    y = np.linspace(-2, 2, 100)
    for rule, factor, prefix in [(0, 2.0, None), (1, 2.0, 1e-9), (2, 1.0, None), (3, 2.0, 0.0), (4, 2.0, 0.0)]:
        L = bp_oracle.llr_from_samples(y, 2.0, rule)
        sigma2 = 1 / (2 * 10 ** 0.2)
        ref = factor * y / sigma2
        lo = 80 if prefix is not None else 0
        np.testing.assert_allclose(L[lo:], ref[lo:], rtol=1e-15)
        if prefix is not None:
            assert (L[:80] == prefix).all()
