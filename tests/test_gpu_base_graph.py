"""The layered kernel compiled per 5G NR base graph (connectivity only) with lifting size, shift table
and number of active block rows at run time: what a link decodes after rate matching, the step the
reference fakes with H[:1400, :] (/root/reference/channel.py:35) and a hard-coded prefix
(/root/reference/bpsk.py:163).  Layered min-sum is a build extension: parity is against this
build's CPU restatement (oracle/bp_oracle.c: decode_lnms_*), bit for bit."""
import numpy as np
import pytest

from ldpc_amd import _lib, codes, config, nr_bg1
from ldpc_amd import nr_rate_match as rm

pytestmark = pytest.mark.gpu


def llrs(n, frames, snr_db, seed):
    rng = np.random.default_rng(seed)
    s2 = 1 / (2 * 10 ** (snr_db / 10))
    y = 1.0 + np.sqrt(s2) * rng.standard_normal((frames, n))     # all-zero codeword, bit 0 -> +1
    return ((2 * y) / s2).astype(np.float32).astype(np.float64)


def same(res, want, posterior=True):
    eh, ei, ec, ep = want
    assert np.array_equal(res.iters, ei) and np.array_equal(res.converged, ec)
    assert np.array_equal(res.hard, eh)
    if posterior:
        assert np.array_equal(np.asarray(res.posterior, np.float64), ep)


@pytest.mark.parametrize("Z,seed", [(384, 11), (128, 12), (96, 13)])
@pytest.mark.parametrize("rows", [46, 26, 10])
def test_bg1_connectivity_any_shifts_any_prefix(Z, seed, rows, c_oracle):
    """BG1 connectivity with shift tables that are not compiled in, cut to 46 / 26 / 10 block rows as
    rate matching does: the base-graph kernel, identical to the CPU restatement in float and double."""
    from ldpc_amd.decoder import Decoder
    qc = rm.active_submatrix(nr_bg1.build(Z=Z, seed=seed), rows)
    code = codes.compile_qc(qc, f"bg1_z{Z}_rows{rows}")
    dec = Decoder(code)
    snr = {46: -0.8, 26: 1.0, 10: 4.0}[rows]
    frames = 24 if Z == 384 else 64
    L = llrs(code.n, frames, snr, [Z, rows])
    for cfg in (config.LNMS, config.LNMS.with_(early_exit=False, max_iters=4)):
        want = c_oracle.c_decode_lnms(code, cfg, L, use_float=True)
        res = dec.decode(L.astype(np.float32), cfg, compute_dtype=np.float32, want_posterior=True)
        assert res.kernel == "bg_lnms_f32"
        same(res, want)
        res = dec.decode(L.astype(np.float32), cfg, compute_dtype=np.float32)   # decisions of degree-one bits from the state word
        same(res, want, posterior=False)
        if cfg.early_exit:
            assert 0 < want[2].sum()
    want = c_oracle.c_decode_lnms(code, config.LNMS, L)
    res = dec.decode(L, config.LNMS, compute_dtype=np.float64, want_posterior=True)
    assert res.kernel == "bg_lnms_f64"
    same(res, want)
    res = dec.decode(L.astype(np.float32), config.LNMS, compute_dtype=np.float64, want_posterior=True)   # float LLRs, double arithmetic
    same(res, want)


@pytest.mark.parametrize("name", ["nr_bg1_z384", "nr_bg2_z40", "nr_bg2_z40_r1400"])
def test_base_graph_kernel_equals_the_compiled_in_kernel(name, c_oracle):
    """Forced on the packaged tables it gives what their own compiled-in kernel gives (the 1400-row cut
    keeps its 280 columns that sit in no check: the channel decides them)."""
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged(name)
    dec = Decoder(code)
    frames = 32 if name == "nr_bg1_z384" else 200
    L = llrs(code.n, frames, {"nr_bg1_z384": -0.5, "nr_bg2_z40": -5.5, "nr_bg2_z40_r1400": -4.0}[name], 5)
    for cfg in (config.LNMS, config.LNMS.with_(early_exit=False, max_iters=3)):
        for dt, kname in ((np.float32, "f32"), (np.float64, "f64")):
            a = dec.decode(L.astype(dt), cfg, compute_dtype=dt, want_posterior=True)
            b = dec.decode(L.astype(dt), cfg, compute_dtype=dt, want_posterior=True, kernel=_lib.KERNEL_GRAPH)
            assert a.kernel == "spec_lnms_" + kname and b.kernel == "bg_lnms_" + kname
            assert np.array_equal(a.hard, b.hard) and np.array_equal(a.iters, b.iters)
            assert np.array_equal(a.converged, b.converged) and np.array_equal(a.posterior, b.posterior)
        same(b, c_oracle.c_decode_lnms(code, cfg, L))


@pytest.mark.parametrize("bg,Z", [(2, 2), (2, 3), (2, 7), (2, 36), (2, 52), (2, 64), (2, 104), (2, 208), (2, 384),
                                  (1, 5), (1, 44), (1, 72), (1, 240), (1, 352)])
def test_lifting_sizes_of_the_standard(bg, Z, c_oracle):
    """Lifting sizes from every row of 38.212 Table 5.3.2-1 (small ones leave most lanes idle, 208 and
    352 do not divide the workgroup evenly, 2 and 3 make single-wave workgroups), tables lifted with
    P = V mod Z from the packaged ones."""
    from ldpc_amd.decoder import Decoder
    base = codes.load_packaged("nr_bg1_z384" if bg == 1 else "nr_bg2_z40").qc
    if bg == 2:   # shifts spread over the lifting size instead of the packaged table's 0..4
        rng = np.random.default_rng(Z)
        base = codes.QCStructure(384, base.mb, base.nb, base.edge_row, base.edge_col,
                                 np.where(base.edge_col >= 14, 0, rng.integers(0, 384, len(base.edge_row))).astype(np.int32))
    code = codes.compile_qc(rm.lift(base, Z), f"bg{bg}_z{Z}")
    dec = Decoder(code)
    frames = max(8, min(200, 20000 // code.n))
    frames += frames % 2 == 0      # an odd count: the last workgroup's slots are not all used
    L = llrs(code.n, frames, 1.0 if Z < 16 else -1.0, [bg, Z])
    for cfg in (config.LNMS.with_(max_iters=12), config.LNMS.with_(early_exit=False, max_iters=3)):
        res = dec.decode(L.astype(np.float32), cfg, compute_dtype=np.float32, want_posterior=True)
        assert res.kernel == "bg_lnms_f32"
        same(res, c_oracle.c_decode_lnms(code, cfg, L, use_float=True))
    res = dec.decode(L, config.LNMS, compute_dtype=np.float64, want_posterior=True)
    assert res.kernel == "bg_lnms_f64"
    same(res, c_oracle.c_decode_lnms(code, config.LNMS, L))


@pytest.mark.parametrize("name,rows", [("nr_bg1_z384", 20), ("nr_bg1_z384", 4), ("nr_bg2_z40", 35), ("nr_bg2_z40", 6)])
def test_num_active_checks_is_the_row_prefix(name, rows, c_oracle):
    """num_active_checks on the full matrix = the decode of the sub-matrix of those rows: sweeps,
    syndrome and "degree one" all see the prefix only, the bits beyond it decide from the channel."""
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged(name)
    Z = code.qc.Z
    dec = Decoder(code)
    L = llrs(code.n, 24 if Z == 384 else 100, 3.0 if rows < 10 else 0.0, [rows, Z])
    cfg = config.LNMS.with_(num_active_checks=rows * Z)
    for dt, use_float, kname in ((np.float32, True, "bg_lnms_f32"), (np.float64, False, "bg_lnms_f64")):
        want = c_oracle.c_decode_lnms(code, cfg, L, use_float=use_float)
        res = dec.decode(L.astype(dt), cfg, compute_dtype=dt, want_posterior=True)
        assert res.kernel == kname
        same(res, want)
        assert 0 < want[2].sum()
    # the same rows as a code of their own
    sub = codes.compile_qc(rm.active_submatrix(code.qc, rows), "sub")
    a = Decoder(sub).decode(L[:, :sub.n].astype(np.float32), config.LNMS, compute_dtype=np.float32)
    assert a.kernel == "bg_lnms_f32"
    assert np.array_equal(a.hard, res.hard[:, :sub.n]) and np.array_equal(a.iters, res.iters)
    # the bits beyond the prefix: the channel's decision
    assert np.array_equal(res.hard[:, sub.n:], (L[:, sub.n:] <= 0).astype(np.uint8))


def test_what_the_base_graph_kernel_refuses():
    from ldpc_amd.decoder import Decoder
    bg1 = Decoder(codes.load_packaged("nr_bg1_z384"))
    L = np.ones((2, 26112), np.float32)
    for cfg in (config.LNMS.with_(num_active_checks=3 * 384),      # fewer rows than the core: a systematic column would have degree one
                config.LNMS.with_(num_active_checks=10 * 384 + 1)):  # not whole block rows
        with pytest.raises(_lib.LdpcError) as err:
            bg1.decode(L, cfg, compute_dtype=np.float32)
        assert err.value.code == _lib.LDPC_ERR_UNSUPPORTED
    with pytest.raises(_lib.LdpcError) as err:                      # not layered min-sum
        bg1.decode(L, config.R1, compute_dtype=np.float32, kernel=_lib.KERNEL_GRAPH)
    assert err.value.code == _lib.LDPC_ERR_ARG
    wifi = Decoder(codes.load_packaged("wifi648"))                  # not an NR base graph
    with pytest.raises(_lib.LdpcError) as err:
        wifi.decode(np.ones((2, 648), np.float32), config.LNMS, compute_dtype=np.float32, kernel=_lib.KERNEL_GRAPH)
    assert err.value.code == _lib.LDPC_ERR_ARG
    with pytest.raises(_lib.LdpcError) as err:                      # the table-driven kernel has no row prefix
        wifi.decode(np.ones((2, 648), np.float32), config.LNMS.with_(num_active_checks=27), compute_dtype=np.float32)
    assert err.value.code == _lib.LDPC_ERR_UNSUPPORTED


def test_ragged_and_tiny_batches(c_oracle):
    """Fewer frames than one workgroup holds, and one frame more than a whole number of workgroups."""
    from ldpc_amd.decoder import Decoder
    code = codes.compile_qc(rm.lift(codes.load_packaged("nr_bg1_z384").qc, 96), "bg1_z96")
    dec = Decoder(code)
    for frames in (1, 7, 8, 9, 8 * 600 + 1):
        L = llrs(code.n, min(frames, 40), -1.0, frames)
        if frames > 40:
            L = np.tile(L, (frames // 40 + 1, 1))[:frames]
        res = dec.decode(L.astype(np.float32), config.LNMS, compute_dtype=np.float32)
        assert res.kernel == "bg_lnms_f32"
        want = c_oracle.c_decode_lnms(code, config.LNMS, L[:40], use_float=True)
        k = min(frames, 40)
        assert np.array_equal(res.hard[:k], want[0][:k]) and np.array_equal(res.iters[:k], want[1][:k])
        if frames > 40:   # the tiled copies decode like their originals, wherever they land in a workgroup
            idx = np.arange(frames) % 40
            assert np.array_equal(res.iters, want[1][idx]) and np.array_equal(res.hard, want[0][idx])
