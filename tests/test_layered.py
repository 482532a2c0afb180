"""Row-layered normalised min-sum (build extension; no counterpart in the reference, so parity is
against this build's own CPU restatements only) and the build-authored NR BG1 Z = 384 code."""
import numpy as np
import pytest

from ldpc_amd import codes, config
from oracle import bp_oracle

SNR = {"wifi648": 0.5, "nr_bg2_z40": -5.5, "nr_bg2_z40_r1400": -4.0, "nr_bg1_z384": -0.5}


def llrs(code, name, frames, seed):
    rng = np.random.default_rng(seed)
    s2 = 1 / (2 * 10 ** (SNR[name] / 10))
    y = 1.0 + np.sqrt(s2) * rng.standard_normal((frames, code.n))     # all-zero codeword, bit 0 -> +1
    return ((2 * y) / s2).astype(np.float32).astype(np.float64)


def test_bg1_structure():
    from ldpc_amd import nr_bg1
    nr_bg1.self_check()
    c = codes.load_packaged("nr_bg1_z384")
    assert (c.m, c.n, c.num_edges, c.max_check_degree, c.max_var_degree) == (17664, 26112, 121344, 19, 30)
    q = c.qc
    assert (q.Z, q.mb, q.nb, q.num_base_edges) == (384, 46, 68, 316)
    shifts = -np.ones((46, 68), int)
    shifts[q.edge_row, q.edge_col] = q.edge_shift
    assert not nr_bg1._has_4cycle(nr_bg1.BG1_ROWS, shifts, 384)
    assert c.is_codeword(np.zeros((1, c.n), np.uint8)).all()
    deg = np.diff(c.col_ptr)
    assert (deg[26 * 384:] == 1).all()                       # extension parity columns


def test_bg1_encoder_makes_codewords():
    from ldpc_amd import nr_bg1
    rng = np.random.default_rng(2)
    for Z, seed in ((384, 38212), (96, 6), (64, 3)):
        qc = nr_bg1.build(Z=Z, seed=seed)
        code = codes.compile_qc(qc, "t")
        info = rng.integers(0, 2, size=(7, 22 * Z), dtype=np.uint8)
        cw = nr_bg1.encode(qc, info)
        assert code.is_codeword(cw).all() and np.array_equal(cw[:, :22 * Z], info)
    enc = codes.load_packaged_generator("nr_bg1_z384")
    info = rng.integers(0, 2, size=(3, enc.k), dtype=np.uint8)
    assert enc.k == 8448 and codes.load_packaged("nr_bg1_z384").is_codeword(enc(info)).all()


@pytest.mark.parametrize("name", ["wifi648", "nr_bg2_z40"])
def test_numpy_and_c_layered_oracles_agree(name, c_oracle):
    code = codes.load_packaged(name)
    L = llrs(code, name, 6, 1)
    for use_float, dt in ((False, np.float64), (True, np.float32)):
        a = c_oracle.c_decode_lnms(code, config.LNMS, L, use_float=use_float)
        b = bp_oracle.numpy_decode_lnms(code, config.LNMS, L, dtype=dt)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    fixed = config.LNMS.with_(early_exit=False, max_iters=4)
    a = c_oracle.c_decode_lnms(code, fixed, L)
    b = bp_oracle.numpy_decode_lnms(code, fixed, L)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and (a[1] == 4).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["wifi648", "nr_bg2_z40", "nr_bg2_z40_r1400", "nr_bg1_z384"])
def test_layered_kernel_matches_oracle(name, c_oracle):
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged(name)
    dec = Decoder(code)
    frames = 96 if name == "nr_bg1_z384" else 384
    L = llrs(code, name, frames, 3)
    for cfg in (config.LNMS, config.LNMS.with_(early_exit=False, max_iters=5)):
        eh, ei, ec, ep = c_oracle.c_decode_lnms(code, cfg, L, use_float=True)
        res = dec.decode(L.astype(np.float32), cfg, compute_dtype=np.float32, want_posterior=True)
        assert res.kernel == "spec_lnms_f32"
        assert np.array_equal(res.iters, ei) and np.array_equal(res.converged, ec)
        assert np.array_equal(res.hard, eh)
        assert np.array_equal(res.posterior.astype(np.float64), ep)
        if cfg.early_exit:
            assert 0 < ec.sum()
        # without posteriors the decisions of degree-one variables come from the check state
        res = dec.decode(L.astype(np.float32), cfg, compute_dtype=np.float32)
        assert np.array_equal(res.hard, eh) and np.array_equal(res.iters, ei) and np.array_equal(res.converged, ec)
        eh, ei, ec, ep = c_oracle.c_decode_lnms(code, cfg, L)
        res = dec.decode(L, cfg, compute_dtype=np.float64, want_posterior=True)
        assert res.kernel == "spec_lnms_f64"
        assert np.array_equal(res.iters, ei) and np.array_equal(res.converged, ec)
        assert np.array_equal(res.hard, eh) and np.array_equal(res.posterior, ep)


@pytest.mark.gpu
def test_layered_refused_for_double_circulant_code():
    from ldpc_amd import _lib
    from ldpc_amd.decoder import Decoder
    dec = Decoder(codes.load_packaged("ccsds512"))
    with pytest.raises(_lib.LdpcError) as ei:
        dec.decode(np.ones((1, 512)), config.LNMS)
    assert ei.value.code == _lib.LDPC_ERR_UNSUPPORTED


@pytest.mark.gpu
def test_flooding_rules_on_bg1_through_the_global_workspace(c_oracle):
    """BG1's flooding messages exceed LDS: the edge-list kernel keeps them in its global workspace."""
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged("nr_bg1_z384")
    dec = Decoder(code)
    L = llrs(code, "nr_bg1_z384", 8, 5)
    for cfg in (config.R0.with_(max_iters=4), config.R1.with_(max_iters=6)):
        Lr = -L if cfg.rule == 0 else L
        eh, ei, ec, ep = c_oracle.c_decode(code, cfg, Lr)
        res = dec.decode(Lr, cfg, want_posterior=True)
        assert res.kernel.startswith("csr_global")
        assert np.array_equal(res.iters, ei) and np.array_equal(res.hard, eh) and np.array_equal(res.posterior, ep)
    # the workspace is sized for the batch (8 frames of about 2.4 MB of messages), not for every workgroup
    # the device could hold (some 5 GB for this code)
    assert 8 * 2 * code.num_edges * 8 <= dec.workspace_bytes() <= 64 << 20
    L2 = np.tile(L, (6, 1))
    res = dec.decode(L2, config.R1.with_(max_iters=2), want_posterior=False)      # a larger batch grows it ...
    grown = dec.workspace_bytes()
    assert 48 * 2 * code.num_edges * 8 <= grown <= 256 << 20
    dec.decode(L, config.R1.with_(max_iters=2))                                   # ... and a smaller one keeps it
    assert dec.workspace_bytes() == grown


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["wifi648", "nr_bg2_z40", "nr_bg2_z40_r1400", "nr_bg1_z384"])
def test_table_driven_layered_kernel_matches_oracle(name, c_oracle):
    """The layered kernel that reads the base graph from device tables (kernel selector QC), on
    the packaged codes: same results as the CPU restatement, bit for bit."""
    from ldpc_amd import _lib
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged(name)
    dec = Decoder(code)
    frames = 64 if name == "nr_bg1_z384" else 300
    L = llrs(code, name, frames, 7)
    for cfg in (config.LNMS, config.LNMS.with_(early_exit=False, max_iters=4)):
        eh, ei, ec, ep = c_oracle.c_decode_lnms(code, cfg, L, use_float=True)
        res = dec.decode(L.astype(np.float32), cfg, compute_dtype=np.float32, want_posterior=True, kernel=2)
        assert res.kernel in ("qc_lnms_f32", "qc_lnms_global_f32")
        assert np.array_equal(res.iters, ei) and np.array_equal(res.converged, ec)
        assert np.array_equal(res.hard, eh) and np.array_equal(res.posterior.astype(np.float64), ep)
        if name == "nr_bg1_z384":   # 209 KiB of posteriors in double: more than LDS
            with pytest.raises(_lib.LdpcError) as err:
                dec.decode(L, cfg, compute_dtype=np.float64, kernel=2)
            assert err.value.code == _lib.LDPC_ERR_UNSUPPORTED
            continue
        eh, ei, ec, ep = c_oracle.c_decode_lnms(code, cfg, L)
        res = dec.decode(L, cfg, compute_dtype=np.float64, want_posterior=True, kernel=2)
        assert res.kernel in ("qc_lnms_f64", "qc_lnms_global_f64")
        assert np.array_equal(res.iters, ei) and np.array_equal(res.converged, ec)
        assert np.array_equal(res.hard, eh) and np.array_equal(res.posterior, ep)


@pytest.mark.gpu
def test_layered_on_a_base_graph_that_is_not_compiled_in(c_oracle):
    """A caller's own table: BG1 connectivity with other shift coefficients and a smaller lifting
    size takes the base-graph kernel automatically (connectivity compiled in, shifts and lifting size at
    run time); the table-driven kernel, forced, gives the same."""
    from ldpc_amd import nr_bg1
    from ldpc_amd.decoder import Decoder
    for Z, seed in ((384, 5), (96, 6)):
        code = codes.compile_qc(nr_bg1.build(Z=Z, seed=seed), f"bg1_z{Z}_seed{seed}")
        dec = Decoder(code)
        rng = np.random.default_rng([Z, seed])
        s2 = 1 / (2 * 10 ** (-0.5 / 10))
        L = ((2 * (1.0 + np.sqrt(s2) * rng.standard_normal((48, code.n)))) / s2).astype(np.float32).astype(np.float64)
        eh, ei, ec, ep = c_oracle.c_decode_lnms(code, config.LNMS, L, use_float=True)
        res = dec.decode(L.astype(np.float32), config.LNMS, compute_dtype=np.float32, want_posterior=True)
        assert res.kernel == "bg_lnms_f32"
        assert np.array_equal(res.iters, ei) and np.array_equal(res.converged, ec)
        assert np.array_equal(res.hard, eh) and np.array_equal(res.posterior.astype(np.float64), ep)
        assert ec.sum() > 0
        res = dec.decode(L.astype(np.float32), config.LNMS, compute_dtype=np.float32, want_posterior=True, kernel=2)
        assert res.kernel.startswith("qc_lnms")
        assert np.array_equal(res.iters, ei) and np.array_equal(res.hard, eh) and np.array_equal(res.posterior.astype(np.float64), ep)
