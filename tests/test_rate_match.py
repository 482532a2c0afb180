"""5G NR rate matching front end (3GPP TS 38.212 sections 5.2.2, 5.3.2, 5.4.2, 7.2.2).  The reference
holds two constants in its place - the matrix cut to its first 1400 rows (channel.py:35) and the
first 80 LLRs overwritten (bpsk.py:163, channel.py:412) - which the first test derives from the
general mapping; everything else is pinned by the standard's text as restated in
ldpc_amd/nr_rate_match.py (PARITY UNPINNED by the reference) and by round trips."""
import numpy as np
import pytest

from ldpc_amd import codes, config, nr_rate_match as rm


def test_reference_constants_follow_from_the_mapping():
    full = codes.load_packaged("nr_bg2_z40")
    cut = codes.load_packaged("nr_bg2_z40_r1400")
    # E = 1720 bits of redundancy version 0 reach codeword position 1799: 35 of 42 block rows are needed
    r = rm.RateMatch(base_graph=2, Z=40, E=1720)
    assert r.offset == 80 and r.k0 == 0 and r.highest_position() == 1799
    assert r.rows_needed() == 35 and r.n_active() == 1800
    act = rm.active_submatrix(full.qc, r.rows_needed())
    H = full.dense()
    assert np.array_equal(act.expand(), H[:1400, :1800])           # channel.py:35 ...
    assert not H[:1400, 1800:].any()                               # ... whose other columns are empty
    assert np.array_equal(cut.dense(), H[:1400])
    # the punctured 2 Z positions come back as LLR 0.0 (channel.py:412), the rest as received
    llr = np.arange(1.0, 1721.0)[None, :]
    out = r.recover_host(llr, 1800, filler_llr=99.0)
    assert not out[0, :80].any() and np.array_equal(out[0, 80:], llr[0])


def test_base_graph_and_lifting_selection():
    assert len(rm.LIFTING_SIZES) == 51 and rm.LIFTING_SIZES[0] == 2 and rm.LIFTING_SIZES[-1] == 384
    assert rm.select_base_graph(8448, 0.9) == 1 and rm.select_base_graph(200, 0.9) == 2
    assert rm.select_base_graph(3000, 0.5) == 2 and rm.select_base_graph(3000, 0.8) == 1
    assert rm.select_base_graph(6000, 0.2) == 2
    assert rm.select_lifting(8448, 1) == (384, 8448, 0)
    assert rm.select_lifting(8000, 1) == (384, 8448, 448)
    assert rm.select_lifting(3840, 2) == (384, 3840, 0)
    assert rm.select_lifting(400, 2) == (52, 520, 120)             # K_b = 8 for 192 < B <= 560
    assert rm.select_lifting(100, 2) == (18, 180, 80)              # K_b = 6
    assert rm.lifting_set_index(384) == 1 and rm.lifting_set_index(40) == 2
    with pytest.raises(ValueError):
        rm.select_lifting(9000, 1)
    # lifting a table: shifts modulo the lifting size, connectivity unchanged
    bg1 = codes.load_packaged("nr_bg1_z384").qc
    q96 = rm.lift(bg1, 96)
    assert q96.Z == 96 and np.array_equal(q96.edge_shift, bg1.edge_shift % 96) and np.array_equal(q96.edge_col, bg1.edge_col)


def test_bit_selection_and_interleaving():
    Z = 384
    assert [rm.RateMatch(1, Z, 100, rv=v).k0 for v in range(4)] == [0, 17 * Z, 33 * Z, 56 * Z]
    assert [rm.RateMatch(2, 40, 100, rv=v).k0 for v in range(4)] == [0, 13 * 40, 25 * 40, 43 * 40]
    # a limited buffer moves the starting positions (floor(17 Ncb / (66 Zc)) Zc)
    assert rm.RateMatch(1, Z, 100, rv=1, n_cb=20000).k0 == (17 * 20000 // (66 * Z)) * Z
    # the closed form equals the standard's loop
    for case in (rm.RateMatch(2, 8, 500, rv=3, fillers=13), rm.RateMatch(1, 96, 5002, rv=1, fillers=112, n_cb=4000),
                 rm.RateMatch(2, 40, 1720), rm.RateMatch(1, 12, 3000, rv=2, fillers=200)):
        assert np.array_equal(case.positions(), case.positions_by_walk()), case
    # fillers are skipped, the walk wraps
    r = rm.RateMatch(2, 8, 500, rv=3, fillers=13)
    pos = r.positions()
    fa, fb = r.filler_range
    assert (fa, fb) == (80 - 13 - 16, 80 - 16) and pos[0] == r.k0 == 43 * 8
    assert not ((pos >= fa) & (pos < fb)).any() and pos.max() < r.n_cb == 400
    per_lap = r.n_cb - 13
    assert np.array_equal(pos[per_lap:], pos[:500 - per_lap])      # second lap repeats the first
    # interleaver: f[i + j Qm] = e[i E/Qm + j]
    r4 = rm.RateMatch(2, 8, 12, Qm=4)
    e = np.arange(12)[None, :]
    cw = np.zeros((1, r4.n), np.int64)
    cw[0, 16:28] = e
    f = r4.match_host(cw)[0]
    assert f.tolist() == [0, 3, 6, 9, 1, 4, 7, 10, 2, 5, 8, 11]
    # recover undoes match where a position was sent once, and adds up repeats
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2, size=(3, r.n))
    sent = r.match_host(bits)
    back = r.recover_host(1.0 - 2.0 * sent, r.n, filler_llr=50.0)
    times = np.bincount(pos + r.offset, minlength=r.n)
    assert np.array_equal(back[:, times > 0], ((1.0 - 2.0 * bits) * times)[:, times > 0])
    assert (back[:, fa + 16:fb + 16] == 50.0).all() and not back[:, :16].any()


@pytest.mark.gpu
def test_device_kernels_equal_host_restatement():
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1)
    cases = [rm.RateMatch(2, 40, 1720), rm.RateMatch(2, 40, 4000, rv=2, fillers=37, Qm=4),
             rm.RateMatch(1, 384, 16896, rv=0), rm.RateMatch(1, 384, 30000, rv=3, fillers=448, Qm=6),
             rm.RateMatch(1, 96, 5002, rv=1, Qm=2, n_cb=4000), rm.RateMatch(2, 2, 7, rv=1, fillers=3)]
    for r in cases:
        frames = 9
        cw = rng.integers(0, 2, size=(frames, r.n), dtype=np.uint8)
        got = r.match_device(torch.as_tensor(cw, device=dev)).cpu().numpy()
        assert np.array_equal(got, r.match_host(cw)), r
        llr = rng.standard_normal((frames, r.E)).astype(np.float32)
        for n_out in (r.n, r.n_active()):
            for tdt in (torch.float32, torch.float64):
                want = r.recover_host(llr, n_out, filler_llr=25.0)
                got = r.recover_device(torch.as_tensor(llr, device=dev), n_out, 25.0, out_dtype=tdt).cpu().numpy()
                assert np.allclose(got, want, rtol=1e-6, atol=1e-6), (r, n_out)
        # soft combining with a second transmission of another redundancy version
        r2 = rm.RateMatch(r.base_graph, r.Z, r.E, rv=(r.rv + 2) % 4, fillers=r.fillers, Qm=r.Qm, n_cb=r.n_cb)
        llr2 = rng.standard_normal((frames, r.E)).astype(np.float32)
        buf = r.recover_device(torch.as_tensor(llr, device=dev), r.n, 25.0, out_dtype=torch.float64)
        r2.recover_device(torch.as_tensor(llr2, device=dev), r.n, 25.0, into=buf)
        want = r2.recover_host(llr2, r.n, 25.0, into=r.recover_host(llr, r.n, 25.0))
        assert np.allclose(buf.cpu().numpy(), want, rtol=1e-6, atol=1e-6), r
    from ldpc_amd import _lib
    with pytest.raises(_lib.LdpcError):
        rm.RateMatch(2, 40, 1720).recover_device(torch.zeros((1, 1720), device=dev), 0)


@pytest.mark.gpu
def test_reference_truncation_is_the_rate_matched_decode(c_oracle):
    """Decoding the rate-recovered LLRs on the active sub-matrix gives what the reference's setup
    gives (first 1400 rows, first 80 LLRs overwritten with 1e-9 as in bpsk.py:163): same hard decisions on the 1800
    columns the rows touch, same iteration counts - rule R1, float64, bit for bit."""
    import torch
    from ldpc_amd.decoder import Decoder
    dev = torch.device("cuda", 0)
    cut = codes.load_packaged("nr_bg2_z40_r1400")
    r = rm.RateMatch(2, 40, 1720)
    act = codes.compile_qc(rm.active_submatrix(codes.load_packaged("nr_bg2_z40").qc, r.rows_needed()), "bg2_active35")
    rng = np.random.default_rng(3)
    frames = 300
    sigma2 = 1.0 / (2.0 * 10 ** (-3.0 / 10))
    y = 1.0 + np.sqrt(sigma2) * rng.standard_normal((frames, r.E))     # all-zero codeword, bit 0 sent as +1
    llr_air = (2.0 * y / sigma2).astype(np.float32)
    got = r.recover_device(torch.as_tensor(llr_air, device=dev), r.n_active(), out_dtype=torch.float64).cpu().numpy()
    got[:, :80] = 1e-9                                                  # bpsk.py:163's value for the punctured prefix
    a = Decoder(act).decode(got, config.R1, compute_dtype=np.float64)
    full = np.zeros((frames, cut.n))
    full[:, :1800] = got
    b = Decoder(cut).decode(full, config.R1, compute_dtype=np.float64)
    assert np.array_equal(a.iters, b.iters) and np.array_equal(a.converged, b.converged)
    assert np.array_equal(a.hard, b.hard[:, :1800])
    eh, ei, ec, _ = c_oracle.c_decode(cut, config.R1, full, want_posterior=False)
    assert np.array_equal(ei, a.iters) and np.array_equal(eh[:, :1800], a.hard)
    assert 0 < a.converged.sum() < frames


@pytest.mark.gpu
def test_rate_matched_link_with_retransmission():
    """BG1-shaped code at Z = 96: encode on the device, send E bits of redundancy version 0, recover,
    decode on the rows that matter; a second transmission (version 2) combined into the same buffer
    rescues frames the first one lost."""
    import torch
    from ldpc_amd import encoder, sweep
    from ldpc_amd.decoder import Decoder
    dev = torch.device("cuda", 0)
    Z, K, fillers = rm.select_lifting(2000, 1)
    assert (Z, K, fillers) == (96, 2112, 112)
    qc = rm.lift(codes.load_packaged("nr_bg1_z384").qc, Z)
    code = codes.compile_qc(qc, "bg1_z96")
    enc = encoder.device_encoder(code)
    frames = 2000
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    msg = torch.randint(0, 2, (frames, K), generator=gen, device=dev, dtype=torch.uint8)
    msg[:, K - fillers:] = 0                                      # filler bits are zeros
    cw = enc(msg)
    E = 3000                                                      # code rate 2000 / 3000
    cfg = config.LNMS.with_(max_iters=20)
    tx = {}
    for rv in (0, 2):
        r = rm.RateMatch(1, Z, E, rv=rv, fillers=fillers, Qm=2)
        bt = sweep.noisy_batch(code, r.match_device(cw), -0.3, cfg, [6, rv], dev)   # the channel sees E bits per frame
        tx[rv] = (r, bt.llr)
    r0, llr0 = tx[0]
    n_act = r0.n_active()
    assert n_act < code.n
    act = codes.compile_qc(rm.active_submatrix(qc, r0.rows_needed()), "bg1_z96_active")
    buf = r0.recover_device(llr0, n_act, filler_llr=40.0)
    a = Decoder(act).decode(buf.cpu().numpy(), cfg, compute_dtype=np.float32)
    assert a.kernel == "bg_lnms_f32"   # the rate-matched sub-matrix runs on the base-graph kernel, not on the table-driven one
    ref = cw.cpu().numpy()
    ok1 = a.converged & (a.hard[:, :K] == ref[:, :K]).all(axis=1)
    # second transmission: more rows become active, the first buffer is widened and combined into
    r2, llr2 = tx[2]
    n2 = max(n_act, r2.n_active())
    wide = torch.zeros((frames, n2), dtype=torch.float32, device=dev)
    r0.recover_device(llr0, n2, 40.0, into=wide)
    r2.recover_device(llr2, n2, 40.0, into=wide)
    act2 = codes.compile_qc(rm.active_submatrix(qc, (n2 - K) // Z), "bg1_z96_active2")
    b = Decoder(act2).decode(wide.cpu().numpy(), cfg, compute_dtype=np.float32)
    assert b.kernel == "bg_lnms_f32"
    ok2 = b.converged & (b.hard[:, :K] == ref[:, :K]).all(axis=1)
    assert 0.02 < 1 - ok1.mean() < 0.98, ok1.mean()               # one transmission: in the waterfall
    assert ok2.mean() > 0.999 and ok2.sum() > ok1.sum()
    assert not (a.converged & ~ok1).any()                         # no undetected errors
