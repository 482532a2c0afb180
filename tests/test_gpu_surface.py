"""The reference-shaped surface, the loop-bound quirks, the error accounting and the edge cases of
the C ABI on a real GPU."""
import ctypes

import numpy as np
import pytest

from conftest import golden_files, load_golden
from ldpc_amd import codes, config
from oracle import bp_oracle

pytestmark = pytest.mark.gpu


def _first(rule, code):
    for p in golden_files(rule):
        g = load_golden(p)
        if g["code"] == code:
            yield g


@pytest.mark.parametrize("cls_name,method,rule,code,ctor", [
    ("LdpcPyEncoder", "minSumDecode", "R0", "wifi648", (4, 8, 648)),
    ("BpskPyEncoder", "minSumDecode", "R1", "wifi648", (4, 8, 648)),
    ("BpskPyEncoder", "sumProductDecodeTest", "R2", "wifi648", (4, 8, 648)),
    ("ChannelPyEncoder", "minSumDecode", "R1", "nr_bg2_z40_r1400", (4, 5, 2080)),
    ("ChannelPyEncoder", "sumProductDecodeTest", "R3", "nr_bg2_z40_r1400", (4, 5, 2080)),
    ("ChannelPyEncoder", "sumProductDecode", "R4", "nr_bg2_z40_r1400", (4, 5, 2080)),
])
def test_reference_call_surface(cls_name, method, rule, code, ctor):
    """Frame by frame through the LDPCEncoder-shaped classes, as the reference's drivers call
    them (ldpc.py:392, bpsk.py:824, channel.py:845): return value, numIterations, messageDecoded."""
    from ldpc_amd import shim
    enc = getattr(shim, cls_name)(*ctor, readDataMatrix=True)
    checked = 0
    for g in _first(rule, code):
        for f in range(min(4, len(g["ret"]))):
            enc.SNR = float(g["snr_attr"][f])
            enc.originalEncoded = g["codeword"][f].astype(int)
            args = (g["y"][f].astype(np.float64),) + ((2,) if rule == "R0" else ())
            ret = getattr(enc, method)(*args)
            assert enc.numIterations == g["iters"][f]
            if g["converged"][f]:
                assert ret == g["ret"][f]
                assert np.array_equal(np.asarray(enc.messageDecoded, dtype=float), g["hard"][f])
            else:
                assert ret is shim.FRAME_ERROR
                if rule in ("R0", "R1"):
                    assert np.array_equal(enc.messageDecoded, g["posterior"][f])
                else:
                    np.testing.assert_allclose(enc.messageDecoded, g["posterior"][f], rtol=1e-4, atol=1e-7)
            checked += 1
    assert checked >= 4


def test_ccsds_caller_shape():
    """What ccsds.py:17-28 sets out to do (its own call is broken against the current ldpc.py):
    CCSDS matrix from the .mat loader, one frame, min-sum decode through the ldpc.py surface."""
    from ldpc_amd import shim
    code = codes.load_packaged("ccsds512")
    enc = shim.LdpcPyEncoder(4, 8, 512, H=code.dense())
    np.random.seed(12)
    msg = np.random.randint(0, 2, size=enc.G.shape[1])
    enc.originalEncoded = np.dot(enc.G, msg) % 2
    y = enc.addNoiseBPSK(3.0, enc.originalEncoded)
    ber = enc.minSumDecode(y, 2)
    L = bp_oracle.llr_from_samples(y, 3.0, 0)
    eh, ei, ec, _ = bp_oracle.numpy_decode(code, config.R0, L)
    assert ec[0] and ber == (eh[0] != enc.originalEncoded).mean() and enc.numIterations == ei[0]


def test_loop_bound_quirks_on_gpu(c_oracle):
    """self.m / self.n taken from the constructor instead of the matrix (ldpc.py:30,175; bpsk.py:244):
    served by the table-driven circulant kernels and by the edge-list kernel everywhere, and by the
    compiled-in kernels (as template switches) for the reference's own callers: active rows on the CCSDS
    matrix (ccsds.py:18 gives 8) and n = 648 on the 2080-column 5G matrix (channel.py:646) - all identical
    to the CPU restatement, early exit and fixed work, double and float."""
    from ldpc_amd.decoder import Decoder
    rng = np.random.default_rng(21)
    cases = [("wifi648", config.R0.with_(num_active_checks=100)),
             ("wifi648", config.R1.with_(num_refresh_vars=300, max_iters=12)),
             ("wifi648", config.R2.with_(num_refresh_vars=300, max_iters=8)),
             ("wifi648", config.OMS.with_(num_refresh_vars=40, max_iters=8)),
             ("ccsds512", config.R0.with_(num_active_checks=8)),
             ("ccsds512", config.R0.with_(num_active_checks=70)),      # boundary inside a block row
             ("ccsds512", config.R0.with_(num_active_checks=70, early_exit=False, max_iters=6)),
             ("ccsds512", config.R1.with_(num_refresh_vars=100, max_iters=10)),
             ("nr_bg2_z40_r1400", config.R1.with_(num_refresh_vars=648, max_iters=12)),
             ("nr_bg2_z40_r1400", config.R1.with_(num_refresh_vars=648, max_iters=5, early_exit=False)),
             ("nr_bg2_z40", config.R1.with_(num_refresh_vars=1000, max_iters=8)),
             ("nr_bg2_z40_r1400", config.R2.with_(num_refresh_vars=648, max_iters=6)),
             ("nr_bg2_z40_r1400", config.R3.with_(num_refresh_vars=648, max_iters=10)),
             ("nr_bg2_z40_r1400", config.R3.with_(num_refresh_vars=648, max_iters=4, early_exit=False)),
             ("nr_bg2_z40_r1400", config.R4.with_(num_refresh_vars=648, max_iters=10))]
    # which of them run on compiled-in kernels under the automatic choice
    compiled = {("ccsds512", 0): "spec_r0_", ("wifi648", 0): "spec_r0_", ("nr_bg2_z40_r1400", 1): "spec_r1m_", ("nr_bg2_z40", 1): "spec_r1m_",
                ("nr_bg2_z40_r1400", 2): "spec_spaw_", ("nr_bg2_z40_r1400", 3): "spec_spaw_", ("nr_bg2_z40_r1400", 4): "spec_spaw_"}
    decs = {}
    for name, cfg in cases:
        code = codes.load_packaged(name)
        dec = decs.setdefault(name, Decoder(code))
        L = 4.0 * (-1.0 + 0.9 * rng.standard_normal((48, code.n)))
        L = L if cfg.rule == 0 else -L
        eh, ei, ec, ep = c_oracle.c_decode(code, cfg, L)
        for kernel, prefix in ((0, compiled.get((name, cfg.rule), "qc")), (2, "qc"), (1, "csr")):
            res = dec.decode(L, cfg, want_posterior=True, kernel=kernel)
            assert res.kernel.startswith(prefix), (name, cfg.rule, res.kernel)
            assert np.array_equal(res.iters, ei) and np.array_equal(res.converged, ec), (name, cfg.rule, kernel)
            assert np.array_equal(res.hard, eh), (name, cfg.rule, kernel)
            if cfg.rule in (0, 1, 6):
                assert np.array_equal(res.posterior, ep), (name, cfg.rule, kernel)
            else:
                np.testing.assert_allclose(res.posterior, ep, rtol=1e-4, atol=1e-7)
        if (name, cfg.rule) in compiled:   # the float instantiation carries the switch as well
            res = dec.decode(L.astype(np.float32), cfg, compute_dtype=np.float32)
            # (the in-place kernel of the full 5G matrix exists in double only: float stays on the multi-wave kernel)
            assert res.kernel.startswith(compiled[(name, cfg.rule)].replace("spec_r1m_", "spec_r1w_")), res.kernel
            assert (res.iters == ei).mean() > 0.9 and ((res.hard != eh).any(axis=1) & (res.converged & ec)).mean() < 0.05


def test_irregular_matrix_without_circulant_structure(c_oracle):
    from ldpc_amd.decoder import Decoder
    rng = np.random.default_rng(5)
    H = (rng.random((60, 150)) < 0.06).astype(np.uint8)
    H[:, 4] = 0
    H[9] = 0
    H[9, 11] = 1
    code = codes.compile_code(H, "irregular")
    dec = Decoder(code)
    assert not dec.has_qc or code.qc.Z < 60
    L = 3.0 * (rng.standard_normal((128, code.n)) - 0.8)
    for name in ("R0", "R1", "R2", "R3", "R4"):
        cfg = config.PRESETS[name].with_(max_iters=10)
        eh, ei, ec, ep = c_oracle.c_decode(code, cfg, L)
        res = dec.decode(L, cfg, want_posterior=True, kernel=1)
        assert np.array_equal(res.iters, ei) and np.array_equal(res.converged, ec), name
        assert np.array_equal(res.hard, eh), name
        if cfg.rule <= 1:
            assert np.array_equal(res.posterior, ep), name
        else:
            np.testing.assert_allclose(res.posterior, ep, rtol=1e-4, atol=1e-7)


def test_r0_iteration_zero_float_check_on_gpu():
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged("ccsds512")
    dec = Decoder(code)
    L = np.zeros((3, code.n))
    L[1] = -4.1
    L[2] = 1.0                                     # every row sums to 8.0: passes, "decisions" are the LLRs
    for kernel in (0, 1, 2):
        r = dec.decode(L, config.R0, kernel=kernel)
        assert list(r.iters) == [0, 1, 0] and r.converged.all()
        assert (r.hard[0] == 0).all() and (r.hard[1] == 0).all() and (r.hard[2] == 1).all()
        r = dec.decode(L, config.R0.with_(float_syndrome0=False), kernel=kernel)
        assert list(r.iters) == [0, 0, 0]


def test_r0_iteration_zero_float_check_against_oracle(c_oracle):
    """`numpy.dot(H, llr) % 2 == 0` on float LLRs (ldpc.py:158,165): the kernels test "finite and an even
    integer" without fmod; integer-valued, half-integer, huge and tiny LLRs against the CPU restatement
    (which calls fmod) on every kernel family, in both arithmetic types."""
    from ldpc_amd.decoder import Decoder
    rng = np.random.default_rng(77)
    for name in ("ccsds512", "wifi648", "nr_bg2_z40_r1400"):
        code = codes.load_packaged(name)
        dec = Decoder(code)
        L = np.zeros((12, code.n))
        L[0] = 2.0 * rng.integers(-3, 4, code.n)             # every row sum even: passes at iteration 0
        L[1] = L[0]; L[1, 5] += 1.0                           # one odd entry: the rows through it fail
        L[2] = rng.integers(-3, 4, code.n)                    # integer sums of either parity
        L[3] = 0.5 * rng.integers(-6, 7, code.n)              # half-integer sums
        L[4] = 2.0 ** 60 * rng.integers(-2, 3, code.n)        # beyond 2^53 every float is an even integer
        L[5] = 2.0 ** 24 * rng.integers(-2, 3, code.n)
        L[6] = 1e-30 * rng.integers(-2, 3, code.n)            # tiny sums: even only when exactly zero
        L[7] = 4.0
        L[8] = -2.0
        L[9] = 3.0                                            # row sums 3 * degree
        L[10] = 0.0
        L[11] = 2.0 * rng.integers(-3, 4, code.n); L[11, code.n - 1] = 0.25
        L = L.astype(np.float32).astype(np.float64)
        eh, ei, ec, _ = c_oracle.c_decode(code, config.R0, L)
        assert ei[0] == 0 and ec[0] and ei[7] == 0 and ei[10] == 0 and ei[1] > 0
        for kernel in (0, 1, 2):
            for dtype in (np.float64, np.float32):
                r = dec.decode(L.astype(dtype), config.R0, kernel=kernel, compute_dtype=dtype)
                zero_pass = ei == 0
                assert np.array_equal(r.iters == 0, zero_pass), (name, kernel, dtype, r.iters, ei)
                assert np.array_equal(r.hard[zero_pass], eh[zero_pass]), (name, kernel, dtype)
                if dtype is np.float64:
                    assert np.array_equal(r.iters, ei) and np.array_equal(r.converged, ec) and np.array_equal(r.hard, eh)


def test_edge_cases_of_the_abi():
    from ldpc_amd import _lib
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged("wifi648")
    dec = Decoder(code)
    empty = dec.decode(np.zeros((0, code.n)), config.R1)
    assert empty.hard.shape == (0, code.n) and empty.iters.shape == (0,)
    with pytest.raises(ValueError):
        dec.decode(np.zeros((2, code.n - 1)), config.R1)
    with pytest.raises(_lib.LdpcError) as ei:
        dec.decode(np.zeros((1, code.n)), config.R1.with_(max_iters=0))
    assert ei.value.code == _lib.LDPC_ERR_ARG
    with pytest.raises(_lib.LdpcError):
        dec.decode(np.zeros((1, code.n)), config.DecodeConfig(rule=8))
    one = dec.decode(np.full((1, code.n), 5.0), config.R1)       # single frame, all-zero word at once
    assert one.converged[0] and one.iters[0] == 0 and not one.hard.any()
    # a batch that is not a multiple of the frames-per-workgroup of any kernel
    rng = np.random.default_rng(0)
    L = 5.0 + rng.standard_normal((77, code.n))
    for kernel in (0, 1, 2):
        r = dec.decode(L, config.R1, kernel=kernel)
        assert r.converged.all() and not r.hard.any()
    # check degree above the supported maximum is refused at compile time of the code
    wide = np.ones((1, 70), np.uint8)
    with pytest.raises(_lib.LdpcError) as ei:
        Decoder(codes.compile_code(wide, "wide", find_qc=False))
    assert ei.value.code == _lib.LDPC_ERR_UNSUPPORTED


def test_same_input_twice_gives_identical_bytes():
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged("nr_bg2_z40")
    dec = Decoder(code)
    rng = np.random.default_rng(4)
    L = 2.0 * (1.0 + 1.4 * rng.standard_normal((96, code.n)))
    a = dec.decode(L, config.R1, want_posterior=True)
    b = dec.decode(L, config.R1, want_posterior=True)
    assert a.posterior.tobytes() == b.posterior.tobytes() and a.hard.tobytes() == b.hard.tobytes()
    assert np.array_equal(a.iters, b.iters)


def test_error_counters_on_device():
    import torch
    from ldpc_amd import sweep
    from ldpc_amd.decoder import Decoder
    code = codes.load_packaged("ccsds512")
    G = codes.load_packaged_generator("ccsds512")
    dec = Decoder(code)
    cfg = config.R0
    batch = sweep.make_batch(code, G, 3000, 0.5, cfg, seed=[1, 2], device=torch.device("cuda", 0))
    sweep.decode_batch(dec, batch, cfg)
    sweep.count_errors(dec, batch)
    torch.cuda.synchronize()
    hard, ref = batch.hard.cpu().numpy(), batch.ref.cpu().numpy()
    conv, iters = batch.converged.cpu().numpy().astype(bool), batch.iters.cpu().numpy()
    errs = (hard != ref).sum(axis=1)
    expect = [3000, int((~conv).sum()), int((conv & (errs > 0)).sum()), int(errs.sum()), int(iters.sum())]
    assert batch.counters.cpu().tolist() == expect
    assert code.is_codeword(ref).all() and 0 < expect[1] < 3000
    # and the batch agrees with the CPU oracle bit for bit
    eh, ei, ec, _ = bp_oracle.c_decode(code, cfg, batch.llr.cpu().numpy().astype(np.float64))
    assert np.array_equal(eh, hard) and np.array_equal(ei, iters) and np.array_equal(ec, conv)


def test_error_counters_word_lengths_strides_and_byte_values():
    """ldpc_count_errors on random bytes: word lengths that are and are not multiples of four, lengths past one
    four-load round of a wave (1 024 positions), rows shifted off word alignment, one reference word for all
    frames (stride 0), and byte values other than 0 / 1 (the float check of ldpc.py:158 marks a position 2) -
    a position counts when its bytes differ."""
    import torch
    from ldpc_amd import _lib
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(12)
    lib = _lib.load()
    for n, frames, shift, stride0 in ((512, 300, 0, False), (2080, 77, 0, False), (26112, 9, 0, False), (1021, 50, 0, False),
                                      (648, 120, 1, False), (1300, 31, 0, True), (4, 5, 0, False), (1028, 3, 2, True)):
        hard = rng.integers(0, 3, size=(frames, n), dtype=np.uint8)
        hard[:, ::7] = 255
        ref = hard.copy()
        flip = rng.random((frames, n)) < 0.03
        ref[flip] ^= rng.integers(1, 256, size=int(flip.sum()), dtype=np.uint8)
        ref[::3] = hard[::3]                                    # a third of the frames without errors
        if stride0:
            ref = np.repeat(ref[:1], frames, axis=0)
        conv = rng.integers(0, 2, size=frames, dtype=np.uint8)
        iters = rng.integers(0, 30, size=frames, dtype=np.int32)
        flat_h = torch.zeros(frames * n + 8, dtype=torch.uint8, device=dev)
        flat_r = torch.zeros(frames * n + 8, dtype=torch.uint8, device=dev)
        flat_h[shift:shift + frames * n] = torch.as_tensor(hard.reshape(-1), device=dev)
        flat_r[shift:shift + frames * n] = torch.as_tensor(ref.reshape(-1), device=dev)
        ctr = torch.zeros((5,), dtype=torch.int64, device=dev)
        ti, tc = torch.as_tensor(iters, device=dev), torch.as_tensor(conv, device=dev)
        _lib.check(lib.ldpc_count_errors(flat_h.data_ptr() + shift, flat_r.data_ptr() + shift, 0 if stride0 else n, ti.data_ptr(),
                                         tc.data_ptr(), frames, n, ctr.data_ptr(), None))
        torch.cuda.synchronize()
        errs = (hard != ref).sum(axis=1)
        c = conv.astype(bool)
        want = [frames, int((~c).sum()), int((c & (errs > 0)).sum()), int(errs.sum()), int(iters.sum())]
        assert ctr.cpu().tolist() == want, (n, frames, shift, stride0)
