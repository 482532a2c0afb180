"""The circulant-schedule encoder (ldpc_encode_batch), which stands in for the reference's dense
generator product `numpy.dot(self.G, message) % 2` (/root/reference/ldpc.py:46, G from
pyldpc.coding_matrix, ldpc.py:369).  Nothing in the reference pins a generator (PARITY UNPINNED:
pyldpc is absent and unversioned), so the checks are: every output is a codeword of the reference's
matrix, the message sits unchanged in the information positions, the map is linear, and it equals
the generator the reference's own CCSDS file carries (LDPC_CCSDS_256.mat, key "G") and this build's
host encoder of the 5G-structured table."""
import numpy as np
import pytest

from ldpc_amd import codes, encoder, nr_bg1

NAMES = ["ccsds512", "wifi648", "nr_bg2_z40", "nr_bg1_z384"]


def _schedule(name):
    code = codes.load_packaged(name)
    G = codes.load_packaged_generator(name) if name in ("ccsds512", "wifi648") else None
    return code, G, encoder.schedule_for(code, G)


@pytest.mark.parametrize("name", NAMES)
def test_schedules_produce_codewords(name):
    code, G, s = _schedule(name)
    assert s is not None and s.n == code.n and s.k == code.n - code.m
    rng = np.random.default_rng(1)
    msgs = rng.integers(0, 2, size=(6, s.k), dtype=np.uint8)
    cw = s.run_host(msgs)
    assert all(code.is_codeword(c) for c in cw)
    # systematic: information block j of the message is block column info_cols[j] of the codeword
    Z = s.Z
    for j, c in enumerate(s.info_cols):
        assert np.array_equal(cw[:, c * Z:(c + 1) * Z], msgs[:, j * Z:(j + 1) * Z])
    # linear
    assert np.array_equal(s.run_host(msgs[:1] ^ msgs[1:2]), cw[:1] ^ cw[1:2])
    # every step reads only columns of lower levels
    known = {c: -1 for c in s.info_cols}
    for out, lv, terms in zip(s.step_out, s.step_level, s.step_terms):
        assert all(known[c] < lv for c, _ in terms)
        known[out] = lv
    if name == "ccsds512":      # the generator in the reference's own .mat file
        assert np.array_equal(cw, codes.encode(G, msgs))
    if name == "nr_bg1_z384":   # this build's host encoder of the same table
        assert np.array_equal(cw, nr_bg1.encode(code.qc, msgs))


def test_no_schedule_where_the_structure_is_missing():
    # the row-truncated 5G matrix of channel.py:35 has unconstrained columns: no parity structure
    code = codes.load_packaged("nr_bg2_z40_r1400")
    assert encoder.schedule_for(code) is None
    # a generator that is not block-circulant
    c2 = codes.load_packaged("wifi648")
    assert encoder.generator_schedule(codes.load_packaged_generator("wifi648"), c2.qc.Z) is None


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_encoder_equals_host_schedule(name):
    import torch
    dev = torch.device("cuda", 0)
    code, G, s = _schedule(name)
    enc = encoder.DeviceEncoder(s)
    rng = np.random.default_rng(2)
    for frames in (1, 7, 300 if code.n > 4096 else 5000):
        msgs = rng.integers(0, 2, size=(frames, s.k), dtype=np.uint8)
        cw = enc(torch.as_tensor(msgs, device=dev)).cpu().numpy()
        assert np.array_equal(cw, s.run_host(msgs)), (name, frames)
    assert enc(torch.zeros((0, s.k), dtype=torch.uint8, device=dev)).shape == (0, code.n)


@pytest.mark.gpu
def test_encoder_feeds_the_sweep_and_rejects_bad_schedules():
    import ctypes
    import torch
    from ldpc_amd import _lib, config, sweep
    dev = torch.device("cuda", 0)
    code, G, s = _schedule("nr_bg1_z384")
    enc = encoder.device_encoder(code)
    cw = sweep.make_codewords(code, enc, 700, [5], dev)
    assert cw.shape == (700, code.n) and cw.any()
    host = cw.cpu().numpy()
    assert all(code.is_codeword(c) for c in host[:20])
    # random codewords decode like the all-zero word (the rule is symmetric; the noise realisations differ
    # in effect, so the agreement is statistical): same convergence, mean sweep count within 3 %
    out = sweep.ber_sweep(code, enc, config.LNMS, [-3.0], 2000, dev, np.float32, seed=8)[0]
    zero = sweep.ber_sweep(code, None, config.LNMS, [-3.0], 2000, dev, np.float32, seed=8)[0]
    assert out["not_converged"] == zero["not_converged"] == 0 and out["bit_errors"] == 0
    assert abs(out["iterations"] - zero["iterations"]) < 0.03 * zero["iterations"]
    # a step that reads a column of its own level is refused
    lib = _lib.load()
    h = ctypes.c_void_p()
    i32 = lambda a: np.asarray(a, np.int32)
    info, outs, ptr, lv, tc, ts = i32([0]), i32([1, 2]), i32([0, 1, 2]), i32([0, 0]), i32([0, 1]), i32([0, 0])
    rc = lib.ldpc_encoder_create(4, 3, 3, 1, info.ctypes.data, 2, outs.ctypes.data, ptr.ctypes.data, lv.ctypes.data,
                                 tc.ctypes.data, ts.ctypes.data, 0, ctypes.byref(h))
    assert rc == _lib.LDPC_ERR_ARG and b"before it is known" in lib.ldpc_last_error()
    lv = i32([0, 1])
    rc = lib.ldpc_encoder_create(4, 3, 3, 1, info.ctypes.data, 2, outs.ctypes.data, ptr.ctypes.data, lv.ctypes.data,
                                 tc.ctypes.data, ts.ctypes.data, 0, ctypes.byref(h))
    assert rc == 0
    msg = torch.tensor([[1, 0, 1, 1]], dtype=torch.uint8, device=dev)
    out = torch.empty((1, 12), dtype=torch.uint8, device=dev)
    assert lib.ldpc_encode_batch(h, msg.data_ptr(), 1, out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert out.cpu().numpy().tolist() == [[1, 0, 1, 1] * 3]
    lib.ldpc_encoder_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ccsds512", "wifi648", "nr_bg1_z384"])
def test_random_messages_on_the_device(name):
    """ldpc_encode_random: the messages are this build's Philox stream (numpy restatement in
    oracle/channel_oracle.py), a function of (seed, absolute frame, bit) - so two launches that split
    a range of frames give what one launch gives - and the codewords are their encodings."""
    import torch
    from oracle import channel_oracle as co
    dev = torch.device("cuda", 0)
    code, G, s = _schedule(name)
    enc = encoder.DeviceEncoder(s)
    frames = 37 if code.n > 4096 else 1001
    cw, msg = enc.random_codewords(frames, [9, 4], first_frame=5, device=dev, want_messages=True)
    key = int(np.random.SeedSequence([9, 4]).generate_state(1, dtype=np.uint64)[0])
    want = co.random_message_bits(frames, s.k, key, first_frame=5)
    assert np.array_equal(msg.cpu().numpy(), want)
    assert np.array_equal(cw.cpu().numpy(), s.run_host(want))
    assert 0.45 < want.mean() < 0.55
    a = enc.random_codewords(frames - 10, [9, 4], first_frame=5, device=dev)
    b = enc.random_codewords(10, [9, 4], first_frame=5 + frames - 10, device=dev)
    assert torch.equal(torch.cat([a, b]), cw)
    # a sweep fed by the device encoder does not depend on how it is cut into launches
    if name == "ccsds512":
        from ldpc_amd import config, sweep
        one = sweep.ber_sweep(code, enc, config.R1, [0.0], 50_000, dev, np.float32, seed=3, batch=1 << 18)
        cut = sweep.ber_sweep(code, enc, config.R1, [0.0], 50_000, dev, np.float32, seed=3, batch=1 << 14)
        two = [sweep.ber_sweep(code, enc, config.R1, [0.0], 50_000, dev, np.float32, seed=3, rank=r, world=2, reduce=False)[0] for r in (0, 1)]
        assert one == cut
        for k in ("frames", "not_converged", "bit_errors", "iterations"):
            assert one[0][k] == two[0][k] + two[1][k]
        assert 0 < one[0]["not_converged"] < 50_000


@pytest.mark.gpu
@pytest.mark.parametrize("bg,Z", [(1, 8), (1, 12), (1, 20), (1, 36), (1, 96), (1, 104), (1, 208), (1, 320), (2, 8), (2, 28), (2, 64),
                                  (2, 384), (1, 6), (1, 9), (2, 15)])
def test_device_encoder_over_lifting_sizes(bg, Z):
    """Both base graphs at lifting sizes that are multiples of four (the word-wide kernel: sizes below one wave,
    sizes that are not multiples of eight or of 64, several frames per workgroup) and sizes that are not (the
    byte-wide kernel): device encoder == host schedule, every word a codeword, ragged frame counts."""
    import torch
    from ldpc_amd import nr_rate_match as rm
    dev = torch.device("cuda", 0)
    base = codes.load_packaged("nr_bg1_z384" if bg == 1 else "nr_bg2_z40").qc
    code = codes.compile_qc(rm.lift(base, Z), f"bg{bg}_z{Z}")
    s = encoder.schedule_for(code, None)
    assert s is not None
    enc = encoder.DeviceEncoder(s)
    rng = np.random.default_rng(Z)
    for frames in (1, 3, 257):
        msgs = rng.integers(0, 2, size=(frames, s.k), dtype=np.uint8)
        msgs[0, :] = 1
        cw = enc(torch.as_tensor(msgs, device=dev)).cpu().numpy()
        assert np.array_equal(cw, s.run_host(msgs)), (bg, Z, frames)
        assert all(code.is_codeword(c) for c in cw[:5])
    # messages drawn on the device: the same stream whatever kernel serves the size, and msg_out is the message
    cw, msg = enc.random_codewords(41, [3, Z], first_frame=2, device=dev, want_messages=True)
    assert np.array_equal(cw.cpu().numpy(), s.run_host(msg.cpu().numpy()))
    # message bytes with upper bits set count by their low bit; rows off word alignment take the byte-wide kernel
    raw = rng.integers(0, 256, size=(9, s.k), dtype=np.uint8)
    cw = enc(torch.as_tensor(raw, device=dev)).cpu().numpy()
    assert np.array_equal(cw, s.run_host(raw & 1))
    flat = torch.zeros(9 * s.k + 4, dtype=torch.uint8, device=dev)
    flat[1:1 + 9 * s.k] = torch.as_tensor(raw.reshape(-1), device=dev)
    out = torch.empty((9, s.n), dtype=torch.uint8, device=dev)
    enc.encode_ptr(flat.data_ptr() + 1, 9, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), cw)
