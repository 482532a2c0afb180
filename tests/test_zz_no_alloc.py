"""ldpc_decode_batch_noalloc against ldpc_decode_batch, through the C ABI, for the two kernel
families that keep frame state in global scratch memory; two calls in a row, the second smaller.  And the
call captured into a HIP graph (one of the callers it exists for) and replayed."""
import numpy as np
import pytest

from ldpc_amd import _lib, codes, config
from ldpc_amd.decoder import Decoder

pytestmark = pytest.mark.gpu


def _run(dec, L, cfg, compute, kernel, noalloc):
    import torch

    dev = torch.device("cuda", 0)
    B, n = L.shape
    llr = torch.as_tensor(L, device=dev)
    hard = torch.full((B, n), 7, dtype=torch.uint8, device=dev)
    iters = torch.full((B,), -1, dtype=torch.int32, device=dev)
    conv = torch.full((B,), 7, dtype=torch.uint8, device=dev)
    post = torch.zeros((B, n), dtype=torch.float64 if compute == np.float64 else torch.float32, device=dev)
    name = dec.decode_device(llr.data_ptr(), L.dtype, B, hard.data_ptr(), iters.data_ptr(), conv.data_ptr(), cfg,
                             compute_dtype=compute, posterior_ptr=post.data_ptr(),
                             stream=torch.cuda.current_stream().cuda_stream, kernel=kernel, noalloc=noalloc)
    torch.cuda.synchronize()
    return name, hard.cpu().numpy(), iters.cpu().numpy(), conv.cpu().numpy(), post.cpu().numpy()


@pytest.mark.parametrize("rule,kernel,compute,expect,first", [
    ("R1", _lib.KERNEL_CSR, np.float64, "csr_global_all_f64", 40),      # 14 slices of the array for 40 frame groups
    ("R0", _lib.KERNEL_CSR, np.float32, "csr_global_all_f32", 40),
    ("LNMS", _lib.KERNEL_QC, np.float32, "qc_lnms_global_f32", 40),
    ("LNMS", _lib.KERNEL_QC, np.float32, "qc_lnms_global_f32", 400),    # 158 slices of 212 KB for 400 frame groups
])
def test_noalloc_matches_decode_batch(rule, kernel, compute, expect, first):
    code = codes.load_packaged("nr_bg1_z384")
    dec = Decoder(code)
    cfg = config.PRESETS[rule].with_(max_iters=6)
    rng = np.random.default_rng(5)
    assert _lib.load().ldpc_noalloc_scratch_bytes() == 32 << 20
    for frames in (first, 9):       # the second call is smaller and reuses the array as the first left it
        L = (2.2 * (1.0 + 0.6 * rng.standard_normal((frames, code.n)))).astype(np.float32)
        if cfg.plus_is_bit:
            L = -L
        ref = _run(dec, L, cfg, compute, kernel, noalloc=False)
        got = _run(dec, L, cfg, compute, kernel, noalloc=True)
        assert ref[0] == got[0] == expect
        for a, b in zip(ref[1:], got[1:]):
            assert np.array_equal(a, b)
        assert (ref[2] >= 0).all() and (ref[1] <= 1).all()
        assert rule == "R0" or ref[3].any()   # R0 is too weak a rule for this code to converge in six updates


def test_noalloc_serves_on_chip_kernels_and_refuses_oversize():
    """Kernels without scratch go through unchanged; a frame state beyond the array is refused."""
    code = codes.load_packaged("ccsds512")
    dec = Decoder(code)
    rng = np.random.default_rng(6)
    L = (4.0 * (-1.0 + 0.6 * rng.standard_normal((100, code.n)))).astype(np.float32)
    ref = _run(dec, L, config.R0, np.float64, _lib.KERNEL_AUTO, noalloc=False)
    got = _run(dec, L, config.R0, np.float64, _lib.KERNEL_AUTO, noalloc=True)
    assert ref[0] == got[0] == "spec_r0_f64"
    for a, b in zip(ref[1:], got[1:]):
        assert np.array_equal(a, b)
    # a matrix whose edge-list state for ONE frame exceeds 32 MiB: 2 E doubles with E = 2.4 M edges
    m, dc = 300_000, 8
    n = 600_000
    cols = (np.arange(m)[:, None] * 2 + np.arange(dc)[None, :] * 7919) % n
    cols.sort(axis=1)
    assert (np.diff(cols, axis=1) > 0).all()
    E = m * dc
    big = codes.CompiledCode("big", m, n, np.arange(0, E + 1, dc, dtype=np.int32), cols.reshape(-1).astype(np.int32),
                             np.zeros(n + 1, np.int32), np.zeros(E, np.int32), np.zeros(E, np.int32))
    bdec = Decoder(big)
    Lb = np.ones((1, n), np.float32)
    with pytest.raises(_lib.LdpcError) as err:
        _run(bdec, Lb, config.R1.with_(max_iters=1), np.float64, _lib.KERNEL_CSR, noalloc=True)
    assert err.value.code == _lib.LDPC_ERR_UNSUPPORTED and "allocation-free" in str(err.value)


@pytest.mark.parametrize("name,rule,kernel,compute,expect,frames", [
    ("ccsds512", "R0", _lib.KERNEL_AUTO, np.float64, "spec_r0_f64", 1000),          # state on chip
    ("nr_bg1_z384", "LNMS", _lib.KERNEL_AUTO, np.float32, "spec_lnms_f32", 33),
    ("nr_bg1_z384", "R1", _lib.KERNEL_CSR, np.float64, "csr_global_all_f64", 20),   # state in the fixed scratch array
])
def test_noalloc_under_graph_capture(name, rule, kernel, compute, expect, frames):
    """One warm-up call outside capture (symbol address, kernel attributes), then the call is captured into a
    HIP graph; replays on new LLRs in the same buffers give what ldpc_decode_batch gives."""
    import torch
    dev = torch.device("cuda", 0)
    code = codes.load_packaged(name)
    dec = Decoder(code)
    cfg = config.PRESETS[rule].with_(max_iters=6)
    rng = np.random.default_rng(8)

    def draw():
        L = (2.2 * (1.0 + 0.6 * rng.standard_normal((frames, code.n)))).astype(np.float32)
        return -L if cfg.plus_is_bit else L

    llr = torch.as_tensor(draw(), device=dev)
    hard = torch.zeros((frames, code.n), dtype=torch.uint8, device=dev)
    iters = torch.zeros((frames,), dtype=torch.int32, device=dev)
    conv = torch.zeros((frames,), dtype=torch.uint8, device=dev)

    def call(noalloc, h=hard, i=iters, c=conv):
        return dec.decode_device(llr.data_ptr(), np.float32, frames, h.data_ptr(), i.data_ptr(), c.data_ptr(), cfg,
                                 compute_dtype=compute, stream=torch.cuda.current_stream().cuda_stream, kernel=kernel,
                                 noalloc=noalloc)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        assert call(True) == expect          # warm-up on the capture stream, outside capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        call(True)
    for _ in range(2):
        llr.copy_(torch.as_tensor(draw(), device=dev))
        hard.fill_(9); iters.fill_(-1); conv.fill_(9)
        graph.replay()
        torch.cuda.synchronize()
        h2, i2, c2 = torch.empty_like(hard), torch.empty_like(iters), torch.empty_like(conv)
        call(False, h2, i2, c2)
        torch.cuda.synchronize()
        assert torch.equal(hard, h2) and torch.equal(iters, i2) and torch.equal(conv, c2)
        assert (hard <= 1).all() and (iters >= 0).all()
