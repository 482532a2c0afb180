"""ldpc_decode_batch_noalloc against ldpc_decode_batch, through the C ABI, for the two kernel
families that keep frame state in global scratch memory; two calls in a row, the second smaller."""
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
