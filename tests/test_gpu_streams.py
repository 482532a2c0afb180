"""Two decodes of one handle in flight on two streams: the kernels that keep frame state in the
handle's global workspace must not share slices (one workspace per stream)."""
import numpy as np
import pytest

from ldpc_amd import _lib, codes, config
from ldpc_amd.decoder import Decoder

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rule,kernel,expect", [("R1", _lib.KERNEL_CSR, "csr_global_all_f32"),
                                                ("LNMS", _lib.KERNEL_QC, "qc_lnms_global_f32")])
def test_two_streams_one_handle(rule, kernel, expect):
    import torch

    dev = torch.device("cuda", 0)
    code = codes.load_packaged("nr_bg1_z384")
    dec = Decoder(code)
    cfg = config.PRESETS[rule].with_(max_iters=8)
    rng = np.random.default_rng(12)
    B = 600   # more frames than resident workgroups: both launches fill the device and overlap
    Ls = [torch.as_tensor((2.0 * (1.0 + 0.5 * rng.standard_normal((B, code.n)))).astype(np.float32), device=dev)
          for _ in range(2)]

    def buffers():
        return (torch.empty((B, code.n), dtype=torch.uint8, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
                torch.empty((B,), dtype=torch.uint8, device=dev))

    def launch(L, out, stream):
        return dec.decode_device(L.data_ptr(), np.float32, B, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), cfg,
                                 compute_dtype=np.float32, stream=stream, kernel=kernel)

    # one after the other on the default stream
    serial = [buffers(), buffers()]
    for L, out in zip(Ls, serial):
        assert launch(L, out, torch.cuda.current_stream().cuda_stream) == expect
        torch.cuda.synchronize()
    # both in flight
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    both = [buffers(), buffers()]
    torch.cuda.synchronize()
    for L, out, st in zip(Ls, both, streams):
        launch(L, out, st.cuda_stream)
    torch.cuda.synchronize()
    for a, b in zip(serial, both):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert int(serial[0][2].sum()) > 0
