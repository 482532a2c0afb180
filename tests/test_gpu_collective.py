"""ldpc_allreduce_counters: the counters of the ranks summed through RCCL from the C ABI (the
reference is one process; its FER loop adds into Python integers, ldpc.py:660-667).  A one-GPU box
can only hold a communicator of one rank, so this checks the plumbing - RCCL found in the process,
the call enqueued on the stream, counters intact; the two-rank sum is covered with gloo on the CPU
(tests/test_host_logic.py) and with two processes on one GPU (tests/test_aa_gpu_two_rank_sweep.py)."""
import ctypes
import glob
import os

import numpy as np
import pytest

from ldpc_amd import _lib

pytestmark = pytest.mark.gpu


class UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


def _rccl():
    import torch
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["/opt/rocm/lib/librccl.so.1"]
    for path in cands:
        try:
            return ctypes.CDLL(path)
        except OSError:
            continue
    pytest.skip("no RCCL library on this machine")


def test_allreduce_counters_single_rank_communicator():
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib = _lib.load()
    counters = torch.tensor([7, 1, 0, 33, 2 ** 40 + 5], dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream or None
    # no communicator: one rank, nothing to do
    assert lib.ldpc_allreduce_counters(None, counters.data_ptr(), 5, stream) == 0
    rccl = _rccl()
    uid = UniqueId()
    rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        _lib.check(lib.ldpc_allreduce_counters(comm, counters.data_ptr(), 5, stream))
        torch.cuda.synchronize()
        assert counters.cpu().numpy().tolist() == [7, 1, 0, 33, 2 ** 40 + 5]
        assert lib.ldpc_allreduce_counters(comm, None, 5, stream) == _lib.LDPC_ERR_ARG
    finally:
        rccl.ncclCommDestroy(comm)
