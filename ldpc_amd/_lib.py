"""ctypes binding of libldpc_amd.so (the C ABI in include/ldpc_amd.h).

Loading fails loudly: there is no CPU fallback behind this module.
"""
from __future__ import annotations

import ctypes
import os

from .config import CDecodeConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
#: LDPC_AMD_LIB names another build of the same library (kernel experiments); default: in-tree
LIB_PATH = os.environ.get("LDPC_AMD_LIB") or os.path.join(_HERE, "libldpc_amd.so")

LDPC_OK, LDPC_ERR_ARG, LDPC_ERR_HIP, LDPC_ERR_UNSUPPORTED, LDPC_ERR_NO_DEVICE = range(5)
F32, F64 = 0, 1
KERNEL_AUTO, KERNEL_CSR, KERNEL_QC, KERNEL_GRAPH = 0, 1, 2, 3

#: every symbol include/ldpc_amd.h declares
SYMBOLS = ["ldpc_last_error", "ldpc_abi_version", "ldpc_device_count", "ldpc_code_create_csr",
           "ldpc_code_attach_qc", "ldpc_code_destroy", "ldpc_code_info", "ldpc_decode_batch",
           "ldpc_decode_batch_host", "ldpc_decode_batch_noalloc", "ldpc_noalloc_scratch_bytes", "ldpc_code_workspace_bytes", "ldpc_count_errors", "ldpc_allreduce_counters", "ldpc_channel_llr", "ldpc_channel_llr_doppler", "ldpc_decode_segments", "ldpc_sweep_batch", "ldpc_sweep_batch_packed", "ldpc_pack_bits", "ldpc_encoder_create", "ldpc_encoder_destroy", "ldpc_encode_batch", "ldpc_encode_random", "ldpc_rate_match", "ldpc_rate_recover",
           "ldpc_last_kernel"]


class CSegment(ctypes.Structure):
    """Binary image of ``ldpc_segment`` (include/ldpc_amd.h): one code's share of a mixed batch."""

    _fields_ = [("code", ctypes.c_void_p), ("llr_dev", ctypes.c_void_p), ("frames", ctypes.c_int64),
                ("hard_dev", ctypes.c_void_p), ("iters_dev", ctypes.c_void_p), ("converged_dev", ctypes.c_void_p),
                ("posterior_dev", ctypes.c_void_p), ("stream", ctypes.c_void_p)]


class CChannelConfig(ctypes.Structure):
    """Binary image of ``ldpc_channel_config`` (include/ldpc_amd.h): the channel of a fused sweep."""

    _fields_ = [("sigma2", ctypes.c_double), ("gain", ctypes.c_double), ("prefix_llr", ctypes.c_double),
                ("cfo_cycles_per_symbol", ctypes.c_double), ("phase0_cycles", ctypes.c_double),
                ("seed", ctypes.c_uint64), ("first_frame", ctypes.c_uint64), ("plus_is_bit", ctypes.c_int32),
                ("prefix_len", ctypes.c_int32)]


class CRateMatchConfig(ctypes.Structure):
    """Binary image of ``ldpc_rate_match_config`` (include/ldpc_amd.h)."""

    _fields_ = [(k, ctypes.c_int32) for k in ("n_cb", "k0", "filler_begin", "filler_end", "E", "Qm", "offset", "n_out")]


class LdpcError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"ldpc_amd error {code}: {text}")
        self.code = code


_lib = None


def _share_hip_runtime_with_torch() -> None:
    """PyTorch wheels carry their own copy of the HIP runtime (torch/lib/libamdhip64.so, soname
    libamdhip64.so.7).  Two copies in one process cannot both own the GPU, and device pointers and
    streams are only meaningful inside the runtime that made them.  Loading torch's copy first
    makes this library (which asks for that soname) bind to it, whichever of the two the caller
    imports first; without torch the system runtime under /opt/rocm is used."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m ldpc_amd.build` "
                          "(hipcc, --offload-arch=gfx950); there is no CPU fallback")
    _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    lib.ldpc_last_error.restype = ctypes.c_char_p
    lib.ldpc_last_error.argtypes = []
    lib.ldpc_last_kernel.restype = ctypes.c_char_p
    lib.ldpc_last_kernel.argtypes = []
    lib.ldpc_abi_version.restype = i32
    lib.ldpc_abi_version.argtypes = []
    lib.ldpc_device_count.restype = i32
    lib.ldpc_device_count.argtypes = [ctypes.POINTER(i32)]
    lib.ldpc_code_create_csr.restype = i32
    lib.ldpc_code_create_csr.argtypes = [i32, i32, vp, vp, ctypes.POINTER(vp)]
    lib.ldpc_code_attach_qc.restype = i32
    lib.ldpc_code_attach_qc.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp]
    lib.ldpc_code_destroy.restype = i32
    lib.ldpc_code_destroy.argtypes = [vp]
    lib.ldpc_code_info.restype = i32
    lib.ldpc_code_info.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32),
                                   ctypes.POINTER(i64), ctypes.POINTER(i32)]
    lib.ldpc_decode_batch.restype = i32
    lib.ldpc_decode_batch.argtypes = [vp, ctypes.POINTER(CDecodeConfig), vp, i32, i32, i32, i64,
                                      vp, vp, vp, vp, vp]
    lib.ldpc_decode_batch_noalloc.restype = i32
    lib.ldpc_decode_batch_noalloc.argtypes = lib.ldpc_decode_batch.argtypes
    lib.ldpc_noalloc_scratch_bytes.restype = i64
    lib.ldpc_noalloc_scratch_bytes.argtypes = []
    lib.ldpc_code_workspace_bytes.restype = i64
    lib.ldpc_code_workspace_bytes.argtypes = [vp]
    lib.ldpc_decode_batch_host.restype = i32
    lib.ldpc_decode_batch_host.argtypes = [vp, ctypes.POINTER(CDecodeConfig), vp, i32, i32, i32,
                                           i64, vp, vp, vp, vp]
    lib.ldpc_count_errors.restype = i32
    lib.ldpc_count_errors.argtypes = [vp, vp, i64, vp, vp, i64, i32, vp, vp]
    lib.ldpc_channel_llr.restype = i32
    lib.ldpc_channel_llr.argtypes = [vp, i64, i64, i32, ctypes.c_double, ctypes.c_double, i32, i32,
                                     ctypes.c_double, ctypes.c_uint64, ctypes.c_uint64, vp, i32, vp]
    lib.ldpc_channel_llr_doppler.restype = i32
    lib.ldpc_channel_llr_doppler.argtypes = [vp, i64, i64, i32, ctypes.c_double, ctypes.c_double, i32, i32,
                                             ctypes.c_double, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_double,
                                             ctypes.c_double, vp, i32, vp]
    lib.ldpc_encode_random.restype = i32
    lib.ldpc_encode_random.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, i64, vp, vp, vp]
    lib.ldpc_allreduce_counters.restype = i32
    lib.ldpc_allreduce_counters.argtypes = [vp, vp, i32, vp]
    lib.ldpc_encoder_create.restype = i32
    lib.ldpc_encoder_create.argtypes = [i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp, i32, ctypes.POINTER(vp)]
    lib.ldpc_encoder_destroy.restype = None
    lib.ldpc_encoder_destroy.argtypes = [vp]
    lib.ldpc_encode_batch.restype = i32
    lib.ldpc_encode_batch.argtypes = [vp, vp, i64, vp, vp]
    lib.ldpc_rate_match.restype = i32
    lib.ldpc_rate_match.argtypes = [ctypes.POINTER(CRateMatchConfig), vp, i64, i64, vp, vp]
    lib.ldpc_rate_recover.restype = i32
    lib.ldpc_rate_recover.argtypes = [ctypes.POINTER(CRateMatchConfig), vp, i32, i64, ctypes.c_double, i32, vp, i32, vp]
    lib.ldpc_sweep_batch.restype = i32
    lib.ldpc_sweep_batch.argtypes = [vp, ctypes.POINTER(CDecodeConfig), ctypes.POINTER(CChannelConfig), vp, i64, i64, i32, vp, vp]
    lib.ldpc_sweep_batch_packed.restype = i32
    lib.ldpc_sweep_batch_packed.argtypes = lib.ldpc_sweep_batch.argtypes
    lib.ldpc_pack_bits.restype = i32
    lib.ldpc_pack_bits.argtypes = [vp, i64, i64, i32, vp, i64, vp]
    lib.ldpc_decode_segments.restype = i32
    lib.ldpc_decode_segments.argtypes = [ctypes.POINTER(CSegment), i32, ctypes.POINTER(CDecodeConfig), i32, i32, i32]
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != LDPC_OK:
        raise LdpcError(rc, load().ldpc_last_error().decode("utf-8", "replace"))


def last_kernel() -> str:
    return load().ldpc_last_kernel().decode()
