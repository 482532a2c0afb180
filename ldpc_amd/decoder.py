"""Batched decoder object over the C ABI.

``Decoder`` owns one compiled-code handle on the current HIP device.  Two call styles:

* ``decode(llr)`` — host numpy arrays in and out (the library stages them through device memory);
* ``decode_device(...)`` — raw device pointers (e.g. ``torch.Tensor.data_ptr()``) and a stream,
  nothing is copied or allocated: this is the throughput path.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from .codes import CompiledCode
from .config import CDecodeConfig, DecodeConfig


@dataclass
class DecodeResult:
    hard: np.ndarray        # uint8 [B, n]   the reference's hardDecisions
    iters: np.ndarray       # int32 [B]      the reference's self.numIterations
    converged: np.ndarray   # bool  [B]      False where the reference returns FRAME_ERROR
    posterior: Optional[np.ndarray] = None  # [B, n] the reference's bitNodes at exit
    kernel: str = ""


def _dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt == np.float32:
        return _lib.F32
    if dt == np.float16:   # arithmetic type only: the packed half-precision fast mode (LDPC_F16)
        return 2
    if dt == np.float64:
        return _lib.F64
    raise TypeError(f"LLRs must be float32 or float64, not {dt}")


class Decoder:
    def __init__(self, code: CompiledCode, use_qc: bool = True):
        self.code = code
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        rp = np.ascontiguousarray(code.row_ptr, np.int32)
        ci = np.ascontiguousarray(code.col_idx, np.int32)
        _lib.check(self._lib.ldpc_code_create_csr(code.m, code.n, rp.ctypes.data, ci.ctypes.data,
                                                  ctypes.byref(self._h)))
        self.has_qc = False
        if use_qc and code.qc is not None:
            q = code.qc
            order = np.lexsort((q.edge_shift, q.edge_col, q.edge_row))
            er = np.ascontiguousarray(q.edge_row[order], np.int32)
            ec = np.ascontiguousarray(q.edge_col[order], np.int32)
            es = np.ascontiguousarray(q.edge_shift[order], np.int32)
            rc = self._lib.ldpc_code_attach_qc(self._h, q.mb, q.nb, q.Z, len(er), er.ctypes.data,
                                               ec.ctypes.data, es.ctypes.data)
            if rc == _lib.LDPC_OK:
                self.has_qc = True
            elif rc != _lib.LDPC_ERR_UNSUPPORTED:
                _lib.check(rc)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.ldpc_code_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def workspace_bytes(self) -> int:
        """Device memory the handle's growable workspaces hold (kernels with frame state in global memory)."""
        return int(self._lib.ldpc_code_workspace_bytes(self._h))

    # ------------------------------------------------------------------ host arrays
    def decode(self, llr: np.ndarray, cfg: DecodeConfig, compute_dtype=np.float64,
               want_posterior: bool = False, kernel: int = _lib.KERNEL_AUTO) -> DecodeResult:
        llr = np.asarray(llr)
        if llr.ndim == 1:
            llr = llr[None, :]
        if llr.ndim != 2 or llr.shape[1] != self.code.n:
            raise ValueError(f"llr must be [frames, {self.code.n}], got {llr.shape}")
        if llr.dtype not in (np.float32, np.float64):
            llr = llr.astype(np.float64)
        llr = np.ascontiguousarray(llr)
        B = llr.shape[0]
        hard = np.zeros((B, self.code.n), np.uint8)
        iters = np.zeros(B, np.int32)
        conv = np.zeros(B, np.uint8)
        post = np.zeros((B, self.code.n), np.float32 if np.dtype(compute_dtype) == np.float16 else compute_dtype) if want_posterior else None
        ccfg = CDecodeConfig.from_config(cfg)
        _lib.check(self._lib.ldpc_decode_batch_host(
            self._h, ctypes.byref(ccfg), llr.ctypes.data if B else None, _dtype_code(llr.dtype),
            _dtype_code(compute_dtype), kernel, B, hard.ctypes.data if B else None,
            iters.ctypes.data if B else None, conv.ctypes.data if B else None,
            post.ctypes.data if (post is not None and B) else None))
        return DecodeResult(hard, iters, conv.astype(bool), post, _lib.last_kernel())

    # ------------------------------------------------------------------ device pointers
    def decode_device(self, llr_ptr: int, llr_dtype, frames: int, hard_ptr: int, iters_ptr: int,
                      conv_ptr: int, cfg: DecodeConfig, compute_dtype=np.float64,
                      posterior_ptr: int = 0, stream: int = 0, kernel: int = _lib.KERNEL_AUTO,
                      noalloc: bool = False) -> str:
        """Enqueue one decode on device buffers.  ``noalloc`` goes through
        ``ldpc_decode_batch_noalloc`` (no device allocation inside the call)."""
        ccfg = CDecodeConfig.from_config(cfg)
        entry = self._lib.ldpc_decode_batch_noalloc if noalloc else self._lib.ldpc_decode_batch
        _lib.check(entry(
            self._h, ctypes.byref(ccfg), llr_ptr, _dtype_code(llr_dtype), _dtype_code(compute_dtype),
            kernel, int(frames), hard_ptr, iters_ptr, conv_ptr, posterior_ptr or None, stream or None))
        return _lib.last_kernel()

    def count_errors_device(self, hard_ptr: int, ref_ptr: int, ref_stride: int, iters_ptr: int,
                            conv_ptr: int, frames: int, counters_ptr: int, stream: int = 0) -> None:
        _lib.check(self._lib.ldpc_count_errors(hard_ptr, ref_ptr, int(ref_stride), iters_ptr,
                                               conv_ptr, int(frames), self.code.n, counters_ptr,
                                               stream or None))
