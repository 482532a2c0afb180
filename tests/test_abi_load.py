"""The C-ABI library: present, loadable, exports every symbol the header declares, and fails loudly
(rather than falling back to anything) when no GPU is there.  No compute call is made here."""
import ctypes
import os
import re

import numpy as np
import pytest

from ldpc_amd import _lib, config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ldpc_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ldpc_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 10
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_lib.SYMBOLS) == names
    assert lib.ldpc_abi_version() == 3


def test_config_struct_layout_matches_header():
    # 6 x int32 then 5 x double, no padding surprises
    assert ctypes.sizeof(config.CDecodeConfig) == 6 * 4 + 5 * 8
    assert config.CDecodeConfig.from_config(config.OMS).offset == 0.5
    c = config.CDecodeConfig.from_config(config.R3)
    assert (c.rule, c.max_iters, c.scale, c.clip_prod, c.clip_post) == (3, 50, 0.85, 0.99999999, 50.0)


def test_argument_errors_are_reported_without_a_device():
    lib = _lib.load()
    h = ctypes.c_void_p()
    rp = np.array([0, 2, 1], np.int32)      # not monotone
    ci = np.array([0, 1], np.int32)
    rc = lib.ldpc_code_create_csr(2, 4, rp.ctypes.data, ci.ctypes.data, ctypes.byref(h))
    assert rc == _lib.LDPC_ERR_ARG and b"monotone" in lib.ldpc_last_error()
    rp = np.array([0, 2], np.int32)
    ci = np.array([1, 1], np.int32)         # repeated column
    rc = lib.ldpc_code_create_csr(1, 4, rp.ctypes.data, ci.ctypes.data, ctypes.byref(h))
    assert rc == _lib.LDPC_ERR_ARG and b"ascending" in lib.ldpc_last_error()


def test_no_cpu_fallback():
    """Without a GPU a handle cannot even be created: the product path never computes on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _lib.load()
    n = ctypes.c_int32(-1)
    assert lib.ldpc_device_count(ctypes.byref(n)) == _lib.LDPC_ERR_NO_DEVICE and n.value == 0
    from ldpc_amd import codes
    from ldpc_amd.decoder import Decoder
    with pytest.raises(_lib.LdpcError) as ei:
        Decoder(codes.load_packaged("ccsds512"))
    assert ei.value.code == _lib.LDPC_ERR_NO_DEVICE
