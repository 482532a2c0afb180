"""Hard-decision bit flipping (/root/reference/ldpc.py:82-135) against golden vectors produced by
the reference's own method (oracle/gen_golden_bitflip.py): both CPU restatements, then the HIP
kernel through the C ABI.  Integer arithmetic: everything is bit-exact."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from ldpc_amd import codes, config
from oracle import bp_oracle

FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "BF_*.npz")))


def _load(path):
    with np.load(path) as z:
        g = {k: z[k] for k in z.files}
    g["code"] = str(g["code"])
    return g


def _cfg(g, **kw):
    m_attr = int(g["m_attr"])
    return config.BF.with_(num_active_checks=m_attr, **kw)


def _check(g, hard, iters, conv):
    assert np.array_equal(conv, g["converged"]) and np.array_equal(iters, g["iters"])
    # the reference's return value: bit error rate of the final bits over all n positions
    assert np.array_equal((hard != g["codeword"]).mean(axis=1), g["ret"])
    assert np.array_equal(hard[g["converged"]], g["last_tested_bits"][g["converged"]])


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracles_match_reference(path):
    assert len(FILES) == 3
    g = _load(path)
    code = codes.load_packaged(g["code"])
    bp_oracle.build_c_oracle()
    _check(g, *bp_oracle.numpy_bitflip(code, _cfg(g), g["y"]))
    _check(g, *bp_oracle.c_bitflip(code, _cfg(g), g["y"]))
    # the bits at every syndrome test of the first frames: the state after t passes is what the
    # restatement hands back when it is stopped after t passes
    for t in range(1, 10):
        ht, _, _ = bp_oracle.c_bitflip(code, _cfg(g, max_iters=t), g["y"][:4])
        for f in range(4):
            seen = g["trace"][f, t]
            if seen[0] != 255:            # the reference made a test after t passes
                assert np.array_equal(ht[f], seen), (f, t)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_gpu_bitflip_matches_reference(path):
    from ldpc_amd.decoder import Decoder
    g = _load(path)
    code = codes.load_packaged(g["code"])
    dec = Decoder(code)
    for llr in (g["y"], g["y"].astype(np.float32)):
        r = dec.decode(llr, _cfg(g))
        assert r.kernel == "bitflip"
        _check(g, r.hard, r.iters, r.converged)


@pytest.mark.gpu
def test_gpu_bitflip_batch_matches_oracle_and_surface(c_oracle):
    from ldpc_amd import shim
    from ldpc_amd.decoder import Decoder
    rng = np.random.default_rng(4)
    for name in ("ccsds512", "nr_bg2_z40", "nr_bg1_z384"):
        code = codes.load_packaged(name)
        B = 3000 if code.n < 4096 else 40
        y = 1.0 + 0.45 * rng.standard_normal((B, code.n))
        for cfg in (config.BF, config.BF.with_(max_iters=3, num_active_checks=code.m // 3)):
            eh, ei, ec = c_oracle.c_bitflip(code, cfg, y)
            r = Decoder(code).decode(y, cfg)
            assert np.array_equal(r.hard, eh) and np.array_equal(r.iters, ei) and np.array_equal(r.converged, ec)
        assert 0 < ec.sum() or name != "ccsds512"
    # the reference-shaped method: returns the bit error rate whether or not a codeword was reached
    by_name = {os.path.basename(p)[:-4]: p for p in FILES}
    g = _load(by_name["BF_wifi648_mall"])
    e = shim.LdpcPyEncoder(4, 8, 648, readDataMatrix=True)
    for f in range(6):
        e.originalEncoded = g["codeword"][f].astype(int)
        assert e.bitFlipDecode(g["y"][f]) == g["ret"][f]
    # self.m smaller than the matrix (ldpc.py:30,113): only the first self.m rows vote
    g = _load(by_name["BF_wifi648_m100"])
    e = shim.LdpcPyEncoder(4, 8, 648, readDataMatrix=True, compat_loop_bounds=True)
    e.m = int(g["m_attr"])
    for f in range(6):
        e.originalEncoded = g["codeword"][f].astype(int)
        assert e.bitFlipDecode(g["y"][f]) == g["ret"][f]
