"""The CPU oracles against the golden vectors (outputs of the reference's own decode methods,
produced by oracle/gen_golden.py).  This is what pins the oracles."""
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden, rule_config
from ldpc_amd import codes
from oracle import bp_oracle

ALL = golden_files()
assert ALL, "tests/golden is empty"


def _ids(files):
    return [os.path.basename(f)[:-4] for f in files]


@pytest.mark.parametrize("path", ALL, ids=_ids(ALL))
def test_numpy_oracle_matches_reference(path):
    g = load_golden(path)
    cfg = rule_config(g["rule"])
    code = codes.load_packaged(g["code"])
    L = bp_oracle.llr_from_samples(g["y"].astype(np.float64), g["snr_attr"], cfg.rule)
    hard, iters, conv, post, trace = bp_oracle.numpy_decode(code, cfg, L, want_trace=True)
    assert np.array_equal(conv, g["converged"])
    assert np.array_equal(iters, g["iters"])
    assert np.array_equal(hard, g["hard"].astype(np.uint8))
    if cfg.rule <= 1:
        # min-sum: same operations in the same order -> identical float64 bits
        assert np.array_equal(post, g["posterior"])
    else:
        np.testing.assert_allclose(post, g["posterior"], rtol=1e-9, atol=1e-12)
    # return value of the reference: BER over all n bits on success, None (nan) on failure
    ber = (hard != g["codeword"]).mean(axis=1)
    ok = g["converged"]
    assert np.array_equal(ber[ok], g["ret"][ok])
    assert np.isnan(g["ret"][~ok]).all()
    # per-iteration posteriors of the traced frames
    for key in [k for k in g if k.startswith("trace_")]:
        f = int(key.split("_")[1])
        ref = g[key]
        for t in range(ref.shape[0]):
            if cfg.rule <= 1:
                assert np.array_equal(trace[t][f], ref[t]), (key, t)
            else:
                np.testing.assert_allclose(trace[t][f], ref[t], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("path", ALL, ids=_ids(ALL))
def test_c_oracle_matches_reference(path, c_oracle):
    g = load_golden(path)
    cfg = rule_config(g["rule"])
    code = codes.load_packaged(g["code"])
    L = bp_oracle.llr_from_samples(g["y"].astype(np.float64), g["snr_attr"], cfg.rule)
    hard, iters, conv, post = c_oracle.c_decode(code, cfg, L)
    assert np.array_equal(conv, g["converged"])
    assert np.array_equal(iters, g["iters"])
    assert np.array_equal(hard, g["hard"].astype(np.uint8))
    if cfg.rule <= 1:
        assert np.array_equal(post, g["posterior"])
    else:
        # libm vs numpy transcendentals: rounding-level differences only
        np.testing.assert_allclose(post, g["posterior"], rtol=1e-4, atol=1e-7)
