import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(rule=None):
    # the belief-propagation fixtures R0..R4; the bit-flipping ones (BF_*) have tests/test_bitflip.py
    files = sorted(glob.glob(os.path.join(GOLDEN_DIR, "R[0-4]*.npz")))
    if rule is not None:
        files = [f for f in files if os.path.basename(f).split("_")[0] == rule]
    return files


def load_golden(path):
    with np.load(path) as z:
        g = {k: z[k] for k in z.files}
    g["rule"] = str(g["rule"])
    g["code"] = str(g["code"])
    g["path"] = path
    return g


def rule_config(rule_name):
    """Reference preset for a fixture's rule label ("R1c" is channel.py's copy of R1)."""
    from ldpc_amd import config
    return config.PRESETS[rule_name[:2]]


@pytest.fixture(scope="session")
def c_oracle():
    from oracle import bp_oracle
    bp_oracle.build_c_oracle()
    return bp_oracle
