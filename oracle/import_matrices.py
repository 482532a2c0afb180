#!/usr/bin/env python3
"""TEST/BUILD INFRASTRUCTURE — runs only where /root/reference is mounted.

Converts the reference's parity-check matrix DATA files into the compact edge-list .npz files
shipped under ldpc_amd/data (the reference tree does not travel to the GPU box).  Checks the
integrity digests and structural facts of each matrix on the way.

    python oracle/import_matrices.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd import codes  # noqa: E402

REF = os.environ.get("LDPC_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "ldpc_amd", "data")

EXPECT = {  # digest, shape, edges, QC (Z, mb, nb, base edges)
    "ccsds512": ("7b1e8934f61bb33e", (256, 512), 2048, (64, 4, 8, 32)),
    "wifi648": ("2bcb6a62e99153e1", (324, 648), 2376, (27, 12, 24, 88)),
    "nr_bg2_z40": ("e55460458486b8c0", (1680, 2080), 7880, (40, 42, 52, 197)),
}


def main() -> None:
    import scipy.io

    os.makedirs(OUT, exist_ok=True)
    mats = {}
    ccsds = scipy.io.loadmat(os.path.join(REF, "Matrices", "LDPC_CCSDS_256.mat"))
    mats["ccsds512"] = (codes.read_mat_matrix(os.path.join(REF, "Matrices", "LDPC_CCSDS_256.mat")),
                        (np.asarray(ccsds["G"]) != 0).astype(np.uint8))
    Hw = codes.read_parity_matrix(os.path.join(REF, "Matrices", "parityMatrix.txt"))
    Hw2 = codes.read_parity_matrix(os.path.join(REF, "Matrices", "H.txt"))
    assert np.array_equal(Hw, Hw2), "parityMatrix.txt and H.txt differ"
    mats["wifi648"] = (Hw, None)
    mats["nr_bg2_z40"] = (codes.read_mat_matrix(os.path.join(REF, "Matrices", "5GMatrix.mat")), None)

    for name, (H, G) in mats.items():
        digest, shape, edges, qc = EXPECT[name]
        assert codes.matrix_digest(H) == digest, (name, codes.matrix_digest(H))
        assert H.shape == shape and int(H.sum()) == edges
        cc = codes.compile_code(H, name)
        assert cc.qc is not None and (cc.qc.Z, cc.qc.mb, cc.qc.nb, cc.qc.num_base_edges) == qc, cc.qc
        assert np.array_equal(cc.qc.expand(), H)
        if G is not None:
            assert not ((H.astype(int) @ G.T.astype(int)) & 1).any()
        path = os.path.join(OUT, name + ".npz")
        codes.save_code_npz(path, H, G)
        H2, G2 = codes.load_code_npz(path)
        assert np.array_equal(H, H2) and (G is None or np.array_equal(G, G2))
        print(f"{name}: {H.shape} E={edges} Z={cc.qc.Z} -> {path} ({os.path.getsize(path)} B)")


if __name__ == "__main__":
    main()
