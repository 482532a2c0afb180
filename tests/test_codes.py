"""Host logic of the code compiler: loaders, edge lists, circulant detection, generator."""
import os

import numpy as np
import pytest

from ldpc_amd import codes

NAMES = ["ccsds512", "wifi648", "nr_bg2_z40", "nr_bg2_z40_r1400"]


def test_three_file_formats_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    H = (rng.random((12, 30)) < 0.2).astype(np.uint8)
    bits = tmp_path / "bits.txt"
    bits.write_text("\n".join("".join(map(str, row)) for row in H) + "\n")       # ldpc.py:360-366
    spaced = tmp_path / "spaced.txt"
    spaced.write_text("\n".join(" ".join(map(str, row)) for row in H) + "\n")    # Matrices/H.txt
    import scipy.io
    mat = tmp_path / "m.mat"
    scipy.io.savemat(mat, {"H": H})                                                # ccsds.py:6-8
    for p in (bits, spaced, mat):
        assert np.array_equal(codes.read_parity_matrix(str(p)), H), p


def test_bad_files_are_rejected(tmp_path):
    p = tmp_path / "ragged.txt"
    p.write_text("0101\n011\n")
    with pytest.raises(ValueError):
        codes.read_bitstring_matrix(str(p))
    p.write_text("01x1\n")
    with pytest.raises(ValueError):
        codes.read_bitstring_matrix(str(p))
    p.write_text("")
    with pytest.raises(ValueError):
        codes.read_bitstring_matrix(str(p))


@pytest.mark.parametrize("name", NAMES)
def test_edge_lists_are_consistent(name):
    c = codes.load_packaged(name)
    H = c.dense()
    assert c.num_edges == int(H.sum())
    for j in (0, c.m // 2, c.m - 1):
        assert np.array_equal(c.col_idx[c.row_ptr[j]:c.row_ptr[j + 1]], np.flatnonzero(H[j]))
    for i in (0, c.n // 3, c.n - 1):
        ks = slice(c.col_ptr[i], c.col_ptr[i + 1])
        assert np.array_equal(c.csc_row[ks], np.flatnonzero(H[:, i]))
        # the cross index points at the same edge in check-major order
        for k in range(ks.start, ks.stop):
            e = c.csc_edge[k]
            assert c.col_idx[e] == i and c.row_ptr[c.csc_row[k]] <= e < c.row_ptr[c.csc_row[k] + 1]


def test_digests_and_shapes_of_the_packaged_matrices():
    expect = {"ccsds512": ("7b1e8934f61bb33e", 256, 512, 2048), "wifi648": ("2bcb6a62e99153e1", 324, 648, 2376),
              "nr_bg2_z40": ("e55460458486b8c0", 1680, 2080, 7880)}
    for name, (dig, m, n, E) in expect.items():
        c = codes.load_packaged(name)
        assert (c.digest, c.m, c.n, c.num_edges) == (dig, m, n, E)
    t = codes.load_packaged("nr_bg2_z40_r1400")
    assert (t.m, t.n, t.num_edges) == (1400, 2080, 6800)
    assert int((np.diff(t.col_ptr) == 0).sum()) == 280      # degree-0 columns left by H[:1400]


def test_circulant_detection_known_answers():
    q = codes.load_packaged("ccsds512").qc
    assert (q.Z, q.mb, q.nb, q.num_base_edges) == (64, 4, 8, 32)
    rows = {r: [] for r in range(4)}
    for r, c, s in zip(q.edge_row, q.edge_col, q.edge_shift):
        rows[int(r)].append((int(c), int(s)))
    assert sorted(rows[0]) == [(0, 0), (0, 63), (1, 30), (2, 50), (3, 25), (5, 43), (6, 62), (7, 0)]
    assert sorted(rows[3]) == [(0, 35), (1, 56), (2, 62), (3, 0), (3, 11), (4, 58), (5, 3), (6, 0)]
    w = codes.load_packaged("wifi648").qc
    assert (w.Z, w.mb, w.nb, w.num_base_edges) == (27, 12, 24, 88)
    first = sorted((int(c), int(s)) for r, c, s in zip(w.edge_row, w.edge_col, w.edge_shift) if r == 0)
    assert first == [(0, 0), (4, 0), (5, 0), (8, 0), (11, 0), (12, 1), (13, 0)]
    g = codes.load_packaged("nr_bg2_z40").qc
    assert (g.Z, g.mb, g.nb, g.num_base_edges) == (40, 42, 52, 197)
    assert set(np.unique(g.edge_shift)) <= {0, 1, 2, 3, 4}
    for name in NAMES:
        c = codes.load_packaged(name)
        assert np.array_equal(c.qc.expand(), c.dense())


def test_no_circulant_structure_in_a_random_matrix():
    rng = np.random.default_rng(3)
    H = (rng.random((24, 48)) < 0.15).astype(np.uint8)
    H[0, 0] = 1
    assert codes.detect_qc(H) is None or codes.detect_qc(H).Z < 24


@pytest.mark.parametrize("name", NAMES)
def test_generator_spans_the_code(name):
    c = codes.load_packaged(name)
    G = codes.load_packaged_generator(name)
    assert G.shape == (c.n - np.linalg.matrix_rank(c.dense().astype(float)) if c.n <= 648 else G.shape[0], c.n)
    rng = np.random.default_rng(1)
    cw = codes.encode(G, rng.integers(0, 2, size=(16, G.shape[0])))
    assert c.is_codeword(cw).all()
    assert len({bytes(r) for r in cw}) == 16


def test_syndrome_matches_dense_product():
    c = codes.load_packaged("wifi648")
    rng = np.random.default_rng(2)
    x = rng.integers(0, 2, size=(5, c.n))
    assert np.array_equal(c.syndrome(x), (x @ c.dense().T.astype(int)) % 2)


def test_user_registered_circulant_table(tmp_path, monkeypatch):
    """ldpc_amd.add_code: a caller's table becomes loadable by name and appears in the generated header."""
    from ldpc_amd import add_code, gen_codes, nr_bg1
    monkeypatch.setattr(codes, "_DATA_DIR", str(tmp_path))
    monkeypatch.setattr(codes, "USER_REGISTRY", str(tmp_path / "user_codes.json"))
    qc = nr_bg1.build(Z=96, seed=6)
    table = tmp_path / "mine.npz"
    codes.save_qc_npz(str(table), qc)
    rec = add_code.register("my_bg1_z96", str(table))
    assert rec["slots"] == 1 and rec["Z"] == 96 and "my_bg1_z96" in codes.user_codes()
    code = codes.load_packaged("my_bg1_z96")
    assert (code.m, code.n, code.qc.Z) == (46 * 96, 68 * 96, 96)
    assert gen_codes.specialised()["my_bg1_z96"] == (1,)
    text = gen_codes.describe("my_bg1_z96")
    assert "struct Code_my_bg1_z96" in text and "Z = 96, MB = 46, NB = 68, EB = 316" in text
    with pytest.raises(ValueError):
        add_code.register("ccsds512", str(table))
    add_code.remove("my_bg1_z96")
    codes._cache.pop("my_bg1_z96", None)
    assert "my_bg1_z96" not in codes.user_codes()
