"""Parity-check matrix ingestion and one-off compilation into decoder layouts.

Covers the H-ingestion step of the reference scripts and everything the decode
kernels need that can be derived once per code:

* loaders for the three on-disk formats the reference reads
  (``readMatrix`` /root/reference/ldpc.py:360-366 — one '0'/'1' character per bit,
  one row per line; whitespace separated integers as in ``Matrices/H.txt``;
  MATLAB v5 files with an ``"H"`` key, /root/reference/ccsds.py:6-8 and
  /root/reference/channel.py:34);
* CSR (check-major, ascending column) and CSC (variable-major, ascending check)
  edge lists with the cross index between the two orderings;
* a quasi-cyclic detector that recovers the lifting size Z and the circulant
  shift table (blocks that are sums of several circulants are supported);
* a GF(2) generator matrix (replaces ``pyldpc.coding_matrix``,
  /root/reference/ldpc.py:369) so callers can produce random codewords.

Nothing here touches the GPU; numpy only.
"""
from __future__ import annotations

import hashlib
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


# --------------------------------------------------------------------------- loaders
def read_bitstring_matrix(path: str) -> np.ndarray:
    """One row per line, one character per bit, no separators (ldpc.py:360-366)."""
    rows = []
    with open(path, "r") as f:
        for line in f:
            s = line.strip()
            if not s:
                continue
            bad = set(s) - {"0", "1"}
            if bad:
                raise ValueError(f"{path}: unexpected characters {sorted(bad)} in bit-string row")
            rows.append(np.frombuffer(s.encode("ascii"), dtype=np.uint8) - ord("0"))
    if not rows:
        raise ValueError(f"{path}: empty matrix file")
    width = len(rows[0])
    if any(len(r) != width for r in rows):
        raise ValueError(f"{path}: ragged rows")
    return np.stack(rows).astype(np.uint8)


def read_spaced_matrix(path: str) -> np.ndarray:
    """Whitespace separated 0/1 integers, one row per line (Matrices/H.txt)."""
    H = np.loadtxt(path, dtype=np.int64, ndmin=2)
    if not np.isin(H, (0, 1)).all():
        raise ValueError(f"{path}: entries other than 0/1")
    return H.astype(np.uint8)


def read_mat_matrix(path: str, key: str = "H") -> np.ndarray:
    """MATLAB v5 file, dense matrix under ``key`` (ccsds.py:6-8, channel.py:34)."""
    import scipy.io

    mat = scipy.io.loadmat(path)
    if key not in mat:
        raise KeyError(f"{path}: no variable {key!r}")
    return (np.asarray(mat[key]) != 0).astype(np.uint8)


def read_parity_matrix(path: str) -> np.ndarray:
    """Dispatch on extension / content to one of the three loaders."""
    if path.endswith(".mat"):
        return read_mat_matrix(path)
    with open(path, "r") as f:
        first = f.readline()
    if " " in first.strip() or "\t" in first.strip():
        return read_spaced_matrix(path)
    return read_bitstring_matrix(path)


def matrix_digest(H: np.ndarray) -> str:
    """First 16 hex digits of sha256 over the dense uint8 matrix (integrity check)."""
    return hashlib.sha256(np.ascontiguousarray(H, dtype=np.uint8).tobytes()).hexdigest()[:16]


# --------------------------------------------------------------------------- QC structure
@dataclass
class QCStructure:
    """Quasi-cyclic description: H is an mb x nb array of Z x Z blocks, each block a sum of
    circulants.  Base edge e = (block row, block column, shift s) means check (br, z) is joined
    to variable (bc, (z + s) mod Z) for every z in [0, Z)."""

    Z: int
    mb: int
    nb: int
    edge_row: np.ndarray   # int32 [Eb]  block row, ascending; within a row ascending column
    edge_col: np.ndarray   # int32 [Eb]
    edge_shift: np.ndarray  # int32 [Eb]

    @property
    def num_base_edges(self) -> int:
        return int(self.edge_row.shape[0])

    def expand(self) -> np.ndarray:
        H = np.zeros((self.mb * self.Z, self.nb * self.Z), dtype=np.uint8)
        z = np.arange(self.Z)
        for r, c, s in zip(self.edge_row, self.edge_col, self.edge_shift):
            H[r * self.Z + z, c * self.Z + (z + s) % self.Z] ^= 1
        return H


def _block_shifts(block: np.ndarray) -> Optional[List[int]]:
    """Return the shifts if ``block`` (Z x Z) is a sum of distinct circulants, else None."""
    Z = block.shape[0]
    shifts = np.flatnonzero(block[0]).tolist()
    z = np.arange(Z)
    rebuilt = np.zeros_like(block)
    for s in shifts:
        rebuilt[z, (z + s) % Z] = 1
    return shifts if np.array_equal(rebuilt, block) else None


def detect_qc(H: np.ndarray, min_z: int = 2) -> Optional[QCStructure]:
    """Largest Z for which every Z x Z block of H is a sum of circulants (None if no Z>=min_z)."""
    m, n = H.shape
    g = int(np.gcd(m, n))
    for Z in sorted((d for d in range(min_z, g + 1) if g % d == 0), reverse=True):
        mb, nb = m // Z, n // Z
        rows, cols, shifts = [], [], []
        ok = True
        for br in range(mb):
            band = H[br * Z:(br + 1) * Z]
            for bc in range(nb):
                block = band[:, bc * Z:(bc + 1) * Z]
                if not block.any():
                    continue
                sh = _block_shifts(block)
                if sh is None:
                    ok = False
                    break
                for s in sh:
                    rows.append(br)
                    cols.append(bc)
                    shifts.append(s)
            if not ok:
                break
        if ok:
            return QCStructure(Z, mb, nb, np.asarray(rows, np.int32), np.asarray(cols, np.int32),
                               np.asarray(shifts, np.int32))
    return None


# --------------------------------------------------------------------------- compiled code
@dataclass
class CompiledCode:
    """Edge-list view of H used by every decoder in this package.

    CSR order (``row_ptr``/``col_idx``) lists each check's variables in ascending column order;
    this is the order the reference visits ``numpy.where(H[j] == 1)[0]`` (ldpc.py:176).
    CSC order (``col_ptr``/``csc_row``) lists each variable's checks in ascending row order,
    the order of ``np.where(H[:, i] == 1)[0]`` (bpsk.py:236); ``csc_edge`` maps each CSC entry to
    the position of the same edge in CSR order, so a message array kept in CSR order can be
    gathered per variable.
    """

    name: str
    m: int
    n: int
    row_ptr: np.ndarray   # int32 [m+1]
    col_idx: np.ndarray   # int32 [E]
    col_ptr: np.ndarray   # int32 [n+1]
    csc_row: np.ndarray   # int32 [E]
    csc_edge: np.ndarray  # int32 [E]
    qc: Optional[QCStructure] = None
    digest: str = ""
    _dense: Optional[np.ndarray] = field(default=None, repr=False)

    @property
    def num_edges(self) -> int:
        return int(self.col_idx.shape[0])

    @property
    def max_check_degree(self) -> int:
        return int(np.diff(self.row_ptr).max()) if self.m else 0

    @property
    def max_var_degree(self) -> int:
        return int(np.diff(self.col_ptr).max()) if self.n else 0

    def dense(self) -> np.ndarray:
        if self._dense is None:
            H = np.zeros((self.m, self.n), dtype=np.uint8)
            rows = np.repeat(np.arange(self.m), np.diff(self.row_ptr))
            H[rows, self.col_idx] = 1
            self._dense = H
        return self._dense

    def syndrome(self, bits: np.ndarray) -> np.ndarray:
        """(H x) mod 2 for bits[..., n] -> [..., m] (isValidCodeword, ldpc.py:76-80)."""
        bits = np.asarray(bits).astype(np.uint8)
        gathered = bits[..., self.col_idx]
        cs = np.concatenate([np.zeros(gathered.shape[:-1] + (1,), np.int64),
                             np.cumsum(gathered, axis=-1, dtype=np.int64)], axis=-1)
        return ((cs[..., self.row_ptr[1:]] - cs[..., self.row_ptr[:-1]]) & 1).astype(np.uint8)

    def is_codeword(self, bits: np.ndarray) -> np.ndarray:
        return ~self.syndrome(bits).any(axis=-1)

    def truncated(self, rows: int, name: Optional[str] = None) -> "CompiledCode":
        """Keep the first ``rows`` checks (the ``H[:1400, :]`` of channel.py:35)."""
        return compile_code(self.dense()[:rows], name or f"{self.name}[:{rows}]")


def compile_code(H: np.ndarray, name: str = "code", find_qc: bool = True) -> CompiledCode:
    H = (np.asarray(H) != 0).astype(np.uint8)
    if H.ndim != 2 or H.size == 0:
        raise ValueError("H must be a non-empty 2-D matrix")
    m, n = H.shape
    rows, cols = np.nonzero(H)               # row-major: ascending row, then ascending column
    E = rows.shape[0]
    row_ptr = np.zeros(m + 1, np.int32)
    np.cumsum(np.bincount(rows, minlength=m), out=row_ptr[1:])
    col_idx = cols.astype(np.int32)
    order = np.lexsort((rows, cols))         # sort by column, then row
    col_ptr = np.zeros(n + 1, np.int32)
    np.cumsum(np.bincount(cols, minlength=n), out=col_ptr[1:])
    csc_row = rows[order].astype(np.int32)
    csc_edge = order.astype(np.int32)
    qc = detect_qc(H) if find_qc else None
    code = CompiledCode(name, m, n, row_ptr, col_idx, col_ptr, csc_row, csc_edge, qc,
                        matrix_digest(H), H)
    assert code.num_edges == E
    return code


def compile_qc(qc: QCStructure, name: str = "qc") -> CompiledCode:
    """Edge lists straight from a circulant description (no dense matrix is formed: the lifted 5G
    graphs are hundreds of megabytes dense)."""
    Z = qc.Z
    z = np.arange(Z)
    rows = (qc.edge_row[:, None].astype(np.int64) * Z + z[None, :]).ravel()
    cols = (qc.edge_col[:, None].astype(np.int64) * Z + (z[None, :] + qc.edge_shift[:, None]) % Z).ravel()
    m, n = qc.mb * Z, qc.nb * Z
    o = np.lexsort((cols, rows))
    rows, cols = rows[o], cols[o]
    if len(rows) > 1 and ((np.diff(rows) == 0) & (np.diff(cols) == 0)).any():
        raise ValueError("repeated edge in the circulant description")
    row_ptr = np.zeros(m + 1, np.int32)
    np.cumsum(np.bincount(rows, minlength=m), out=row_ptr[1:])
    order = np.lexsort((rows, cols))
    col_ptr = np.zeros(n + 1, np.int32)
    np.cumsum(np.bincount(cols, minlength=n), out=col_ptr[1:])
    digest = hashlib.sha256(np.stack([rows, cols]).astype(np.int32).tobytes()).hexdigest()[:16]
    return CompiledCode(name, m, n, row_ptr, cols.astype(np.int32), col_ptr, rows[order].astype(np.int32),
                        order.astype(np.int32), qc, digest, None)


# --------------------------------------------------------------------------- GF(2) generator
def gf2_generator(H: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Generator matrix G (k x n, uint8) with H @ G.T == 0 (mod 2), and the k information
    positions (columns of G that carry the message unchanged).

    Plain Gauss-Jordan elimination over GF(2) on bit-packed rows.  Stands in for
    ``pyldpc.coding_matrix`` (ldpc.py:369); the reference applies its G as ``G @ msg % 2``
    with G stored n x k (ldpc.py:46) — use ``encode`` below for that product.
    """
    A = (np.asarray(H) != 0).astype(np.uint8).copy()
    m, n = A.shape
    pivots: List[int] = []
    r = 0
    for c in range(n):
        if r == m:
            break
        nz = np.flatnonzero(A[r:, c])
        if nz.size == 0:
            continue
        p = r + nz[0]
        if p != r:
            A[[r, p]] = A[[p, r]]
        others = np.flatnonzero(A[:, c])
        others = others[others != r]
        A[others] ^= A[r]
        pivots.append(c)
        r += 1
    rank = r
    pivots_arr = np.asarray(pivots, np.int64)
    free = np.setdiff1d(np.arange(n), pivots_arr)
    k = n - rank
    G = np.zeros((k, n), np.uint8)
    G[np.arange(k), free] = 1
    # row i of the reduced system: x[pivot_i] = sum_j A[i, free_j] x[free_j]
    G[:, pivots_arr] = A[:rank][:, free].T
    assert not ((H.astype(np.int64) @ G.T.astype(np.int64)) & 1).any()
    return G, free


def encode(G: np.ndarray, msgs: np.ndarray) -> np.ndarray:
    """Codewords for messages [..., k] with a k x n generator: (msg @ G) mod 2."""
    return ((np.asarray(msgs, np.int64) @ np.asarray(G, np.int64)) & 1).astype(np.uint8)


# --------------------------------------------------------------------------- packaged codes
#: name -> (file in ldpc_amd/data, truncate-to-rows or None)
PACKAGED = {
    "ccsds512": ("ccsds512.npz", None),       # Matrices/LDPC_CCSDS_256.mat  (n=512, k=256)
    "wifi648": ("wifi648.npz", None),         # Matrices/parityMatrix.txt == H.txt (802.11n 648,324)
    "nr_bg2_z40": ("nr_bg2_z40.npz", None),   # Matrices/5GMatrix.mat (1680 x 2080)
    "nr_bg2_z40_r1400": ("nr_bg2_z40.npz", 1400),  # H[:1400, :] as used by channel.py:35
}


#: circulant tables registered by the user (``python -m ldpc_amd.add_code``): name -> file under
#: ldpc_amd/data holding Z, mb, nb, edge_row, edge_col, edge_shift.  Listed in user_codes.json;
#: each gets a compiled-in kernel at the next build, like the packaged codes.
USER_REGISTRY = os.path.join(_DATA_DIR, "user_codes.json")


def user_codes() -> dict:
    """name -> {"file": ..., "slots": frames per workgroup} of the registered circulant tables."""
    import json
    if not os.path.exists(USER_REGISTRY):
        return {}
    with open(USER_REGISTRY) as f:
        return json.load(f)


def save_qc_npz(path: str, qc: QCStructure) -> None:
    order = np.lexsort((qc.edge_shift, qc.edge_col, qc.edge_row))
    np.savez_compressed(path, Z=np.int32(qc.Z), mb=np.int32(qc.mb), nb=np.int32(qc.nb),
                        edge_row=np.asarray(qc.edge_row, np.int32)[order],
                        edge_col=np.asarray(qc.edge_col, np.int32)[order],
                        edge_shift=np.asarray(qc.edge_shift, np.int32)[order])


def load_qc_npz(path: str) -> QCStructure:
    with np.load(path, allow_pickle=False) as d:
        return QCStructure(int(d["Z"]), int(d["mb"]), int(d["nb"]), d["edge_row"].astype(np.int32),
                           d["edge_col"].astype(np.int32), d["edge_shift"].astype(np.int32))


def save_code_npz(path: str, H: np.ndarray, G: Optional[np.ndarray] = None) -> None:
    H = (np.asarray(H) != 0).astype(np.uint8)
    rows, cols = np.nonzero(H)
    payload = dict(m=np.int32(H.shape[0]), n=np.int32(H.shape[1]),
                   rows=rows.astype(np.int32), cols=cols.astype(np.int32),
                   digest=np.frombuffer(matrix_digest(H).encode(), np.uint8))
    if G is not None:
        payload["G_packed"] = np.packbits((np.asarray(G) != 0).astype(np.uint8), axis=1)
        payload["G_shape"] = np.asarray(G.shape, np.int32)
    np.savez_compressed(path, **payload)


def load_code_npz(path: str) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    with np.load(path) as z:
        H = np.zeros((int(z["m"]), int(z["n"])), np.uint8)
        H[z["rows"], z["cols"]] = 1
        if matrix_digest(H) != bytes(z["digest"]).decode():
            raise ValueError(f"{path}: digest mismatch")
        G = None
        if "G_packed" in z:
            k, n = (int(v) for v in z["G_shape"])
            G = np.unpackbits(z["G_packed"], axis=1)[:, :n].astype(np.uint8)
            assert G.shape == (k, n)
    return H, G


_cache = {}


def load_packaged(name: str) -> CompiledCode:
    """One of the reference's own matrices, shipped as an edge list under ldpc_amd/data."""
    if name == "nr_bg1_z384" and name not in _cache:
        # build-authored (see nr_bg1.py): BG1 connectivity, synthetic information-part shifts
        from . import nr_bg1
        _cache[name] = compile_qc(nr_bg1.build(), name)
    if name in _cache:
        return _cache[name]
    if name not in PACKAGED:
        reg = user_codes()
        if name in reg:
            _cache[name] = compile_qc(load_qc_npz(os.path.join(_DATA_DIR, reg[name]["file"])), name)
            return _cache[name]
        raise KeyError(f"unknown packaged code {name!r}; have {sorted(PACKAGED) + sorted(reg)}")
    if name not in _cache:
        fname, trunc = PACKAGED[name]
        H, _ = load_code_npz(os.path.join(_DATA_DIR, fname))
        if trunc is not None:
            H = H[:trunc]
        _cache[name] = compile_code(H, name)
    return _cache[name]


def load_packaged_generator(name: str):
    """k x n generator for a packaged code (for nr_bg1_z384 a callable encoder, see nr_bg1.Encoder): the one stored with the matrix if the reference's
    file carried one (the CCSDS .mat does), else derived by ``gf2_generator``."""
    key = ("G", name)
    if name == "nr_bg1_z384":
        # too large for a dense generator: a structured encoder (callable, attribute k) instead
        if key not in _cache:
            from . import nr_bg1
            _cache[key] = nr_bg1.Encoder(nr_bg1.build())
        return _cache[key]
    if key not in _cache:
        fname, trunc = PACKAGED[name]
        H, G = load_code_npz(os.path.join(_DATA_DIR, fname))
        if trunc is not None:
            H, G = H[:trunc], None
        if G is None:
            G, _ = gf2_generator(H)
        _cache[key] = G
    return _cache[key]
