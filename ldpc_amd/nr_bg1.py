"""5G NR base graph 1 lifted to Z = 384 — BUILD-AUTHORED, not from the reference.

The reference tree holds no BG1 table (its only 5G matrix is a pre-expanded BG2-shaped Z = 40
matrix with non-standard shifts), and nothing in the build container does either.  This module
therefore authors the code:

* CONNECTIVITY: the 46 x 68 base graph with 316 edges of 3GPP TS 38.212 Table 5.3.2-2, written from
  the published structure (row degrees 19,19,19,19,3,8,9,7,10,9,7,8,7,6,7,7,6,6,6,6,6,6,5,5,6,5,5,4,
  5,5,5,5,5,5,5,5,5,4,5,5,4,5,4,5,5,4; columns 22-25 the dual-diagonal core parity, columns 26-67 the
  degree-1 extension parity).  ``self_check`` verifies those structural facts.
* SHIFTS: the parity part uses the standard's structure (dual diagonal with one shift of 1,
  identity extension).  The 248 shift coefficients of the information part are NOT the standard's
  (they cannot be reproduced here without the table): they are drawn from a fixed-seed generator and
  re-drawn until no length-4 cycle remains in the lifted graph.

Consequences, stated wherever results for this code appear: decoder work per frame (edges, degree
profile, lifting size, memory footprint) is exactly that of NR BG1 Z = 384, so throughput figures
carry over; error-rate curves are those of THIS matrix, not of the standard's, and there is no
reference decoder output to compare with (parity unpinned; checked against this build's own CPU
oracle only).
"""
from __future__ import annotations

import numpy as np

from .codes import QCStructure

MB, NB, Z384 = 46, 68, 384

#: columns of each base row
BG1_ROWS = [
    [0, 1, 2, 3, 5, 6, 9, 10, 11, 12, 13, 15, 16, 18, 19, 20, 21, 22, 23],
    [0, 2, 3, 4, 5, 7, 8, 9, 11, 12, 14, 15, 16, 17, 19, 21, 22, 23, 24],
    [0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 13, 14, 15, 17, 18, 19, 20, 24, 25],
    [0, 1, 3, 4, 6, 7, 8, 10, 11, 12, 13, 14, 16, 17, 18, 20, 21, 22, 25],
    [0, 1, 26],
    [0, 1, 3, 12, 16, 21, 22, 27],
    [0, 6, 10, 11, 13, 17, 18, 20, 28],
    [0, 1, 4, 7, 8, 14, 29],
    [0, 1, 3, 12, 16, 19, 21, 22, 24, 30],
    [0, 1, 10, 11, 13, 17, 18, 20, 31],
    [1, 2, 4, 7, 8, 14, 32],
    [0, 1, 12, 16, 21, 22, 23, 33],
    [0, 1, 10, 11, 13, 18, 34],
    [0, 3, 7, 20, 23, 35],
    [0, 12, 15, 16, 17, 21, 36],
    [0, 1, 10, 13, 18, 25, 37],
    [1, 3, 11, 20, 22, 38],
    [0, 14, 16, 17, 21, 39],
    [1, 12, 13, 18, 19, 40],
    [0, 1, 7, 8, 10, 41],
    [0, 3, 9, 11, 22, 42],
    [1, 5, 16, 20, 21, 43],
    [0, 12, 13, 17, 44],
    [1, 2, 10, 18, 45],
    [0, 3, 4, 11, 22, 46],
    [1, 6, 7, 14, 47],
    [0, 2, 4, 15, 48],
    [1, 6, 8, 49],
    [0, 4, 19, 21, 50],
    [1, 14, 18, 25, 51],
    [0, 10, 13, 24, 52],
    [1, 7, 22, 25, 53],
    [0, 12, 14, 24, 54],
    [1, 2, 11, 21, 55],
    [0, 7, 15, 17, 56],
    [1, 6, 12, 22, 57],
    [0, 14, 15, 18, 58],
    [1, 13, 23, 59],
    [0, 9, 10, 12, 60],
    [1, 3, 7, 19, 61],
    [0, 8, 17, 62],
    [1, 3, 9, 18, 63],
    [0, 4, 24, 64],
    [1, 16, 18, 25, 65],
    [0, 7, 9, 22, 66],
    [1, 6, 10, 67],
]

ROW_DEGREES = [19, 19, 19, 19, 3, 8, 9, 7, 10, 9, 7, 8, 7, 6, 7, 7, 6, 6, 6, 6, 6, 6, 5, 5, 6, 5, 5, 4,
               5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 5, 5, 4, 5, 4, 5, 5, 4]

#: shifts of the core parity columns 22..25 (dual diagonal): (row, col) -> shift
CORE_PARITY_SHIFTS = {(0, 22): 1, (0, 23): 0, (1, 22): 0, (1, 23): 0, (1, 24): 0, (2, 24): 0, (2, 25): 0,
                      (3, 22): 1, (3, 25): 0}


def self_check() -> None:
    assert len(BG1_ROWS) == MB and [len(r) for r in BG1_ROWS] == ROW_DEGREES
    assert sum(ROW_DEGREES) == 316
    for r, cols in enumerate(BG1_ROWS):
        assert cols == sorted(set(cols)) and 0 <= cols[0] and cols[-1] < NB
        if r >= 4:
            assert cols[-1] == 22 + r            # identity extension: one new parity column per row
    col_deg = np.zeros(NB, int)
    for cols in BG1_ROWS:
        for c in cols:
            col_deg[c] += 1
    assert (col_deg[26:] == 1).all() and col_deg.max() == 30 and col_deg[0] == 30 and col_deg[1] == 28
    for key in CORE_PARITY_SHIFTS:
        assert key[1] in BG1_ROWS[key[0]]


def _has_4cycle(rows, shifts, Z) -> bool:
    """Length-4 cycle condition between every pair of base rows sharing two columns."""
    for a in range(len(rows)):
        ca = rows[a]
        for b in range(a + 1, len(rows)):
            common = [c for c in rows[b] if c in ca]
            for i in range(len(common)):
                for j in range(i + 1, len(common)):
                    c1, c2 = common[i], common[j]
                    if (shifts[a, c1] - shifts[b, c1] + shifts[b, c2] - shifts[a, c2]) % Z == 0:
                        return True
    return False


def build(Z: int = Z384, seed: int = 38212) -> QCStructure:
    self_check()
    rng = np.random.default_rng(seed)
    shifts = -np.ones((MB, NB), int)
    rowsets = [dict.fromkeys(r) for r in BG1_ROWS]
    for r, cols in enumerate(BG1_ROWS):
        for c in cols:
            if (r, c) in CORE_PARITY_SHIFTS:
                shifts[r, c] = CORE_PARITY_SHIFTS[(r, c)] % Z
            elif c >= 26:
                shifts[r, c] = 0
            else:
                shifts[r, c] = int(rng.integers(0, Z))
    # remove length-4 cycles by re-drawing one information-part shift of an offending pair
    for _ in range(10000):
        fixed = True
        for a in range(MB):
            for b in range(a + 1, MB):
                common = [c for c in BG1_ROWS[b] if c in rowsets[a]]
                for i in range(len(common)):
                    for j in range(i + 1, len(common)):
                        c1, c2 = common[i], common[j]
                        if (shifts[a, c1] - shifts[b, c1] + shifts[b, c2] - shifts[a, c2]) % Z == 0:
                            free = [(r, c) for r in (a, b) for c in (c1, c2)
                                    if c < 22 and (r, c) not in CORE_PARITY_SHIFTS]
                            if not free:
                                continue
                            r, c = free[int(rng.integers(0, len(free)))]
                            shifts[r, c] = int(rng.integers(0, Z))
                            fixed = False
        if fixed:
            break
    er, ec, es = [], [], []
    for r, cols in enumerate(BG1_ROWS):
        for c in cols:
            er.append(r)
            ec.append(c)
            es.append(int(shifts[r, c]))
    return QCStructure(Z, MB, NB, np.asarray(er, np.int32), np.asarray(ec, np.int32), np.asarray(es, np.int32))


def encode(qc: QCStructure, info: np.ndarray) -> np.ndarray:
    """Systematic encoder for the tables ``build`` makes: info[B, 22 Z] -> codewords[B, 68 Z].

    Uses the parity structure directly instead of a dense generator (the lifted matrix is 17 664 x
    26 112).  With x_c the Z-vector of block column c and ``s`` the shift of block (r, c), the block's
    contribution to check row r is ``roll(x_c, -s)``.  The four core rows
        r0: l0 + roll(p0, -1) + p1 = 0      r1: l1 + p0 + p1 + p2 = 0
        r2: l2 + p2 + p3 = 0                r3: l3 + roll(p0, -1) + p3 = 0
    (l_r = information part of row r) sum to p0 = l0 + l1 + l2 + l3; p1, p3, p2 follow by
    substitution, and every extension row gives its own parity block as the sum of the rest.
    """
    Z = qc.Z
    info = np.asarray(info, dtype=np.uint8)
    B = info.shape[0]
    assert info.shape == (B, 22 * Z) and qc.mb == MB and qc.nb == NB
    x = np.zeros((B, NB, Z), np.uint8)
    x[:, :22] = info.reshape(B, 22, Z)
    shifts = {(int(r), int(c)): int(s) for r, c, s in zip(qc.edge_row, qc.edge_col, qc.edge_shift)}
    for key, s in CORE_PARITY_SHIFTS.items():
        assert shifts[key] == s % Z, "not a table made by nr_bg1.build"

    def row_sum(r, cols):
        acc = np.zeros((B, Z), np.uint8)
        for c in cols:
            acc ^= np.roll(x[:, c], -shifts[(r, c)], axis=1)
        return acc

    lam = [row_sum(r, [c for c in BG1_ROWS[r] if c < 22]) for r in range(4)]
    p0 = lam[0] ^ lam[1] ^ lam[2] ^ lam[3]
    p0s = np.roll(p0, -(1 % Z), axis=1)
    x[:, 22] = p0
    x[:, 23] = lam[0] ^ p0s
    x[:, 25] = lam[3] ^ p0s
    x[:, 24] = lam[2] ^ x[:, 25]
    for r in range(4, MB):
        x[:, 22 + r] = row_sum(r, BG1_ROWS[r][:-1])
    return x.reshape(B, NB * Z)


class Encoder:
    """Callable message -> codeword map for a table made by ``build`` (what ``sweep.make_batch``
    takes in place of a dense generator matrix)."""

    def __init__(self, qc: QCStructure):
        self.qc = qc
        self.k = 22 * qc.Z

    def __call__(self, info: np.ndarray) -> np.ndarray:
        return encode(self.qc, info)
