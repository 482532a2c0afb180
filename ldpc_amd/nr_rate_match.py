"""5G NR rate matching front end: the numbers around the decoder that 3GPP TS 38.212 prescribes and
that the reference replaces by two constants - the matrix cut to its first 1400 rows
(/root/reference/channel.py:35) and the first 80 LLRs overwritten (bpsk.py:163, channel.py:412).

* ``select_base_graph`` (38.212 section 7.2.2), ``LIFTING_SIZES`` / ``select_lifting`` (Table 5.3.2-1,
  section 5.2.2): base graph, lifting size Z_c, filler count for a block of ``B`` bits;
* ``lift``: shift coefficients of a base table for a lifting size (P = V mod Z_c, section 5.3.2);
* ``RateMatch``: circular buffer, starting position per redundancy version (Table 5.4.2.1-2), bit
  selection and bit interleaving (sections 5.4.2.1, 5.4.2.2) as the configuration of the device
  kernels ``ldpc_rate_match`` / ``ldpc_rate_recover``, with a numpy restatement of both;
* ``active_rows`` / ``active_submatrix``: the part of the parity-check matrix a receiver needs when
  only the first E bits of the buffer were sent - extension parity bits that never went over the
  air, and the checks that involve them, drop out, which is what ``H[:1400, :]`` stands for.

The standard's shift tables (Tables 5.3.2-2 / 5.3.2-3) are in neither the reference nor the build
container; everything here is independent of the shift values and works on any table with the NR
layout (information columns, four core parity columns, one extension parity column per further
row).  PARITY UNPINNED by the reference except for the two constants above, which
``tests/test_rate_match.py`` reproduces from these functions.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from .codes import QCStructure

#: Table 5.3.2-1: lifting sizes per set index i_LS
LIFTING_SETS = {0: [2, 4, 8, 16, 32, 64, 128, 256], 1: [3, 6, 12, 24, 48, 96, 192, 384], 2: [5, 10, 20, 40, 80, 160, 320],
                3: [7, 14, 28, 56, 112, 224], 4: [9, 18, 36, 72, 144, 288], 5: [11, 22, 44, 88, 176, 352],
                6: [13, 26, 52, 104, 208], 7: [15, 30, 60, 120, 240]}
LIFTING_SIZES = sorted(z for zs in LIFTING_SETS.values() for z in zs)

#: (information block columns, block columns, block rows, maximum code block size) of the base graphs
BASE_GRAPHS = {1: (22, 68, 46, 8448), 2: (10, 52, 42, 3840)}
#: Table 5.4.2.1-2: k0 = floor(num * Ncb / (den * Zc)) * Zc per redundancy version
K0_NUMERATORS = {1: (0, 17, 33, 56), 2: (0, 13, 25, 43)}


def select_base_graph(payload_bits: int, rate: float) -> int:
    """38.212 section 7.2.2: base graph 2 for small or low-rate blocks, else base graph 1."""
    if payload_bits <= 292 or (payload_bits <= 3824 and rate <= 0.67) or rate <= 0.25:
        return 2
    return 1


def lifting_set_index(Z: int) -> int:
    for i, zs in LIFTING_SETS.items():
        if Z in zs:
            return i
    raise ValueError(f"{Z} is not a lifting size of Table 5.3.2-1")


def select_lifting(block_bits: int, base_graph: int) -> Tuple[int, int, int]:
    """(Z_c, K, fillers) for one code block of ``block_bits`` bits (38.212 section 5.2.2): the smallest
    lifting size with K_b * Z_c >= block_bits; K = 22 Z_c or 10 Z_c; K - block_bits filler bits."""
    kb_cols, _, _, kcb = BASE_GRAPHS[base_graph]
    if not 0 < block_bits <= kcb:
        raise ValueError(f"a code block of base graph {base_graph} holds 1..{kcb} bits")
    if base_graph == 1:
        kb = 22
    else:
        kb = 10 if block_bits > 640 else 9 if block_bits > 560 else 8 if block_bits > 192 else 6
    Z = next(z for z in LIFTING_SIZES if kb * z >= block_bits)
    K = kb_cols * Z
    return Z, K, K - block_bits


def lift(base: QCStructure, Z: int) -> QCStructure:
    """A base table (shift coefficients V, lifting size of the table >= any shift) at lifting size Z:
    P = V mod Z (38.212 section 5.3.2)."""
    return QCStructure(Z, base.mb, base.nb, base.edge_row.copy(), base.edge_col.copy(),
                       (base.edge_shift % Z).astype(np.int32))


def active_rows(base_graph: int, Z: int, highest_position: int) -> int:
    """Block rows of H a receiver needs when codeword positions above ``highest_position`` were never
    sent: the four core rows plus one row per extension parity block that was (partly) sent."""
    kb_cols, nb, mb, _ = BASE_GRAPHS[base_graph]
    last_block = highest_position // Z
    return int(min(mb, max(4, last_block - kb_cols + 1)))


def active_submatrix(qc: QCStructure, rows: int) -> QCStructure:
    """First ``rows`` block rows of H and the block columns they touch (information + core parity +
    their own extension columns): H[:rows Z, :(nb - mb + rows) Z]."""
    keep = qc.edge_row < rows
    nb = qc.nb - qc.mb + rows
    assert (qc.edge_col[keep] < nb).all(), "not an NR-shaped table: a kept row reaches a dropped column"
    return QCStructure(qc.Z, rows, nb, qc.edge_row[keep].copy(), qc.edge_col[keep].copy(), qc.edge_shift[keep].copy())


@dataclass(frozen=True)
class RateMatch:
    """Rate matching of one code block: ``E`` transmitted bits out of the circular buffer."""
    base_graph: int
    Z: int
    E: int
    rv: int = 0
    fillers: int = 0
    Qm: int = 1
    n_cb: Optional[int] = None     # limited buffer (I_LBRM = 1); default: the whole codeword

    def __post_init__(self):
        kb_cols, nb, _, _ = BASE_GRAPHS[self.base_graph]
        N = (nb - 2) * self.Z
        if self.n_cb is None:
            object.__setattr__(self, "n_cb", N)
        if not 0 < self.n_cb <= N or self.E < 1 or self.E % self.Qm or self.rv not in (0, 1, 2, 3):
            raise ValueError("bad rate matching parameters")
        if not 0 <= self.fillers <= (kb_cols - 2) * self.Z:   # fillers never reach into the punctured 2 Z prefix
            raise ValueError("bad filler count")

    @property
    def offset(self) -> int:                     # the first 2 Z systematic bits are never in the buffer
        return 2 * self.Z

    @property
    def K(self) -> int:
        return BASE_GRAPHS[self.base_graph][0] * self.Z

    @property
    def n(self) -> int:
        return BASE_GRAPHS[self.base_graph][1] * self.Z

    @property
    def k0(self) -> int:
        num = K0_NUMERATORS[self.base_graph][self.rv]
        den = BASE_GRAPHS[self.base_graph][1] - 2
        return (num * self.n_cb // (den * self.Z)) * self.Z

    @property
    def filler_range(self) -> Tuple[int, int]:   # in buffer positions
        return self.K - self.fillers - self.offset, self.K - self.offset

    def positions_by_walk(self) -> np.ndarray:
        """Buffer position of e_k, k = 0 .. E-1: the loop of 38.212 section 5.4.2.1 restated literally."""
        fa, fb = self.filler_range
        out = np.empty(self.E, np.int64)
        k = j = 0
        while k < self.E:
            d = (self.k0 + j) % self.n_cb
            if not fa <= d < fb:
                out[k] = d
                k += 1
            j += 1
        return out

    def positions(self) -> np.ndarray:
        """The same in closed form (what the device kernels compute): positions are ranked among the
        non-filler positions of the buffer, e_k is the one of rank (rank(k0) + k) mod M."""
        fa, fb = (min(x, self.n_cb) for x in self.filler_range)
        nf = max(fb - fa, 0)
        M = self.n_cb - nf
        rank0 = self.k0 - min(max(self.k0 - fa, 0), nf)
        r = (rank0 + np.arange(self.E, dtype=np.int64)) % M
        return np.where(r < fa, r, r + nf)

    def air_order(self) -> np.ndarray:
        """Index in the transmitted sequence f of e_k (section 5.4.2.2: f[i + j Qm] = e[i E/Qm + j])."""
        k = np.arange(self.E)
        rows = self.E // self.Qm
        return (k // rows) + (k % rows) * self.Qm

    def highest_position(self) -> int:
        """Highest codeword position that is sent or known: decides the active rows."""
        return int(self.positions().max()) + self.offset

    def rows_needed(self) -> int:
        return active_rows(self.base_graph, self.Z, max(self.highest_position(), self.K - 1))

    def n_active(self) -> int:
        kb_cols = BASE_GRAPHS[self.base_graph][0]
        return (kb_cols + self.rows_needed()) * self.Z

    # ---- numpy restatements of the device kernels (test infrastructure lives on this, not the product)
    def match_host(self, cw: np.ndarray) -> np.ndarray:
        cw = np.asarray(cw)
        out = np.empty((cw.shape[0], self.E), cw.dtype)
        out[:, self.air_order()] = cw[:, self.positions() + self.offset]
        return out

    def recover_host(self, llr: np.ndarray, n_out: int, filler_llr: float, into: Optional[np.ndarray] = None) -> np.ndarray:
        llr = np.asarray(llr, dtype=np.float64)
        out = np.zeros((llr.shape[0], n_out)) if into is None else np.array(into, dtype=np.float64)
        e = llr[:, self.air_order()]
        pos = self.positions() + self.offset
        for k in range(self.E):          # in transmission order, as a receiver would add them up
            if pos[k] < n_out:
                out[:, pos[k]] += e[:, k]
        fa, fb = self.filler_range
        out[:, fa + self.offset:min(fb + self.offset, n_out)] = filler_llr
        return out

    # ---- device
    def c_config(self, n_out: int):
        from . import _lib
        fa, fb = self.filler_range
        return _lib.CRateMatchConfig(self.n_cb, self.k0, fa, fb, self.E, self.Qm, self.offset, int(n_out))

    def match_device(self, cw):
        """cw: torch uint8 [frames, n] on the device -> transmitted bits [frames, E]."""
        import torch
        from . import _lib
        cw = cw.contiguous()
        out = torch.empty((cw.shape[0], self.E), dtype=torch.uint8, device=cw.device)
        cfg = self.c_config(cw.shape[1])
        _lib.check(_lib.load().ldpc_rate_match(ctypes.byref(cfg), cw.data_ptr(), cw.shape[1], cw.shape[0], out.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream or None))
        return out

    def recover_device(self, llr, n_out: int, filler_llr: float = 100.0, into=None, out_dtype=None, out=None):
        """llr: torch float [frames, E] -> LLRs of codeword positions [0, n_out); ``into`` = a buffer of
        an earlier transmission to combine with, ``out`` = a buffer to overwrite."""
        import torch
        from . import _lib
        llr = llr.contiguous()
        if into is not None:
            out = into
        elif out is None:
            out = torch.empty((llr.shape[0], n_out), dtype=out_dtype or llr.dtype, device=llr.device)
        assert out.shape == (llr.shape[0], n_out) and out.is_contiguous()
        cfg = self.c_config(n_out)
        dt = lambda t: 1 if t.dtype == torch.float64 else 0
        _lib.check(_lib.load().ldpc_rate_recover(ctypes.byref(cfg), llr.data_ptr(), dt(llr), llr.shape[0], float(filler_llr),
                                                 int(into is not None), out.data_ptr(), dt(out),
                                                 torch.cuda.current_stream().cuda_stream or None))
        return out
