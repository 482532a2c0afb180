"""Encoding on the device for circulant codes (``ldpc_encode_batch``).

The reference encodes with a dense generator, ``numpy.dot(self.G, message) % 2``
(/root/reference/ldpc.py:46), where G comes from ``pyldpc.coding_matrix`` (ldpc.py:369).  For a
lifted 5G matrix that G would be 8448 x 26112 bits; a circulant code needs none.  This module turns
a code into a SCHEDULE over block columns (Z-vectors of bits) that the device kernel executes:

    step:  x[out] = XOR over terms (col, s) of roll(x[col], -s)      out[z] ^= col[(z + s) mod Z]

* ``parity_structure_schedule``: works on the circulant description of H.  Repeats two moves until
  every block column is known: (a) a block row with exactly one unknown block column gives that
  column by substitution; (b) when every row has two or more unknowns (the start of a dual-diagonal
  parity part: 5G NR, 802.11n), a leading group of rows is added up - columns that appear twice with
  equal shifts cancel - and if exactly one circulant of one unknown column is left, that column
  follows from the known ones.
* ``generator_schedule``: for a systematic generator [I | P] whose P is made of Z x Z circulants
  (CCSDS): parity block j is the sum over information blocks i and the taps of circulant (i, j).

Nothing in the reference pins a generator (pyldpc is absent and unpinned): the encoders are checked
by H c = 0, by systematic positions, and against this build's host encoders.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .codes import CompiledCode, QCStructure


@dataclass
class Schedule:
    Z: int
    nb: int                       # block columns of the codeword
    n_cols: int                   # ... plus temporaries
    info_cols: List[int]          # block column of each information block, in message order
    step_out: List[int]
    step_level: List[int]
    step_terms: List[List[Tuple[int, int]]]   # per step: (block column, shift)

    @property
    def k(self) -> int:
        return len(self.info_cols) * self.Z

    @property
    def n(self) -> int:
        return self.nb * self.Z

    def run_host(self, msgs: np.ndarray) -> np.ndarray:
        """The schedule executed with numpy (the checker of the device kernel)."""
        msgs = np.asarray(msgs, dtype=np.uint8) & 1
        B = msgs.shape[0]
        x = np.zeros((B, self.n_cols, self.Z), np.uint8)
        x[:, self.info_cols] = msgs.reshape(B, len(self.info_cols), self.Z)
        for out, terms in zip(self.step_out, self.step_terms):
            acc = np.zeros((B, self.Z), np.uint8)
            for col, s in terms:
                acc ^= np.roll(x[:, col], -s, axis=1)
            x[:, out] = acc
        return x[:, :self.nb].reshape(B, self.n)


def _levels(info_cols: Sequence[int], n_cols: int, outs, terms) -> List[int]:
    """Level of every step: one more than the highest level among the columns it reads."""
    known = {c: -1 for c in info_cols}
    levels = []
    for out, tt in zip(outs, terms):
        lv = 1 + max([known[c] for c, _ in tt], default=-1)
        levels.append(lv)
        known[out] = lv
    return levels


def _sorted_by_level(sched_outs, sched_terms, levels):
    order = sorted(range(len(levels)), key=lambda i: (levels[i], i))
    return [sched_outs[i] for i in order], [sched_terms[i] for i in order], [levels[i] for i in order]


def parity_structure_schedule(qc: QCStructure, info_cols: Optional[Sequence[int]] = None) -> Optional[Schedule]:
    """Schedule from the circulant description of H, or None if the two moves get stuck."""
    Z, mb, nb = qc.Z, qc.mb, qc.nb
    rows: List[List[Tuple[int, int]]] = [[] for _ in range(mb)]
    for r, c, s in zip(qc.edge_row.tolist(), qc.edge_col.tolist(), qc.edge_shift.tolist()):
        rows[r].append((c, s))
    if info_cols is None:
        info_cols = list(range(nb - mb))
    known = set(info_cols)
    used_rows: set = set()
    outs, terms = [], []
    while len(known) < nb:
        progressed = False
        # (a) substitution: a row with one unknown column that holds a single circulant
        for r in range(mb):
            if r in used_rows:
                continue
            unknown = [(c, s) for c, s in rows[r] if c not in known]
            if len(unknown) == 1:
                c0, s0 = unknown[0]
                outs.append(c0)
                terms.append([(c, (s - s0) % Z) for c, s in rows[r] if c != c0])
                known.add(c0)
                used_rows.add(r)
                progressed = True
        if progressed:
            continue
        # (b) add up a leading group of unused rows; equal (column, shift) pairs cancel
        free = [r for r in range(mb) if r not in used_rows]
        for g in range(2, len(free) + 1):
            tally = {}
            for r in free[:g]:
                for c, s in rows[r]:
                    tally[(c, s)] = tally.get((c, s), 0) ^ 1
            left_unknown = [(c, s) for (c, s), v in tally.items() if v and c not in known]
            if len(left_unknown) == 1:
                c0, s0 = left_unknown[0]
                outs.append(c0)
                terms.append(sorted((c, (s - s0) % Z) for (c, s), v in tally.items() if v and c in known))
                known.add(c0)
                progressed = True
                break
        if not progressed:
            return None
    levels = _levels(info_cols, nb, outs, terms)
    outs, terms, levels = _sorted_by_level(outs, terms, levels)
    return Schedule(Z, nb, nb, list(info_cols), outs, levels, terms)


def generator_schedule(G: np.ndarray, Z: int) -> Optional[Schedule]:
    """Schedule from a systematic generator [I | P] (k x n) whose P consists of Z x Z circulants."""
    G = np.asarray(G, dtype=np.uint8)
    k, n = G.shape
    if k % Z or n % Z or not np.array_equal(G[:, :k], np.eye(k, dtype=np.uint8)):
        return None
    kb, nb = k // Z, n // Z
    outs, terms = [], []
    for j in range(kb, nb):
        tt = []
        for i in range(kb):
            blk = G[i * Z:(i + 1) * Z, j * Z:(j + 1) * Z]
            first = blk[0]
            for r in range(1, Z):
                if not np.array_equal(blk[r], np.roll(first, r)):
                    return None
            # p_j[z] += s_i[(z - t) mod Z] for every tap t of the first row
            tt += [(i, int((Z - t) % Z)) for t in np.flatnonzero(first)]
        outs.append(j)
        terms.append(tt)
    return Schedule(Z, nb, nb, list(range(kb)), outs, [0] * len(outs), terms)


def schedule_for(code: CompiledCode, G: Optional[np.ndarray] = None) -> Optional[Schedule]:
    """The schedule of a compiled code: from its parity structure if that works, else from a
    block-circulant systematic generator, else None (dense codes keep the generator product)."""
    if code.qc is None:
        return None
    pairs = {}
    for r, c in zip(code.qc.edge_row.tolist(), code.qc.edge_col.tolist()):
        pairs[(r, c)] = pairs.get((r, c), 0) + 1
    candidates = []
    if max(pairs.values()) == 1:      # blocks that are sums of two circulants cannot be substituted through
        candidates.append(parity_structure_schedule(code.qc))
    if G is not None:
        candidates.append(generator_schedule(G, code.qc.Z))
    for s in candidates:
        if s is None or s.n != code.n:
            continue
        # a schedule is only handed out if what it produces are codewords
        msgs = np.random.default_rng(5).integers(0, 2, size=(4, s.k), dtype=np.uint8)
        if all(code.is_codeword(c) for c in s.run_host(msgs)):
            return s
    return None


class DeviceEncoder:
    """message -> codeword on the device.  Callable on torch uint8 tensors [frames, k] (what
    ``sweep.make_codewords`` takes in place of a dense generator matrix); ``encode_ptr`` takes raw
    device pointers."""

    on_device = True

    def __init__(self, schedule: Schedule, device: int = -1):
        from . import _lib
        self.schedule = schedule
        self.k, self.n = schedule.k, schedule.n
        self._lib = _lib.load()
        i32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.int32))
        ptr = np.zeros(len(schedule.step_out) + 1, np.int32)
        ptr[1:] = np.cumsum([len(t) for t in schedule.step_terms])
        flat = [ct for t in schedule.step_terms for ct in t]
        info, outs, lv = i32(schedule.info_cols), i32(schedule.step_out), i32(schedule.step_level)
        tc, ts = i32([c for c, _ in flat]), i32([s for _, s in flat])
        self._h = ctypes.c_void_p()
        _lib.check(self._lib.ldpc_encoder_create(schedule.Z, schedule.n_cols, schedule.nb, len(schedule.info_cols),
                                                 info.ctypes.data, len(outs), outs.ctypes.data, ptr.ctypes.data,
                                                 lv.ctypes.data, tc.ctypes.data, ts.ctypes.data, int(device),
                                                 ctypes.byref(self._h)))

    def encode_ptr(self, msg_ptr: int, frames: int, cw_ptr: int, stream: int = 0) -> None:
        from . import _lib
        _lib.check(self._lib.ldpc_encode_batch(self._h, msg_ptr, int(frames), cw_ptr, stream or None))

    def random_codewords(self, frames: int, seed, first_frame: int = 0, device=None, want_messages: bool = False):
        """Codewords of random messages drawn on the device (``ldpc_encode_random``): frame f gets the
        message of absolute frame ``first_frame + f`` under ``seed``, whatever launch produces it."""
        import torch
        device = device or torch.device("cuda", torch.cuda.current_device())
        cw = torch.empty((frames, self.n), dtype=torch.uint8, device=device)
        msg = torch.empty((frames, self.k), dtype=torch.uint8, device=device) if want_messages else None
        key = int(np.random.SeedSequence(seed).generate_state(1, dtype=np.uint64)[0])
        from . import _lib
        _lib.check(self._lib.ldpc_encode_random(self._h, key, int(first_frame), int(frames), cw.data_ptr(),
                                                msg.data_ptr() if want_messages else None,
                                                torch.cuda.current_stream().cuda_stream or None))
        return (cw, msg) if want_messages else cw

    def __call__(self, msgs):
        import torch
        msgs = msgs.contiguous()
        assert msgs.dtype == torch.uint8 and msgs.shape[1] == self.k
        cw = torch.empty((msgs.shape[0], self.n), dtype=torch.uint8, device=msgs.device)
        self.encode_ptr(msgs.data_ptr(), msgs.shape[0], cw.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return cw

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.ldpc_encoder_destroy(h)
            h.value = None   # (module globals may already be gone at interpreter shutdown)


def device_encoder(code: CompiledCode, G: Optional[np.ndarray] = None, device: int = -1) -> Optional[DeviceEncoder]:
    s = schedule_for(code, G)
    return None if s is None else DeviceEncoder(s, device)
