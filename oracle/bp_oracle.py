"""TEST INFRASTRUCTURE — CPU oracles for the belief-propagation decode path.

Two independent restatements of the reference's decoders live here; only tests/,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of bench.py may use them, as the checker:

* ``numpy_decode``  — batched numpy float64 restatement.  Every floating-point operation is the
  same numpy operation, applied in the same order, as in the reference, so the results are meant to
  be identical to the reference's to the last bit (pinned by tests/golden, which holds outputs of
  the reference's own methods produced by oracle/gen_golden.py).
* ``c_decode``      — ctypes front end of oracle/bp_oracle.c (scalar C, OpenMP over frames), the
  fast checker for large batches and the CPU baseline.  Bit-exact for R0/R1; rounding-level
  differences for the sum-product rules (libm instead of numpy transcendentals).

Rules follow /root/reference/ldpc.py:137-227 (R0), bpsk.py:154-260 (R1), bpsk.py:263-387 (R2),
channel.py:402-513 (R3), channel.py:260-398 (R4); syndrome ldpc.py:76-80.  Rule 5 (row-layered
min-sum) and rule 6 (offset min-sum, flooding, R1's conventions) are build extensions with no
counterpart in the reference: for them these oracles are the definition, not a restatement.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


# =========================================================================== numpy restatement
def _groups(ptr: np.ndarray):
    """Yield (degree, node indices) for nodes grouped by degree."""
    deg = np.diff(ptr)
    for d in np.unique(deg):
        yield int(d), np.flatnonzero(deg == d)


def _np_sum_seq(terms):
    """numpy add.reduce order over a short contiguous vector (pairwise, 8 accumulators)."""
    n = len(terms)
    if n < 8:
        res = 0.0
        for t in terms:
            res = res + t
        return res
    r = list(terms[:8])
    i = 8
    while i < n - (n % 8):
        for k in range(8):
            r[k] = r[k] + terms[i + k]
        i += 8
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    while i < n:
        res = res + terms[i]
        i += 1
    return res


def _phi(x):
    # channel.py:392-398
    x = np.clip(x, 1e-12, 100)
    return np.log((np.exp(x) + 1) / (np.exp(x) - 1))


def llr_from_samples(y: np.ndarray, snr_db, rule: int) -> np.ndarray:
    """Channel LLRs exactly as each decode method forms them from channel samples.

    R0 ldpc.py:143-144, R1 bpsk.py:162-167 (first 80 set to 1e-9), R2 bpsk.py:271-277 (no factor
    2, no puncturing), R3 channel.py:410-416 and R4 channel.py:268-274 (first 80 set to 0.0).
    ``snr_db`` may be a scalar or one value per frame.
    """
    y = np.asarray(y, dtype=np.float64)
    snr = np.asarray(snr_db, dtype=np.float64)
    sigma2 = 1 / (2 * (10 ** (snr / 10)))
    if sigma2.ndim:
        sigma2 = sigma2[..., None]
    if rule == 2:
        L = y / sigma2
    else:
        L = (2 * y) / sigma2
    if rule == 1:
        L[..., :80] = 1e-9
    elif rule in (3, 4):
        L[..., :80] = 0.0
    return L


def numpy_decode(code, cfg, llr: np.ndarray, want_trace: bool = False):
    """Decode llr[B, n] (float64).  Returns (hard u8[B,n], iters i32[B], converged bool[B],
    posterior f64[B,n]) and, with ``want_trace``, the list of posteriors seen at each top-of-loop
    check while at least one frame was still running (frames that finished keep their last value).
    """
    L = np.array(llr, dtype=np.float64, ndmin=2)
    B, n = L.shape
    assert n == code.n
    rule = int(cfg.rule)
    row_ptr, col_idx, col_ptr, csc_edge = code.row_ptr, code.col_idx, code.col_ptr, code.csc_edge
    E = code.num_edges
    m = code.m
    mact = m if cfg.num_active_checks < 0 else min(m, cfg.num_active_checks)
    nref = n if cfg.num_refresh_vars < 0 else min(n, cfg.num_refresh_vars)

    hard_out = np.zeros((B, n), np.uint8)
    iters_out = np.zeros(B, np.int32)
    conv_out = np.zeros(B, bool)
    post_out = np.zeros((B, n), np.float64)
    active = np.arange(B)           # indices of frames still running
    trace = []

    def hard_of(b):
        return np.where(b < 0, 0, 1).astype(np.uint8) if rule == 0 else np.where(b > 0, 0, 1).astype(np.uint8)

    def retire(mask, t, b, hard):
        nonlocal active, L
        idx = active[mask]
        hard_out[idx] = hard[mask]
        iters_out[idx] = t
        conv_out[idx] = True
        post_out[idx] = b[mask]
        keep = ~mask
        active = active[keep]
        return keep

    b = L.copy()
    Lc = L
    q = None if rule == 0 else Lc[:, col_idx].copy()        # v2c in CSR order
    r = np.zeros((B, E))                                      # c2v in CSR order
    hard = hard_of(b)
    row_groups = list(_groups(row_ptr))
    col_groups = list(_groups(col_ptr))

    for t in range(cfg.max_iters):
        if want_trace:
            snap = post_out.copy()
            snap[active] = b
            trace.append(snap)
        if cfg.early_exit:
            if rule == 0 and t == 0 and cfg.float_syndrome0:
                # numpy.dot(H, llr) % 2 == 0 on the float LLRs (ldpc.py:158,165)
                ok = np.ones(len(active), bool)
                for j in range(m):
                    s = np.zeros(len(active))
                    for e in range(row_ptr[j], row_ptr[j + 1]):
                        s = s + b[:, col_idx[e]]
                    ok &= (np.fmod(s, 2.0) == 0)
                special = np.where(b == 0.0, 0, np.where(b == 1.0, 1, 2)).astype(np.uint8)
                done = ok
                hard_for_exit = special
            else:
                done = code.is_codeword(hard)
                hard_for_exit = hard
            if done.any():
                keep = retire(done, t, b, hard_for_exit)
                b, Lc, r, hard = b[keep], Lc[keep], r[keep], hard[keep]
                if q is not None:
                    q = q[keep]
            if len(active) == 0:
                break

        # ---------------------------------------------------------------- check nodes
        src = b[:, col_idx] if rule == 0 else q              # [Bact, E] inputs in CSR order
        for d, rows in row_groups:
            if d == 0:
                continue
            base = row_ptr[rows]                              # first edge of each check
            if rule == 0:
                live = rows < mact
                if d < 2:
                    r[:, base] = 0.0
                    continue
            if rule in (1, 6) and d < 2:
                continue
            vals = [src[:, base + p] for p in range(d)]       # each [Bact, nchecks]
            for tp in range(d):
                others = [vals[p] for p in range(d) if p != tp]
                if rule in (0, 1, 6):
                    mn = None
                    sg = None
                    for v in others:
                        a = np.abs(v)
                        mn = a if mn is None else np.minimum(mn, a)
                        s = np.sign(v)
                        if rule != 0:
                            s = np.where(s == 0.0, 1.0, s)
                        sg = s if sg is None else sg * s
                    if rule == 6:   # offset min-sum (build extension): max(scale * min - offset, 0) * sign
                        out = np.maximum(cfg.scale * mn - cfg.offset, 0.0) * sg
                    else:
                        out = mn * sg if rule == 0 else cfg.scale * mn * sg
                    if rule == 0:
                        out = np.where(live[None, :], out, 0.0)
                elif rule in (2, 3):
                    prod = None
                    for v in others:
                        a = v / 2
                        if cfg.clip_tanh_arg > 0:
                            a = np.clip(a, -cfg.clip_tanh_arg, cfg.clip_tanh_arg)
                        th = np.tanh(a)
                        prod = th if prod is None else prod * th
                    if prod is None:
                        prod = np.ones((len(active), len(rows)))
                    prod = np.clip(prod, -cfg.clip_prod, cfg.clip_prod)
                    out = 2 * np.arctanh(prod) if rule == 2 else cfg.scale * 2 * np.arctanh(prod)
                else:
                    sg = None
                    terms = []
                    for v in others:
                        terms.append(_phi(np.abs(v)))
                        s = np.where((v == 0) | (v == 1e-9), 1.0, np.sign(v))
                        sg = s if sg is None else sg * s
                    if sg is None:
                        sg = np.ones((len(active), len(rows)))
                        ps = np.zeros((len(active), len(rows)))
                    else:
                        ps = _np_sum_seq(terms)
                    ps = np.clip(ps, 1e-12, 20)
                    out = sg * _phi(ps)
                r[:, base + tp] = out

        # ---------------------------------------------------------------- variable nodes
        newb = Lc.copy()
        for d, cols in col_groups:
            if d == 0:
                continue
            base = col_ptr[cols]
            msgs = [r[:, csc_edge[base + k]] for k in range(d)]
            s = 0.0
            for v in msgs:
                s = s + v
            if rule == 0:
                newb[:, cols] = cfg.scale * (s + Lc[:, cols])
            else:
                newb[:, cols] = Lc[:, cols] + s
                refresh = cols < nref
                for k in range(d):
                    e = 0.0
                    for kk in range(d):
                        if kk != k:
                            e = e + msgs[kk]
                    tgt = csc_edge[base + k]
                    val = Lc[:, cols] + e
                    q[:, tgt[refresh]] = val[:, refresh]
        if rule == 0:
            zero_deg = np.flatnonzero(np.diff(col_ptr) == 0)
            if zero_deg.size:
                newb[:, zero_deg] = cfg.scale * (0.0 + Lc[:, zero_deg])
        if cfg.clip_post > 0:
            newb = np.clip(newb, -cfg.clip_post, cfg.clip_post)
        b = newb
        hard = hard_of(b)

    # frames that never passed a top-of-loop check
    if len(active):
        hard_out[active] = hard
        post_out[active] = b
        if cfg.early_exit:
            iters_out[active] = cfg.max_iters - 1
            conv_out[active] = False
        else:
            iters_out[active] = cfg.max_iters
            conv_out[active] = code.is_codeword(hard)
    if want_trace:
        return hard_out, iters_out, conv_out, post_out, trace
    return hard_out, iters_out, conv_out, post_out


# =========================================================================== C restatement
class _OracleCfg(ctypes.Structure):
    _fields_ = [("rule", ctypes.c_int32), ("max_iters", ctypes.c_int32),
                ("early_exit", ctypes.c_int32), ("float_syndrome0", ctypes.c_int32),
                ("num_active_checks", ctypes.c_int32), ("num_refresh_vars", ctypes.c_int32),
                ("scale", ctypes.c_double), ("clip_tanh_arg", ctypes.c_double),
                ("clip_prod", ctypes.c_double), ("clip_post", ctypes.c_double), ("offset", ctypes.c_double)]


class _OracleCode(ctypes.Structure):
    _fields_ = [("m", ctypes.c_int32), ("n", ctypes.c_int32), ("E", ctypes.c_int32),
                ("row_ptr", ctypes.c_void_p), ("col_idx", ctypes.c_void_p),
                ("col_ptr", ctypes.c_void_p), ("csc_edge", ctypes.c_void_p)]


_clib = None


def build_c_oracle(force: bool = False) -> str:
    """Compile oracle/bp_oracle.c into oracle/libbp_oracle.so (gcc, OpenMP)."""
    so = os.path.join(_HERE, "libbp_oracle.so")
    src = os.path.join(_HERE, "bp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-ffp-contract=off",
                               "-o", so, src, "-lm"])
    return so


def _load_c():
    global _clib
    if _clib is None:
        so = os.path.join(_HERE, "libbp_oracle.so")
        if not os.path.exists(so):
            build_c_oracle()
        _clib = ctypes.CDLL(so)
        _clib.bp_oracle_decode.restype = ctypes.c_int
        _clib.bp_oracle_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _clib.bp_oracle_decode_lnms.restype = ctypes.c_int
        _clib.bp_oracle_decode_lnms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        _clib.bp_oracle_max_threads.restype = ctypes.c_int
    return _clib


def numpy_decode_lnms(code, cfg, llr, dtype=np.float64):
    """Row-layered normalised min-sum (build extension, no counterpart in the reference): checks in
    ascending order on the current posteriors; q = P - r_old (the channel LLR itself for a variable
    that takes part in this check only), r_new = scale * min-of-others * sign, P = q + r_new.
    Syndrome test at the top of each iteration, R1's polarity and conventions."""
    L = np.array(llr, dtype=np.float64, ndmin=2)
    B, n = L.shape
    P = L.astype(dtype) + dtype(0)
    L0 = P.copy()
    lone = np.diff(code.col_ptr) == 1
    R = np.zeros((B, code.num_edges), dtype)
    scale = dtype(cfg.scale)
    hard = np.where(P > 0, 0, 1).astype(np.uint8)
    iters = np.full(B, cfg.max_iters - 1 if cfg.early_exit else cfg.max_iters, np.int32)
    conv = np.zeros(B, bool)
    live = np.ones(B, bool)
    hard_out = hard.copy()
    post_out = P.copy()
    for t in range(cfg.max_iters):
        if cfg.early_exit:
            ok = code.is_codeword(hard) & live
            iters[ok] = t
            conv[ok] = True
            hard_out[ok] = hard[ok]
            post_out[ok] = P[ok]
            live &= ~ok
            if not live.any():
                break
        for j in range(code.m):
            e0, e1 = code.row_ptr[j], code.row_ptr[j + 1]
            d = e1 - e0
            if d < 2:
                continue
            cols = code.col_idx[e0:e1]
            q = np.where(lone[cols][None, :], L0[:, cols], P[:, cols] - R[:, e0:e1])
            a = np.abs(q)
            srt = np.sort(a, axis=1)
            m1, m2 = srt[:, 0], srt[:, 1]
            neg = (q < 0)
            par = np.logical_xor.reduce(neg, axis=1)
            mag = np.where(a == m1[:, None], (scale * m2)[:, None], (scale * m1)[:, None])
            r = np.where(par[:, None] ^ neg, -mag, mag).astype(dtype)
            R[:, e0:e1] = r
            P[:, cols] = q + r
        hard = np.where(P > 0, 0, 1).astype(np.uint8)
    hard_out[live] = hard[live]
    post_out[live] = P[live]
    if not cfg.early_exit:
        conv = code.is_codeword(hard)
    return hard_out, iters, conv, post_out.astype(np.float64)


def row_prefix_lists(code, rows: int):
    """Edge lists (row_ptr, col_idx, col_ptr, csc_edge) of the first ``rows`` checks of ``code`` on all
    of its n columns: the matrix a layered decode with num_active_checks = rows sees."""
    E = int(code.row_ptr[rows])
    row_ptr = np.asarray(code.row_ptr[:rows + 1], np.int32)
    col_idx = np.asarray(code.col_idx[:E], np.int32)
    order = np.argsort(col_idx, kind="stable")     # by column; rows ascend inside a column
    col_ptr = np.zeros(code.n + 1, np.int32)
    np.cumsum(np.bincount(col_idx, minlength=code.n), out=col_ptr[1:])
    return row_ptr, col_idx, col_ptr, order.astype(np.int32)


def c_decode_lnms(code, cfg, llr, nthreads: int = 0, use_float: bool = False, want_posterior: bool = True):
    """cfg.num_active_checks in [0, m): the decode runs on the sub-matrix of those rows (sweeps, syndrome
    and variable degrees), as LDPC_RULE_LNMS defines it in include/ldpc_amd.h."""
    lib = _load_c()
    L = np.ascontiguousarray(np.array(llr, dtype=np.float64, ndmin=2))
    B, n = L.shape
    assert n == code.n
    m, E = code.m, code.num_edges
    lists = (code.row_ptr, code.col_idx, code.col_ptr, code.csc_edge)
    if 0 <= cfg.num_active_checks < code.m:
        m = int(cfg.num_active_checks)
        lists = row_prefix_lists(code, m)
        E = int(lists[0][m])
    arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in lists]
    ccode = _OracleCode(m, code.n, E, *[a.ctypes.data for a in arrs])
    ccfg = _OracleCfg(int(cfg.rule), int(cfg.max_iters), int(bool(cfg.early_exit)), 0, -1, -1,
                      float(cfg.scale), 0.0, 0.0, 0.0)
    hard = np.zeros((B, n), np.uint8)
    iters = np.zeros(B, np.int32)
    conv = np.zeros(B, np.uint8)
    post = np.zeros((B, n), np.float64) if want_posterior else None
    rc = lib.bp_oracle_decode_lnms(ctypes.byref(ccode), ctypes.byref(ccfg), L.ctypes.data, B,
                                   hard.ctypes.data, iters.ctypes.data, conv.ctypes.data,
                                   post.ctypes.data if post is not None else None, int(nthreads),
                                   int(bool(use_float)))
    if rc != 0:
        raise ValueError("bp_oracle_decode_lnms rejected its arguments")
    return hard, iters, conv.astype(bool), post


def numpy_bitflip(code, cfg, samples):
    """Bit flipping as ldpc.py:82-135 does it, one frame at a time with the reference's own loops
    over numpy arrays.  Returns (hard u8[B,n], iters i32[B], converged bool[B])."""
    y = np.array(samples, dtype=np.float64, ndmin=2)
    B, n = y.shape
    m = code.m
    mact = m if cfg.num_active_checks < 0 or cfg.num_active_checks > m else int(cfg.num_active_checks)
    hard = np.zeros((B, n), np.uint8)
    iters = np.zeros(B, np.int32)
    conv = np.zeros(B, bool)
    rows = [code.col_idx[code.row_ptr[j]:code.row_ptr[j + 1]] for j in range(m)]
    for f in range(B):
        bits = np.where(y[f] > 0, 0, 1).astype(np.int64)             # ldpc.py:86-90
        t = 0
        while t < cfg.max_iters:                                     # ldpc.py:102
            if not code.syndrome(bits).any():                        # ldpc.py:103
                conv[f] = True
                break
            msgs = bits.copy()
            votes = np.zeros(n, np.int64)
            Ej = np.zeros(0, np.int64)
            for j in range(mact):                                    # ldpc.py:113
                Ej = rows[j]
                par = int(np.bitwise_xor.reduce(msgs[Ej])) if len(Ej) else 0
                votes[Ej] += np.where((par ^ msgs[Ej]) == 1, 1, -1)  # ldpc.py:123-128
            bits = np.where(votes > len(Ej) / 2, bits ^ 1, bits)      # ldpc.py:130-132
            t += 1
        hard[f] = bits
        iters[f] = t
    return hard, iters, conv


def c_bitflip(code, cfg, samples, nthreads: int = 0):
    lib = _load_c()
    y = np.ascontiguousarray(np.array(samples, dtype=np.float64, ndmin=2))
    B, n = y.shape
    arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (code.row_ptr, code.col_idx, code.col_ptr, code.csc_edge)]
    ccode = _OracleCode(code.m, code.n, code.num_edges, *[a.ctypes.data for a in arrs])
    ccfg = _OracleCfg(7, int(cfg.max_iters), 1, 0, int(cfg.num_active_checks), -1, 1.0, 0.0, 0.0, 0.0, 0.0)
    hard = np.zeros((B, n), np.uint8)
    iters = np.zeros(B, np.int32)
    conv = np.zeros(B, np.uint8)
    lib.bp_oracle_bitflip.restype = ctypes.c_int
    lib.bp_oracle_bitflip.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    rc = lib.bp_oracle_bitflip(ctypes.byref(ccode), ctypes.byref(ccfg), y.ctypes.data, B, hard.ctypes.data,
                               iters.ctypes.data, conv.ctypes.data, int(nthreads))
    if rc != 0:
        raise ValueError("bp_oracle_bitflip rejected its arguments")
    return hard, iters, conv.astype(bool)


def c_max_threads() -> int:
    return int(_load_c().bp_oracle_max_threads())


def c_decode(code, cfg, llr: np.ndarray, nthreads: int = 0, want_posterior: bool = True
             ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, Optional[np.ndarray]]:
    lib = _load_c()
    L = np.ascontiguousarray(np.array(llr, dtype=np.float64, ndmin=2))
    B, n = L.shape
    assert n == code.n
    arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in
            (code.row_ptr, code.col_idx, code.col_ptr, code.csc_edge)]
    ccode = _OracleCode(code.m, code.n, code.num_edges, *[a.ctypes.data for a in arrs])
    ccfg = _OracleCfg(int(cfg.rule), int(cfg.max_iters), int(bool(cfg.early_exit)),
                      int(bool(cfg.float_syndrome0)), int(cfg.num_active_checks),
                      int(cfg.num_refresh_vars), float(cfg.scale), float(cfg.clip_tanh_arg),
                      float(cfg.clip_prod), float(cfg.clip_post), float(getattr(cfg, "offset", 0.0)))
    hard = np.zeros((B, n), np.uint8)
    iters = np.zeros(B, np.int32)
    conv = np.zeros(B, np.uint8)
    post = np.zeros((B, n), np.float64) if want_posterior else None
    rc = lib.bp_oracle_decode(ctypes.byref(ccode), ctypes.byref(ccfg), L.ctypes.data, B,
                              hard.ctypes.data, iters.ctypes.data, conv.ctypes.data,
                              post.ctypes.data if post is not None else None, int(nthreads))
    if rc != 0:
        raise ValueError("bp_oracle_decode rejected its arguments")
    return hard, iters, conv.astype(bool), post
