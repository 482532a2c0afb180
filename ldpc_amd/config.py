"""Decode configuration: one frozen record per update rule, mirrored by the POD struct
``ldpc_decode_config`` across the C ABI (include/ldpc_amd.h).

The reference has no configuration system; its constants are edited in the source.  The presets
below carry those constants, with the file:line each one comes from.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, replace

RULE_R0, RULE_R1, RULE_R2, RULE_R3, RULE_R4, RULE_LNMS, RULE_OMS, RULE_BF = range(8)
RULE_NAMES = {"R0": 0, "R1": 1, "R2": 2, "R3": 3, "R4": 4, "LNMS": 5, "OMS": 6, "BF": 7}


@dataclass(frozen=True)
class DecodeConfig:
    rule: int = RULE_R0
    max_iters: int = 10
    #: True: syndrome check at the top of every iteration and none after the last update, so a
    #: frame that only becomes valid on the last update is reported as a failure and ``iters`` is
    #: the reference's ``numIterations`` (0 .. max_iters-1).  False (build extension, fixed work
    #: for throughput runs): all ``max_iters`` updates run, ``iters = max_iters`` and ``converged``
    #: is the syndrome of the final hard decisions.
    early_exit: bool = True
    #: R0 only: the iteration-0 check is evaluated on the raw LLR floats as ldpc.py:158,165 does.
    float_syndrome0: bool = True
    #: R0 only: check-node loop covers rows [0, num_active_checks) (ldpc.py:30,175); -1 = all rows.
    num_active_checks: int = -1
    #: R1-R4: variable-to-check refresh covers variables [0, num_refresh_vars) (bpsk.py:244); -1 = all.
    num_refresh_vars: int = -1
    scale: float = 0.75
    clip_tanh_arg: float = 0.0
    clip_prod: float = 0.0
    clip_post: float = 0.0
    #: OMS only: the offset beta in max(scale * min - offset, 0)
    offset: float = 0.0

    def with_(self, **kw) -> "DecodeConfig":
        return replace(self, **kw)

    @property
    def plus_is_bit(self) -> int:
        """Which bit a positive posterior decodes to: 1 for R0 (ldpc.py:198-201), 0 for R1-R4
        (bpsk.py:240-243)."""
        return 1 if self.rule == RULE_R0 else 0


#: ldpc.py:137-227 — 10 iterations (ldpc.py:161), 0.75 on the whole posterior (ldpc.py:196)
R0 = DecodeConfig(rule=RULE_R0, max_iters=10, scale=0.75)
#: bpsk.py:154-260 — 30 iterations (bpsk.py:200), alpha 0.75 (bpsk.py:233)
R1 = DecodeConfig(rule=RULE_R1, max_iters=30, scale=0.75)
#: bpsk.py:263-387 — 50 iterations (bpsk.py:309), clips bpsk.py:333,337,370
R2 = DecodeConfig(rule=RULE_R2, max_iters=50, scale=1.0, clip_tanh_arg=20.0,
                  clip_prod=0.999999999, clip_post=30.0)
#: channel.py:402-513 — 50 iterations (channel.py:448), 0.85 (channel.py:474), clips channel.py:473,491
R3 = DecodeConfig(rule=RULE_R3, max_iters=50, scale=0.85, clip_prod=0.99999999, clip_post=50.0)
#: channel.py:260-398 — 40 iterations (channel.py:306), clip channel.py:367
R4 = DecodeConfig(rule=RULE_R4, max_iters=40, scale=1.0, clip_post=500.0)

#: row-layered normalised min-sum, alpha 0.75, 10 iterations: build extension, nothing in the
#: reference corresponds to it (see include/ldpc_amd.h)
LNMS = DecodeConfig(rule=RULE_LNMS, max_iters=10, scale=0.75)

#: offset min-sum, flooding, R1's conventions and iteration cap, beta = 0.5 on LLRs 2y/sigma^2: build
#: extension, nothing in the reference corresponds to it (see include/ldpc_amd.h)
OMS = DecodeConfig(rule=RULE_OMS, max_iters=30, scale=1.0, offset=0.5)

#: ldpc.py:82-135 — hard-decision bit flipping, 10 passes (ldpc.py:102)
BF = DecodeConfig(rule=RULE_BF, max_iters=10, scale=1.0, float_syndrome0=False)

PRESETS = {"R0": R0, "R1": R1, "R2": R2, "R3": R3, "R4": R4, "LNMS": LNMS, "OMS": OMS, "BF": BF}


class CDecodeConfig(ctypes.Structure):
    """Binary image of ``ldpc_decode_config`` (include/ldpc_amd.h)."""

    _fields_ = [
        ("rule", ctypes.c_int32),
        ("max_iters", ctypes.c_int32),
        ("early_exit", ctypes.c_int32),
        ("float_syndrome0", ctypes.c_int32),
        ("num_active_checks", ctypes.c_int32),
        ("num_refresh_vars", ctypes.c_int32),
        ("scale", ctypes.c_double),
        ("clip_tanh_arg", ctypes.c_double),
        ("clip_prod", ctypes.c_double),
        ("clip_post", ctypes.c_double),
        ("offset", ctypes.c_double),
    ]

    @classmethod
    def from_config(cls, cfg: DecodeConfig) -> "CDecodeConfig":
        return cls(int(cfg.rule), int(cfg.max_iters), int(bool(cfg.early_exit)),
                   int(bool(cfg.float_syndrome0)), int(cfg.num_active_checks),
                   int(cfg.num_refresh_vars), float(cfg.scale), float(cfg.clip_tanh_arg),
                   float(cfg.clip_prod), float(cfg.clip_post), float(cfg.offset))
