"""Register a circulant (quasi-cyclic) table so that it gets its own compiled-in kernels.

    python -m ldpc_amd.add_code NAME TABLE.npz [--slots F] [--no-build]
    python -m ldpc_amd.add_code --remove NAME

TABLE.npz holds Z, mb, nb and the base edges edge_row / edge_col / edge_shift (check (r, z) is
joined to variable (c, (z + shift) mod Z)), e.g. a 38.212 base graph with its shift coefficients
taken modulo the lifting size.  The table is copied to ldpc_amd/data, listed in user_codes.json,
and the library is rebuilt: the code generator emits its tables as constant expressions and the
specialised kernels (flooding for small graphs, layered min-sum for any) are instantiated for it.
``codes.load_packaged(NAME)`` then returns it, and a handle whose attached description equals the
table is routed to those kernels.  Without this step any table still decodes through the
table-driven kernels, several times slower.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import shutil

from . import codes


#: frames per workgroup the compiled-in kernels support (size of their per-slot flag arrays)
MAX_SLOTS = 8


def register(name: str, table: str, slots: int = 0) -> dict:
    slots_given = slots > 0
    if not re.fullmatch(r"[A-Za-z][A-Za-z0-9_]*", name):
        raise ValueError("NAME must be a C identifier")
    if name in codes.PACKAGED or name == "nr_bg1_z384":
        raise ValueError(f"{name} is a packaged code")
    qc = codes.load_qc_npz(table)
    codes.compile_qc(qc, name)            # validates the description (raises on repeated edges)
    if slots <= 0:                        # frames per workgroup: fill one wave when Z is small
        slots = max(1, 64 // qc.Z) if qc.Z < 64 else 1
    if slots > MAX_SLOTS:                 # the kernels keep one syndrome flag per slot in a fixed array
        if slots_given:
            raise ValueError(f"--slots {slots}: at most {MAX_SLOTS} frames per workgroup")
        slots = MAX_SLOTS
    fname = f"{name}.qc.npz"
    dst = os.path.join(os.path.dirname(codes.USER_REGISTRY), fname)
    if os.path.abspath(table) != os.path.abspath(dst):
        shutil.copyfile(table, dst)
    reg = codes.user_codes()
    reg[name] = {"file": fname, "slots": int(slots), "Z": qc.Z, "mb": qc.mb, "nb": qc.nb}
    with open(codes.USER_REGISTRY, "w") as f:
        json.dump(reg, f, indent=1, sort_keys=True)
    return reg[name]


def remove(name: str) -> None:
    reg = codes.user_codes()
    rec = reg.pop(name, None)
    if rec is None:
        raise KeyError(name)
    path = os.path.join(os.path.dirname(codes.USER_REGISTRY), rec["file"])
    if os.path.exists(path):
        os.remove(path)
    with open(codes.USER_REGISTRY, "w") as f:
        json.dump(reg, f, indent=1, sort_keys=True)


def main() -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("name")
    ap.add_argument("table", nargs="?")
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--remove", action="store_true")
    ap.add_argument("--no-build", action="store_true")
    a = ap.parse_args()
    if a.remove:
        remove(a.name)
    else:
        if not a.table:
            ap.error("TABLE.npz is required")
        print(register(a.name, a.table, a.slots))
    if not a.no_build:
        from . import build
        build.build()


if __name__ == "__main__":
    main()
