"""In-tree build of libldpc_amd.so (hipcc, gfx950 only).

    python -m ldpc_amd.build [--force] [--jobs N]

hipcc cross-compiles without a GPU; the shared library is written next to this file so that it
travels with the source tree.  Floating-point contraction is switched off: the float64 kernels
must perform the reference's operations one by one to reproduce its results bit for bit.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libldpc_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC] + os.environ.get("LDPC_EXTRA_FLAGS", "").split()  # development aid: extra -D switches

#: (object name, source, extra flags)
# slowest units first: the pool starts them in this order
UNITS = [(f"decode_spec_spaw_r{k}.o", "decode_spec_spaw.hip", [f"-DLDPC_SPAW_RULE={k}"]) for k in (2, 3, 4)]
UNITS += [(f"decode_spec_spaw_fused_r{k}.o", "decode_spec_spaw.hip", [f"-DLDPC_SPAW_RULE={k}", "-DLDPC_SPEC_FUSED"]) for k in (2, 3, 4)]
UNITS += [("decode_spec_spa.o", "decode_spec_spa.hip", []),
          ("decode_spec_spa_fused.o", "decode_spec_spa.hip", ["-DLDPC_SPEC_FUSED"]),
          ("decode_lnms.o", "decode_lnms.hip", []),
          ("decode_lnms_rt_bg1.o", "decode_lnms_rt.hip", ["-DLDPC_BG_GRAPH=0"]),
          ("decode_lnms_rt_bg2.o", "decode_lnms_rt.hip", ["-DLDPC_BG_GRAPH=1"]),
          ("decode_lnms_fused.o", "decode_lnms.hip", ["-DLDPC_SPEC_FUSED"]),
          ("decode_spec_fused.o", "decode_spec.hip", ["-DLDPC_SPEC_FUSED"]),
          ("decode_spec_r1w_r1.o", "decode_spec_r1w.hip", ["-DLDPC_R1W_RULE=1"]),
          ("decode_spec_r1w_r0.o", "decode_spec_r1w.hip", ["-DLDPC_R1W_RULE=0"]),
          ("decode_spec_r1w_r6.o", "decode_spec_r1w.hip", ["-DLDPC_R1W_RULE=6"]),
          ("decode_spec_r1m.o", "decode_spec_r1m.hip", []),
          ("decode_spec.o", "decode_spec.hip", []),
          ("decode_spec_h.o", "decode_spec_h.hip", []),
          ("decode_lnms_h.o", "decode_lnms_h.hip", []),
          ("decode_lnms_h_fused.o", "decode_lnms_h.hip", ["-DLDPC_SPEC_FUSED"])]
UNITS += [(f"decode_qc_r{k}.o", "decode_qc.hip", [f"-DLDPC_QC_RULE={k}"]) for k in (1, 2, 3, 4, 6)]
UNITS += [("decode_qc_r0.o", "decode_qc_r0.hip", []),
          ("decode_qc_lnms.o", "decode_qc_lnms.hip", []),
          ("decode_csr.o", "decode_csr.hip", []),
          ("ldpc_abi.o", "ldpc_abi.hip", []),
          ("count_errors.o", "count_errors.hip", []),
          ("channel.o", "channel.hip", []),
          ("bitflip.o", "bitflip.hip", []),
          ("encode_qc.o", "encode_qc.hip", []),
          ("rate_match.o", "rate_match.hip", [])]
HEADERS = ["bp_common.h", "extra_kernels.h", "sweep_fused.h", "bp_math.h", "spec_common.h", "spec_r1w_plan.h", "lnms_plan.h", "half2_ops.h", os.path.join("gen", "codes_gen.h"),
           os.path.join(ROOT, "include", "ldpc_amd.h")]


#: source files each kernel family is built from (for stamping counter profiles)
KERNEL_SOURCES = {"spec_lnms_f16x2": ["decode_lnms_h.hip", "lnms_plan.h", "half2_ops.h"], "spec_lnms": ["decode_lnms.hip", "lnms_plan.h"], "bg_lnms": ["decode_lnms_rt.hip", "lnms_plan.h"], "spec_r0w": ["decode_spec_r1w.hip", "spec_r1w_plan.h"],
                  "spec_r1w": ["decode_spec_r1w.hip", "spec_r1w_plan.h"], "spec_spaw": ["decode_spec_spaw.hip", "spec_r1w_plan.h"],
                  "spec_spa": ["decode_spec_spa.hip"], "spec_r0": ["decode_spec.hip"], "spec_r1": ["decode_spec.hip"],
                  "spec_oms": ["decode_spec.hip"], "spec_r1_f16x2": ["decode_spec_h.hip", "half2_ops.h"], "spec_omsw": ["decode_spec_r1w.hip", "spec_r1w_plan.h"],
                  "spec_r1m": ["decode_spec_r1m.hip", "spec_r1w_plan.h"], "spec_omsm": ["decode_spec_r1m.hip", "spec_r1w_plan.h"], "qc_lnms": ["decode_qc_lnms.hip"], "qc_r0": ["decode_qc_r0.hip"],
                  "qc": ["decode_qc.hip"], "csr": ["decode_csr.hip"]}


def kernel_source_digest(kernel: str) -> str:
    """First 16 hex digits of the SHA-256 over the sources the named kernel variant (as
    ``ldpc_last_kernel`` reports it) is built from, plus the shared headers and the generated code
    tables.  Counter profiles record it; a profile whose digest differs is from another kernel."""
    import hashlib

    files = None
    if kernel.startswith("fused_"):  # the same sources compiled with -DLDPC_SPEC_FUSED
        kernel = kernel[len("fused_"):]
    for prefix in sorted(KERNEL_SOURCES, key=len, reverse=True):
        if kernel.startswith(prefix):
            files = KERNEL_SOURCES[prefix]
            break
    if files is None:
        return ""
    h = hashlib.sha256()
    for f in list(files) + ["spec_common.h", "bp_common.h", "bp_math.h", os.path.join("gen", "codes_gen.h")]:
        path = os.path.join(CSRC, f)
        h.update(open(path, "rb").read() if os.path.exists(path) else b"")
    return h.hexdigest()[:16]


def _includes(path: str, seen=None):
    """Files `path` includes with quotes, transitively (csrc/ and include/)."""
    import re
    seen = set() if seen is None else seen
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), flags=re.M):
        for base in (os.path.dirname(path), CSRC, os.path.join(ROOT, "include")):
            cand = os.path.normpath(os.path.join(base, name))
            if os.path.exists(cand):
                if cand not in seen:
                    seen.add(cand)
                    _includes(cand, seen)
                break
    return seen


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, jobs: int = 8, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    from . import gen_codes
    gen_codes.generate()
    headers = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    objs = [os.path.join(OBJ, obj) for obj, _, _ in UNITS]
    sources = sorted({os.path.join(CSRC, src) for _, src, _ in UNITS}) + headers
    if not force and os.path.exists(LIB) and not any(os.path.exists(o) for o in objs) and not _stale(LIB, sources):
        # a tree that travelled with the built library but without the objects (they are not shipped):
        # the library is newer than every source, there is nothing to do
        if verbose:
            print("up to date", LIB, flush=True)
        return LIB
    todo = []
    for obj, src, extra in UNITS:
        o, s = os.path.join(OBJ, obj), os.path.join(CSRC, src)
        if force or _stale(o, [s] + sorted(_includes(s))):
            todo.append((o, [HIPCC] + FLAGS + extra + ["-c", s, "-o", o]))

    def run(item):
        o, cmd = item
        if verbose:
            print("hipcc", os.path.basename(o), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip() and verbose:
            print(r.stderr.strip(), file=sys.stderr)

    with ThreadPoolExecutor(max_workers=max(1, jobs)) as pool:
        list(pool.map(run, todo))
    if force or todo or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose:
            print("linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=8)
    a = ap.parse_args()
    build(a.force, a.jobs)
