#!/usr/bin/env python3
"""Markdown table of the profiled kernels: one row per workload of a profile round, from the counter
files under profiles/ and the bench line of the same kernels.

    python tools/design_table.py r03b [profiles/r03b_bench.json]

Columns
  Gbit/s        information bits per second of the unprofiled bench run (bench JSON), or, marked "p", from
                the kernel time under the profiler
  HBM           measured bytes per frame (FETCH_SIZE x 2 + WRITE_SIZE) / algorithmic bytes per frame, and
                the measured bytes over the kernel time as a share of 8 TB/s
  LDS busy      SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES: share of the compute units' busy cycles in which the
                LDS arrays worked; in brackets the share of those cycles that were bank conflicts
  VALU          SQ_INSTS_VALU per frame-iteration, and SQ_INSTS_VALU / SQ_BUSY_CU_CYCLES: wave-instructions
                per compute-unit cycle, 1.0 = every SIMD issuing one every fourth cycle
  waiting       SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of the wave-cycles spent waiting on a counter
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# workload key of the counter files -> where the same kernel's rate sits in the bench line
BENCH_PATH = {
    "ccsds512/R0/f64/f32llr": ("value",),
    "ccsds512/R0/f64/f32llr/early_exit": ("operating_point", "value"),
    "ccsds512/R0/f64/fused": ("other", "sweep_R0_f64_es_n0_1dB", "fused_gbit_s"),
    "ccsds512/R0/f64/fused_packed": None,
    "nr_bg1_z384/LNMS/f32/f32llr": ("bg1", "f32", "value"),
    "nr_bg1_z384/LNMS/f64/f32llr": ("bg1", "f64", "value"),
    "nr_bg1_z384/LNMS/f16/f32llr": ("bg1", "f16x2", "value"),
    "nr_bg1_z384/LNMS/f32/f32llr/graph": ("bg1", "f32_run_time_shifts", "value"),
    "nr_bg1_z384/LNMS/f64/f32llr/graph": None,
    "ccsds512/R3/f32/f32llr": ("other", "R3_f32_fixed_10", "gbit_s"),
    "ccsds512/R3/f64/f32llr": ("other", "R3_f64_fixed_10", "gbit_s"),
    "ccsds512/R1/f64/f32llr": ("other", "R1_f64_fixed_10", "gbit_s"),
    "wifi648/R1/f64/f32llr": ("other", "WIFI648_R1_f64_fixed_10", "gbit_s"),
    "nr_bg2_z40/R0/f64/f32llr": ("other", "BG2_Z40_R0_f64_fixed_10", "gbit_s"),
    "nr_bg2_z40/R0/f32/f32llr": ("other", "BG2_Z40_R0_f32_fixed_10", "gbit_s"),
    "nr_bg2_z40/R1/f64/f32llr": ("other", "BG2_Z40_R1_f64_fixed_10", "gbit_s"),
    "nr_bg2_z40/R1/f32/f32llr": ("other", "BG2_Z40_R1_f32_fixed_10", "gbit_s"),
    "nr_bg2_z40_r1400/R1/f64/f32llr": ("other", "BG2_Z40_rows1400_R1_f64_fixed_10", "gbit_s"),
    "nr_bg2_z40/R3/f32/f32llr": ("other", "BG2_Z40_R3_f32_fixed_10", "gbit_s"),
}
INFO_BITS = {"ccsds512": 256, "wifi648": 324, "nr_bg2_z40": 400, "nr_bg2_z40_r1400": 680, "nr_bg1_z384": 8448}


def dig(d, path):
    for p in path:
        if not isinstance(d, dict) or p not in d:
            return None
        d = d[p]
    return d


def main():
    tag = sys.argv[1]
    bench = None
    bpath = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", f"{tag}_bench.json")
    if os.path.exists(bpath):
        bench = json.loads(open(bpath).read().strip().splitlines()[-1])
    load = lambda kind: json.load(open(os.path.join(ROOT, "profiles", f"{tag}_{kind}.json")))["workloads"]
    hbm, sq, lds = load("hbm_traffic"), load("sq_counters"), load("lds_counters")
    print("| workload | kernel | Gbit/s | HBM B/frame measured / algorithmic (share of 8 TB/s) | LDS busy (conflicts) | VALU per frame-iteration (per CU-cycle) | waiting |")
    print("|---|---|---|---|---|---|---|")
    for key, s in sq.items():
        code, rule, dtype = key.split("/")[:3]
        mode = "early exit" if s["early_exit"] else "10 fixed"
        if key.endswith("/fused"):
            mode = "fused sweep, early exit"
        if key.endswith("/fused_packed"):
            mode = "fused sweep, packed reference words, early exit"
        l, h = lds.get(key), hbm.get(key)
        ms = l["kernel_ms"] if l else None
        rate = dig(bench, BENCH_PATH[key]) if bench and BENCH_PATH.get(key) else None
        if rate is not None:
            rate_s = f"{rate:.2f}"
        elif ms:
            rate_s = f"{s['frames'] * INFO_BITS[code] / (ms * 1e-3) / 1e9:.2f} p"
        else:
            rate_s = "-"
        if h:
            share = h["hbm_bytes_corrected"] / (ms * 1e-3) / 8e12 if ms else float("nan")
            hbm_s = f"{h['hbm_bytes_per_frame']:,.0f} / {h['algorithmic_bytes_per_frame']:,} ({share:.3f})"
        else:
            hbm_s = "-"
        if l:
            lds_s = f"{l['lds_array_busy_frac']:.2f} ({l['SQ_LDS_BANK_CONFLICT'] / max(l['SQ_LDS_IDX_ACTIVE'], 1):.0%})"
        else:
            lds_s = "-"
        valu_s = f"{s['valu_wave_instructions_per_frame_iteration']:,.0f} ({s['SQ_INSTS_VALU'] / s['SQ_BUSY_CU_CYCLES']:.2f})"
        wait_s = f"{s['SQ_WAIT_ANY'] / s['SQ_WAVE_CYCLES']:.2f}"
        print(f"| {code} {rule} {dtype}, {mode} | `{s['kernel']}` | {rate_s} | {hbm_s} | {lds_s} | {valu_s} | {wait_s} |")


if __name__ == "__main__":
    main()
