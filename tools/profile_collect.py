#!/usr/bin/env python3
"""Turn the raw output of tools/profile_round.sh or tools/profile_bg2.sh (gpurun_out/prof_<tag>/) into the summaries kept
under profiles/: the kernel-trace statistics of each bench workload (decoder kernels and the ten
largest others) and the HBM traffic of the decode kernel from the FETCH_SIZE / WRITE_SIZE passes.

    python tools/profile_collect.py r01b
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_amd.build import kernel_source_digest  # noqa: E402
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
WORKLOADS = {"ccsds_r0_f64": "ccsds512/R0/f64/f32llr", "ccsds_r0_f64_fused": "ccsds512/R0/f64/fused", "ccsds_r0_f64_fused_packed": "ccsds512/R0/f64/fused_packed",
             "ccsds_r0_f64_ee": "ccsds512/R0/f64/f32llr/early_exit", "bg1_lnms_f32": "nr_bg1_z384/LNMS/f32/f32llr",
             "bg1_lnms_f64": "nr_bg1_z384/LNMS/f64/f32llr", "bg1_lnms_f16": "nr_bg1_z384/LNMS/f16/f32llr",
             "bg1_bglnms_f32": "nr_bg1_z384/LNMS/f32/f32llr/graph", "bg1_bglnms_f64": "nr_bg1_z384/LNMS/f64/f32llr/graph",
             "ccsds_R3_f32": "ccsds512/R3/f32/f32llr", "ccsds_R3_f64": "ccsds512/R3/f64/f32llr", "ccsds_R1_f64": "ccsds512/R1/f64/f32llr", "wifi_R1_f64": "wifi648/R1/f64/f32llr",
             "bg2_R0_f64": "nr_bg2_z40/R0/f64/f32llr", "bg2_R0_f32": "nr_bg2_z40/R0/f32/f32llr",
             "bg2_R1_f64": "nr_bg2_z40/R1/f64/f32llr", "bg2r1400_R1_f64": "nr_bg2_z40_r1400/R1/f64/f32llr", "bg2_R1_f32": "nr_bg2_z40/R1/f32/f32llr",
             "bg2_R3_f32": "nr_bg2_z40/R3/f32/f32llr", "bg2_R3_f64": "nr_bg2_z40/R3/f64/f32llr"}
# only the workloads this tag's run covered
WORKLOADS = {wl: key for wl, key in WORKLOADS.items()
             if os.path.isdir(os.path.join(src, f"stats_{wl}")) or os.path.isdir(os.path.join(src, f"fetch_{wl}"))}


def decoder_kernel(name):
    return "ldpc::" in name and not any(k in name for k in ("count_errors", "channel_llr", "encode_qc", "rate_"))


for wl in WORKLOADS:
    if not os.path.isdir(os.path.join(src, f"stats_{wl}")):
        continue
    rows = list(csv.DictReader(open(os.path.join(src, f"stats_{wl}", "t_kernel_stats.csv"))))
    keep = [r for r in rows if decoder_kernel(r["Name"])] + [r for r in rows if not decoder_kernel(r["Name"])][:10]
    out = os.path.join(dst, f"{tag}_{wl}_kernel_stats.csv")
    with open(out, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in keep:
            r = dict(r)
            if len(r["Name"]) > 200:
                r["Name"] = r["Name"][:200] + "..."
            w.writerow(r)
    print("wrote", out)

traffic = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python bench.py --steps 2 "
                   "--warmup 1 --no-other --no-cpu [workload flags]`; values in KiB per dispatch of the decode kernel "
                   "(mean over its dispatches).  On gfx950 FETCH_SIZE reports half of the bytes actually fetched "
                   "(128-byte requests tallied as 64), so the read side is doubled; WRITE_SIZE is taken as is.",
           "workloads": {}}
for wl, key in WORKLOADS.items():
    if not os.path.isdir(os.path.join(src, f"fetch_{wl}")):
        continue
    vals = {}
    for ctr, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        rows = list(csv.DictReader(open(os.path.join(src, f"{d}_{wl}", "t_counter_collection.csv"))))
        per = {}
        for r in rows:
            if decoder_kernel(r["Kernel_Name"]) and r["Counter_Name"] == ctr:
                per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        vals[ctr] = sum(per.values()) / len(per)
    bench = json.load(open(os.path.join(src, f"fetch_{wl}.json")))
    frames = bench["config"]["frames_per_launch_per_gpu"]
    total = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
    traffic["workloads"][key] = {
        "frames": frames, "FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
        "hbm_bytes_corrected": total, "hbm_bytes_per_frame": total / frames,
        "algorithmic_bytes_per_frame": bench["roofline"]["bytes_per_frame"],
        "early_exit": bench["config"]["early_exit"], "kernel": bench["config"]["kernel"],
        "kernel_source_sha16": kernel_source_digest(bench["config"]["kernel"])}
# instruction counters of the decode kernel (one more pass), per dispatch
sq = {"note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES "
              "SQ_BUSY_CU_CYCLES SQ_WAIT_ANY over the same bench command; sums over the device per dispatch of the "
              "decode kernel (mean over its dispatches); SQ_WAVE_CYCLES and SQ_WAIT_ANY count in units of 4 cycles.",
      "workloads": {}}
for wl, key in WORKLOADS.items():
    path = os.path.join(src, f"sq_{wl}", "t_counter_collection.csv")
    if not os.path.exists(path):
        continue
    per = {}
    for r in csv.DictReader(open(path)):
        if decoder_kernel(r["Kernel_Name"]):
            d = per.setdefault(r["Dispatch_Id"], {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    mean = {c: sum(d[c] for d in per.values()) / len(per) for c in next(iter(per.values()))}
    bench = json.load(open(os.path.join(src, f"sq_{wl}.json")))
    frames = bench["config"]["frames_per_launch_per_gpu"]
    executed = bench["counters"]["iterations"] + (0 if not bench["config"]["early_exit"] else bench["counters"]["not_converged"])
    mean.update({"frames": frames, "executed_frame_iterations": executed, "kernel": bench["config"]["kernel"],
                 "early_exit": bench["config"]["early_exit"],
                 "kernel_source_sha16": kernel_source_digest(bench["config"]["kernel"]),
                 "valu_wave_instructions_per_frame_iteration": mean["SQ_INSTS_VALU"] / executed})
    sq["workloads"][key] = mean
# LDS activity of the decode kernel (one more pass), per dispatch
lds = {"note": "rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS_LOAD "
               "SQ_INSTS_LDS_STORE SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS over the same bench command; sums over the device per "
               "dispatch of the decode kernel (mean over its dispatches).  SQ_LDS_IDX_ACTIVE = LDS-array cycles (bank "
               "conflicts included), SQ_BUSY_CU_CYCLES = cycles a compute unit had waves: their ratio is the share of the "
               "time the LDS arrays were busy.",
       "workloads": {}}
for wl, key in WORKLOADS.items():
    path = os.path.join(src, f"lds_{wl}", "t_counter_collection.csv")
    if not os.path.exists(path):
        continue
    per = {}
    for r in csv.DictReader(open(path)):
        if decoder_kernel(r["Kernel_Name"]):
            d = per.setdefault(r["Dispatch_Id"], {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    mean = {c: sum(d[c] for d in per.values()) / len(per) for c in next(iter(per.values()))}
    bench = json.load(open(os.path.join(src, f"lds_{wl}.json")))
    frames = bench["config"]["frames_per_launch_per_gpu"]
    executed = bench["counters"]["iterations"] + (0 if not bench["config"]["early_exit"] else bench["counters"]["not_converged"])
    mean.update({"frames": frames, "executed_frame_iterations": executed, "kernel": bench["config"]["kernel"],
                 "early_exit": bench["config"]["early_exit"], "kernel_ms": bench["roofline"]["kernel_ms"],
                 "kernel_source_sha16": kernel_source_digest(bench["config"]["kernel"]),
                 "lds_array_busy_frac": mean["SQ_LDS_IDX_ACTIVE"] / mean["SQ_BUSY_CU_CYCLES"],
                 "lds_array_busy_frac_of_kernel_time": mean["SQ_LDS_IDX_ACTIVE"] / (bench["roofline"]["kernel_ms"] * 1e-3 * 2.4e9 * 256),
                 "lds_array_cycles_per_frame_iteration": mean["SQ_LDS_IDX_ACTIVE"] / executed,
                 "lds_instructions_per_frame_iteration": mean["SQ_INSTS_LDS"] / executed})
    lds["workloads"][key] = mean
if lds["workloads"]:
    out = os.path.join(dst, f"{tag}_lds_counters.json")
    json.dump(lds, open(out, "w"), indent=1)
    print("wrote", out)
    print(json.dumps(lds["workloads"], indent=1))
if sq["workloads"]:
    out = os.path.join(dst, f"{tag}_sq_counters.json")
    json.dump(sq, open(out, "w"), indent=1)
    print("wrote", out)
    print(json.dumps(sq["workloads"], indent=1))

out = os.path.join(dst, f"{tag}_hbm_traffic.json")
json.dump(traffic, open(out, "w"), indent=1)
print("wrote", out)
print(json.dumps(traffic["workloads"], indent=1))
