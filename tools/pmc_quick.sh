#!/bin/bash
# One counter pass over a bench workload, decode kernel only, mean per dispatch:
#   gpurun -- 'bash tools/pmc_quick.sh NAME "SQ_INSTS_VALU SQ_INSTS_SALU ..." [bench.py flags]'
# (at most 8 SQ_ counters per pass; no tracing options beside --pmc)
set -e -o pipefail
NAME=$1; CTR=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$NAME
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc $CTR -d "$OUT" -o t --output-format csv -- \
    python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-other --no-cpu "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 - "$OUT" <<'PY'
import csv, json, sys, os
out = sys.argv[1]
per = {}
for r in csv.DictReader(open(os.path.join(out, "t_counter_collection.csv"))):
    k = r["Kernel_Name"]
    if "ldpc::" in k and not any(x in k for x in ("count_errors", "channel_llr", "encode_qc", "rate_")):
        d = per.setdefault(r["Dispatch_Id"], {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
mean = {c: sum(d[c] for d in per.values()) / len(per) for c in next(iter(per.values()))}
b = json.load(open(os.path.join(out, "bench.json")))
mean["kernel"] = b["config"]["kernel"]; mean["kernel_ms"] = b["roofline"]["kernel_ms"]; mean["dispatches"] = len(per)
json.dump(mean, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(mean))
PY
