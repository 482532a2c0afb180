#!/bin/bash
# Counter passes of the BG1 float32 layered kernel only (development aid).
#   gpurun -- 'bash tools/profile_lnms_quick.sh tag'
set -e -o pipefail
TAG=${1:-q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lnmsq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
ARGS="--steps 2 --warmup 1 --no-other --no-cpu --code nr_bg1_z384 --rule LNMS --dtype ${2:-f32} --frames 16384"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d "$OUT/sq1" -o t --output-format csv -- python3 "$B" $ARGS > "$OUT/sq1.json" 2> "$OUT/sq1.err"
timeout -s KILL 120 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM -d "$OUT/sq2" -o t --output-format csv -- python3 "$B" $ARGS > "$OUT/sq2.json" 2> "$OUT/sq2.err"
python3 - "$OUT" <<'P'
import csv,sys,glob,collections
out=sys.argv[1]
for d in ("sq1","sq2"):
    acc=collections.defaultdict(float); n=collections.Counter()
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "lnms" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    for k in sorted(acc): print(d,k,acc[k]/max(1,n[k]/ (n[k] and 1)), n[k])
P
