#!/bin/bash
# Instruction and wait counters of the layered kernel on BG1 Z=384: float32 against packed half precision.
#   gpurun -- 'bash tools/profile_lnms_half.sh TAG'
set -e -o pipefail
TAG=${1:-r02f}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
ARGS="--steps 2 --warmup 1 --no-other --no-cpu --code nr_bg1_z384 --rule LNMS --frames 16384"
run_pmc() {  # name, counters, dtype
    timeout -s KILL 200 rocprofv3 --pmc $2 -d "$OUT/$1" -o t --output-format csv -- \
        python3 "$B" $ARGS --dtype $3 > "$OUT/$1.json" 2> "$OUT/$1.err"
}
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY"
run_pmc sq_bg1_lnms_f32 "$SQ" f32
run_pmc sq_bg1_lnms_f16 "$SQ" f16
echo "first pass done"
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY"
run_pmc sq2_bg1_lnms_f32 "$SQ2" f32
run_pmc sq2_bg1_lnms_f16 "$SQ2" f16
echo "second pass done"
