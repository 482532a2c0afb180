#!/bin/bash
# Profiles of the bench workloads on the GPU box: kernel-trace statistics, HBM traffic counters, and two
# SQ counter passes (instruction mix / waiting, and LDS activity).
#   gpurun -- 'bash tools/profile_round.sh r03 [workload ...]'
# Raw output goes to gpurun_out/prof_<tag>/; tools/profile_collect.py turns it into profiles/.
# Every rocprofv3 run has the program itself after "--", one counter group per run, no trace options beside --pmc.
set -e -o pipefail
TAG=${1:-r03}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
declare -A W
W[ccsds_r0_f64]=""
W[ccsds_r0_f64_fused]="--fused --early-exit"
W[ccsds_r0_f64_fused_packed]="--fused --packed-ref --early-exit"
W[ccsds_r0_f64_ee]="--early-exit"
W[bg1_lnms_f32]="--code nr_bg1_z384 --rule LNMS --dtype f32 --frames 16384"
W[bg1_lnms_f64]="--code nr_bg1_z384 --rule LNMS --dtype f64 --frames 16384"
W[bg1_lnms_f16]="--code nr_bg1_z384 --rule LNMS --dtype f16 --frames 16384"
W[bg1_bglnms_f32]="--code nr_bg1_z384 --rule LNMS --dtype f32 --frames 16384 --kernel 3"
W[bg1_bglnms_f64]="--code nr_bg1_z384 --rule LNMS --dtype f64 --frames 16384 --kernel 3"
W[ccsds_R3_f32]="--rule R3 --dtype f32 --frames 131072"
W[ccsds_R3_f64]="--rule R3 --dtype f64 --frames 65536"
W[ccsds_R1_f64]="--rule R1 --dtype f64"
W[bg2_R0_f64]="--code nr_bg2_z40 --frames 32768 --snr -2.0 --rule R0 --dtype f64"
W[bg2_R0_f32]="--code nr_bg2_z40 --frames 32768 --snr -2.0 --rule R0 --dtype f32"
W[bg2_R1_f64]="--code nr_bg2_z40 --frames 32768 --snr -2.0 --rule R1 --dtype f64"
W[bg2r1400_R1_f64]="--code nr_bg2_z40_r1400 --frames 32768 --snr -2.0 --rule R1 --dtype f64"
W[wifi_R1_f64]="--code wifi648 --frames 65536 --rule R1 --dtype f64"
W[bg2_R1_f32]="--code nr_bg2_z40 --frames 32768 --snr -2.0 --rule R1 --dtype f32"
W[bg2_R3_f32]="--code nr_bg2_z40 --frames 32768 --snr -2.0 --rule R3 --dtype f32"
W[bg2_R3_f64]="--code nr_bg2_z40 --frames 8192 --snr -2.0 --rule R3 --dtype f64"
LIST=${@:-ccsds_r0_f64 ccsds_r0_f64_fused ccsds_r0_f64_fused_packed ccsds_r0_f64_ee ccsds_R1_f64 wifi_R1_f64 bg1_lnms_f32 bg1_lnms_f64 bg1_lnms_f16 bg1_bglnms_f32 ccsds_R3_f32 ccsds_R3_f64 bg2_R0_f64 bg2_R1_f64 bg2r1400_R1_f64 bg2_R1_f32 bg2_R3_f32}
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY"
LDS="SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS"
for name in $LIST; do
    args=${W[$name]}
    timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/stats_$name" -o t --output-format csv -- \
        python3 "$B" --steps 5 --warmup 2 --no-other --no-cpu $args > "$OUT/stats_$name.json" 2> "$OUT/stats_$name.err"
    for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:$SQ" "lds:$LDS"; do
        timeout -s KILL 200 rocprofv3 --pmc ${pass#*:} -d "$OUT/${pass%%:*}_$name" -o t --output-format csv -- \
            python3 "$B" --steps 2 --warmup 1 --no-other --no-cpu $args > "$OUT/${pass%%:*}_$name.json" 2> "$OUT/${pass%%:*}_$name.err"
    done
    echo "$name done"
done
ls "$OUT" | wc -l
