#!/bin/bash
# Profiles of the multi-wave flooding kernels on the BG2-shaped 5G matrix: kernel-trace statistics,
# HBM traffic and SQ counters.   gpurun -- 'bash tools/profile_bg2.sh r01d'
set -e -o pipefail
TAG=${1:-r01d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
ARGS="--code nr_bg2_z40 --frames 65536 --fixed --snr -2.0 --no-other --no-cpu"
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY"
for cfg in "R0 f64" "R0 f32" "R1 f32" "R3 f32"; do
    set -- $cfg
    name=bg2_$1_$2
    timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/stats_$name" -o t --output-format csv -- \
        python3 "$B" --steps 5 --warmup 2 $ARGS --rule $1 --dtype $2 > "$OUT/stats_$name.json" 2> "$OUT/stats_$name.err"
    echo "stats $name done"
done
for pass in "FETCH_SIZE fetch" "WRITE_SIZE write"; do
    set -- $pass
    timeout -s KILL 200 rocprofv3 --pmc $1 -d "$OUT/$2_bg2_R0_f32" -o t --output-format csv -- \
        python3 "$B" --steps 2 --warmup 1 $ARGS --rule R0 --dtype f32 > "$OUT/$2_bg2_R0_f32.json" 2> "$OUT/$2_bg2_R0_f32.err"
done
timeout -s KILL 200 rocprofv3 --pmc $SQ -d "$OUT/sq_bg2_R0_f32" -o t --output-format csv -- \
    python3 "$B" --steps 2 --warmup 1 $ARGS --rule R0 --dtype f32 > "$OUT/sq_bg2_R0_f32.json" 2> "$OUT/sq_bg2_R0_f32.err"
echo "pmc done"
ls "$OUT"
