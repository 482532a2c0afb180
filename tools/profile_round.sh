#!/bin/bash
# Profiles of the bench workloads on the GPU box: kernel-trace statistics and HBM traffic counters.
#   gpurun -- 'bash tools/profile_round.sh r01'
# Raw output goes to gpurun_out/prof_<tag>/; tools/profile_collect.py turns it into profiles/.
set -e -o pipefail
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
run_stats() {  # name, bench args...
    local name=$1; shift
    timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/$name" -o t --output-format csv -- \
        python3 "$B" --steps 5 --warmup 2 --no-other --no-cpu "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
}
run_pmc() {  # name, counter, bench args...
    local name=$1 ctr=$2; shift 2
    timeout -s KILL 200 rocprofv3 --pmc $ctr -d "$OUT/$name" -o t --output-format csv -- \
        python3 "$B" --steps 2 --warmup 1 --no-other --no-cpu "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
}
run_stats stats_ccsds_r0_f64
echo "stats ccsds done"
run_stats stats_ccsds_r0_f64_fused --fused --early-exit
run_stats stats_bg1_lnms_f32 --code nr_bg1_z384 --rule LNMS --dtype f32 --frames 16384
run_stats stats_bg1_lnms_f64 --code nr_bg1_z384 --rule LNMS --dtype f64 --frames 16384
run_stats stats_bg1_lnms_f16 --code nr_bg1_z384 --rule LNMS --dtype f16 --frames 16384
echo "stats bg1 done"
run_pmc fetch_ccsds_r0_f64 FETCH_SIZE
run_pmc write_ccsds_r0_f64 WRITE_SIZE
run_pmc fetch_ccsds_r0_f64_fused FETCH_SIZE --fused --early-exit
run_pmc write_ccsds_r0_f64_fused WRITE_SIZE --fused --early-exit
run_pmc fetch_ccsds_r0_f64_ee FETCH_SIZE --early-exit
run_pmc write_ccsds_r0_f64_ee WRITE_SIZE --early-exit
echo "pmc ccsds done"
run_pmc fetch_bg1_lnms_f32 FETCH_SIZE --code nr_bg1_z384 --rule LNMS --dtype f32 --frames 16384
run_pmc write_bg1_lnms_f32 WRITE_SIZE --code nr_bg1_z384 --rule LNMS --dtype f32 --frames 16384
run_pmc fetch_bg1_lnms_f64 FETCH_SIZE --code nr_bg1_z384 --rule LNMS --dtype f64 --frames 16384
run_pmc write_bg1_lnms_f64 WRITE_SIZE --code nr_bg1_z384 --rule LNMS --dtype f64 --frames 16384
run_pmc fetch_bg1_lnms_f16 FETCH_SIZE --code nr_bg1_z384 --rule LNMS --dtype f16 --frames 16384
run_pmc write_bg1_lnms_f16 WRITE_SIZE --code nr_bg1_z384 --rule LNMS --dtype f16 --frames 16384
echo "pmc bg1 done"
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY"
run_pmc sq_ccsds_r0_f64 "$SQ"
run_pmc sq_bg1_lnms_f32 "$SQ" --code nr_bg1_z384 --rule LNMS --dtype f32 --frames 16384
run_pmc sq_bg1_lnms_f64 "$SQ" --code nr_bg1_z384 --rule LNMS --dtype f64 --frames 16384
run_pmc sq_bg1_lnms_f16 "$SQ" --code nr_bg1_z384 --rule LNMS --dtype f16 --frames 16384
echo "sq counters done"
ls "$OUT"
