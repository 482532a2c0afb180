# quick rates of the single-wave sum-product kernels on CCSDS (development aid)
set -e
for args in "--rule R3 --dtype f32 --frames 262144" "--rule R3 --dtype f64 --frames 65536" "--rule R2 --dtype f32 --frames 262144" "--rule R4 --dtype f32 --frames 262144" "--rule R3 --dtype f32 --frames 262144 --early-exit"; do
 timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu --no-other $args 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$args',round(d['value'],2),round(d['ms_per_step'],3),d['config']['kernel'])"
done
