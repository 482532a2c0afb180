# quick rates of the single-wave flooding kernels on CCSDS (development aid)
set -e
for args in "--rule R0 --dtype f64" "--rule R0 --dtype f32" "--rule R1 --dtype f64" "--rule R1 --dtype f32" "--rule R1 --dtype f64 --early-exit" "--rule OMS --dtype f32"; do
 timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu --no-other $args 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$args',round(d['value'],2),round(d['ms_per_step'],3),d['config']['kernel'])"
done
for args in "--rule R1 --dtype f16" "--rule R1 --dtype f16 --early-exit"; do
 timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu --no-other $args 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$args',round(d['value'],2),round(d['ms_per_step'],3),d['config']['kernel'])"
done
