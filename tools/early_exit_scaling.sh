set -e
for fr in 262144 1048576 4194304; do
for args in "--rule R1 --dtype f64 --early-exit" "--rule R0 --dtype f64 --early-exit" "--rule R1 --dtype f16 --early-exit"; do
 timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu --no-other --frames $fr $args 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$fr $args',round(d['value'],2),round(d['ms_per_step'],3),d['mean_iters'])"
done; done
