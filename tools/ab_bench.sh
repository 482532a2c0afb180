# development aid: bench lines with the built library and with the one in ldpc_amd/alt
#   bash tools/ab_bench.sh "--code X --rule R --dtype f64 ..." ...
for a in "$@"; do for lib in "" alt "" alt; do
LDPC_AMD_LIB=${lib:+$GRAFT_REPO_ROOT/ldpc_amd/alt/libldpc_amd.so} timeout -k 10 200 python3 bench.py $a --no-other --no-cpu --steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"$lib|$a|\", round(d[\"value\"],3), round(d[\"ms_per_step\"],4), d[\"config\"][\"kernel\"], d[\"mean_iters\"])" || exit 1; done; done
