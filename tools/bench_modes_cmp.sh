R=$GRAFT_REPO_ROOT; cd $R
pr() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['ms_per_step_p50'],4), round(d['ms_per_step_min'],4))"; }
for r in 1 2; do
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | pr default_nocpu
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants 2>/dev/null | pr novariants
STOVE_BENCH_NO_PARITY=1 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants 2>/dev/null | pr noparity
STOVE_BENCH_NO_PARITY=1 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants --profile-steps 0 2>/dev/null | pr noprofile
STOVE_BENCH_NO_PARITY=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 2>/dev/null | pr ab_config
done
