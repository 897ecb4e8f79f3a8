#!/bin/bash
# A/B of one environment switch on one box, alternating runs: VAR=STOVE_X bash tools/ab_env.sh [bench args]   (values 0 and 1)
R=$GRAFT_REPO_ROOT; cd $R
for v in 0 1 0 1; do
  env $VAR=$v STOVE_BENCH_NO_PARITY=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['ms_per_step'],4), round(d['ms_per_step_p50'],4))"
done
