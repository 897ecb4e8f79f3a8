#!/bin/bash
# A/B of one environment variable over a list of values on one box, alternating runs:
#   VAR=STOVE_X VALS="1 2 4" bash tools/ab_vals.sh [bench args]
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for v in $VALS; do
  env $VAR=$v STOVE_BENCH_NO_PARITY=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v $*', round(d['ms_per_step'],4), round(d['ms_per_step_p50'],4))"
done; done
