#!/bin/bash
# several values of one environment switch on one box, three alternating rounds: VAR=STOVE_X VALS="0 1 2" bash tools/ab_multi.sh [bench args]
R=$GRAFT_REPO_ROOT; cd $R
for r in 1 2 3; do for v in $VALS; do
  env $VAR=$v STOVE_BENCH_NO_PARITY=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['ms_per_step'],4), round(d['ms_per_step_p50'],4))"
done; done
