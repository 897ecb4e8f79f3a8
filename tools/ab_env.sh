#!/bin/bash
# A/B of an environment switch on the default bench inside ONE gpurun call (boxes differ by ~1 %): VAR unset vs VAR=1, 3 rounds
V=$1
for i in 1 2 3; do
  echo -n "unset   "; python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), round(d["ms_per_step_p50"],4))'
  echo -n "$V=1  "; env $V=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), round(d["ms_per_step_p50"],4))'
done
