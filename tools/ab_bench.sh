#!/bin/bash
# A/B of an environment switch on the default bench: bash tools/ab_bench.sh VAR  (runs VAR=0 and VAR=1 twice, interleaved)
V=$1
for i in 1 2; do
  for x in 0 1; do
    echo -n "$V=$x  "; env $V=$x python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
  done
done
