#!/bin/bash
# A/B of step variants on one box: prints p50 / mean ms per step.  Usage: tools/ab.sh "label:ENV=.. ENV=.." ...
B="python3 bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0"
for rep in 1 2; do for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  out=$(env $envs $B 2>/dev/null | tail -1)
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('%-28s p50 %.3f  mean %.3f  min %.3f' % (sys.argv[2], d['ms_per_step_p50'], d['ms_per_step'], d['ms_per_step_min']))" "$out" "$label"
done; done
