#!/bin/bash
# A/B at the reference's default training shape (256 clips x 8 frames) and at the headline shape
for rep in 1 2; do for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  for fr in 8 100; do
    out=$(env $envs python3 bench.py --frames $fr --steps 60 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 2>/dev/null | tail -1)
    python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('%-16s T=%-3s p50 %.3f  mean %.3f  min %.3f' % (sys.argv[2], sys.argv[3], d['ms_per_step_p50'], d['ms_per_step'], d['ms_per_step_min']))" "$out" "$label" $fr
  done
done; done
