#!/bin/bash
# Per-kernel launch times of the default bench step, overlapped (as shipped) and with the side streams serialised
# (STOVE_NO_OVERLAP=1: every kernel alone on the device).  bash tools/kernel_times.sh [extra bench args]
for x in 0 1; do
  echo "== STOVE_NO_OVERLAP=$x"
  STOVE_NO_OVERLAP=$x timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants "$@" 2>/dev/null | python3 -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("ms_per_step", round(d["ms_per_step"], 4), "p50", round(d["ms_per_step_p50"], 4))
for k, v in d["roofline"]["kernels_ms_per_step"].items(): print("  %-28s %.4f" % (k, v))
'
done
