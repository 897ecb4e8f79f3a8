#!/bin/bash
# one eager bench with the stall watcher on this (fresh) box; keeps the log only if a step stalled
mkdir -p gpurun_out/stalls
f=gpurun_out/stalls/eager_$(date +%s).err
STOVE_BENCH_WATCH=1 STOVE_BENCH_WATCH_MS=5 STOVE_BENCH_GCLOG=1 timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --step-mode eager --no-cpu-baseline --no-variants --profile-steps 0 2> $f > /dev/null
if grep -q "stalled step" $f; then echo "STALL CAPTURED in $f"; grep "device ms\|host ms" $f | cut -c1-250; else rm -f $f; echo "no stall on this box"; fi
