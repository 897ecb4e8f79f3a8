#!/bin/bash
# The other BASELINE.json configurations through the same bench step (parity-test cases, not bench lines), plus the
# per-kernel isolated / overlapped times of the headline step.  Usage: gpurun -- 'bash tools/workloads_bench.sh r02'
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for w in gravity avoidance multibilliards; do
  timeout 400 python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $OUT/bench_$w.json 2> $OUT/bench_$w.log
  tail -c 300 $OUT/bench_$w.json | head -c 0
done
bash tools/kernel_times.sh > $OUT/kernel_times.txt 2>&1
python3 - <<PY
import json
for w in ('gravity', 'avoidance', 'multibilliards'):
    try:
        d = json.loads(open('$OUT/bench_%s.json' % w).read().strip().splitlines()[-1])
        print(w, round(d['ms_per_step'], 3), 'ms', round(d['value'] / 1e6, 3), 'M frames/s', d['config']['workload'])
    except Exception as e:
        print(w, 'failed', e)
PY
