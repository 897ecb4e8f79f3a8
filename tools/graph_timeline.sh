#!/bin/bash
# Kernel timelines of the median step, eager vs captured-graph replay.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/gt; mkdir -p $OUT
cd $R
for m in eager graph; do
  timeout 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode $m > $OUT/bench_$m.json 2> $OUT/bench_$m.err
  grep "ms/step" $OUT/bench_$m.err
done
STOVE_GRAPH_ONE=1 timeout 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode graph 2>&1 | grep "ms/step"
cd /tmp && export TMPDIR=/tmp
for m in ${MODES:-graph}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$m -o ks -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode $m > $OUT/$m.log 2>&1
  f=$(find $OUT/$m -name "*kernel_trace.csv" | head -1); python3 $R/tools/timeline.py $f 10 > $OUT/timeline_$m.txt
  rm -rf $OUT/$m
done
grep "ms/step" $OUT/*.log
