#!/bin/bash
# Kernel timelines of the median step: MODES="graph eager" PIECES="1 2"
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/gt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in ${MODES:-graph}; do for p in ${PIECES:-2}; do
  STOVE_PIECES=$p timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$m$p -o ks -- python3 $R/bench.py --workload ${WORKLOAD:-billiards} --steps 20 --warmup 3 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode $m > $OUT/$m$p.log 2>&1
  f=$(find $OUT/$m$p -name "*kernel_trace.csv" | head -1); python3 $R/tools/timeline.py $f 10 > $OUT/timeline_$m$p.txt
  rm -rf $OUT/$m$p
done; done
grep "ms/step" $OUT/*.log
