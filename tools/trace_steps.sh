#!/bin/bash
# rocprofv3 kernel trace of 20 bench steps -> per-kernel stats + the median step's timeline (tools/timeline.py).
# Usage: gpurun -- 'bash tools/trace_steps.sh r02a [bench args]'
TAG=${1:-trace}
shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants --profile-steps 0 "$@" > $OUT/ks.log 2>&1
cd $R
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f 20 > $OUT/timeline.txt
python3 tools/timeline.py $f 0 > $OUT/timeline_all.txt
s=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && cp $s $OUT/kernel_stats.csv
rm -rf $OUT/ks      # the raw trace is tens of MB
cat $OUT/timeline.txt
