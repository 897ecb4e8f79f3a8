#!/bin/bash
# Every kernel of the median replayed step (no duration cut-off): gpurun -- 'bash tools/full_timeline.sh [tag] [bench args]'
TAG=${1:-full}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export STOVE_BENCH_NO_PARITY=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode graph "$@" > $OUT/ks.log 2>&1
f=$(find $OUT/ks -name "*kernel_trace.csv" | head -1); python3 $R/tools/timeline.py $f 0 > $OUT/timeline_all.txt
rm -rf $OUT/ks
