#!/bin/bash
# timeline of the replayed step at the reference's default training shape (256 clips x 8 frames)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/short; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python3 $R/bench.py --frames 8 --steps 50 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 > $OUT/ks.log 2>&1
python3 $R/tools/timeline.py $(find $OUT/ks -name "*kernel_trace.csv" | head -1) 0 > $OUT/timeline.txt; rm -rf $OUT/ks
grep "ms/step" $OUT/ks.log; head -90 $OUT/timeline.txt
