R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/gt0; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o ks -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode graph > $OUT/log 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1); python3 $R/tools/timeline.py $f 0 > $OUT/timeline_all.txt
rm -rf $OUT/t
