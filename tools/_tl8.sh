R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/tl8; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export STOVE_BENCH_NO_PARITY=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/all -o ks -- python3 $R/bench.py --frames 8 --steps 40 --warmup 3 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode graph > $OUT/all.log 2>&1
python3 $R/tools/timeline.py $(find $OUT/all -name "*kernel_trace.csv" | head -1) 0 > $OUT/timeline_all.txt; rm -rf $OUT/all
