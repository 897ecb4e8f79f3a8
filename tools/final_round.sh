#!/bin/bash
# Everything the round's profiles/ are made from, on ONE box.  Usage: gpurun --timeout 2400 -- 'bash tools/final_round.sh r03'
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
tools/stall_watch.sh
bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1
# timeline of the step enqueued eagerly (the default three-graph replay is profile_round's timeline.txt)
cd /tmp && export TMPDIR=/tmp
export STOVE_BENCH_NO_PARITY=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eag -o ks -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode eager > $OUT/eag.log 2>&1
python3 $R/tools/timeline.py $(find $OUT/eag -name "*kernel_trace.csv" | head -1) 10 > $OUT/timeline_eager.txt; rm -rf $OUT/eag
cd $R
for fs in bw32 u8; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 --frame-store $fs 2>/dev/null | tail -1 > $OUT/bench_store_$fs.json
done
bash tools/workloads_bench.sh $TAG > $OUT/workloads.log 2>&1
python3 - <<PY
import json
for n in ('bench', 'bench_store_bw32', 'bench_store_u8', 'bench_gravity', 'bench_avoidance', 'bench_multibilliards'):
    try:
        d = json.loads(open('$OUT/%s.json' % n).read().strip().splitlines()[-1])
        print('%-24s %.3f ms/step  p50 %.3f  p99 %.3f  %.3f M frames/s' % (n, d['ms_per_step'], d['ms_per_step_p50'], d['ms_per_step_p99'], d['value'] / 1e6))
    except Exception as e:
        print(n, 'failed', e)
PY
# every kernel of the median step (headline and six objects), the recursion's in-kernel stamps
cd /tmp && export TMPDIR=/tmp
export STOVE_BENCH_NO_PARITY=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/all -o ks -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode graph > $OUT/all.log 2>&1
python3 $R/tools/timeline.py $(find $OUT/all -name "*kernel_trace.csv" | head -1) 0 > $OUT/timeline_all.txt; rm -rf $OUT/all
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mb -o ks -- python3 $R/bench.py --workload multibilliards --steps 20 --warmup 3 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode graph > $OUT/mb.log 2>&1
python3 $R/tools/timeline.py $(find $OUT/mb -name "*kernel_trace.csv" | head -1) 0 > $OUT/timeline_multibilliards.txt; rm -rf $OUT/mb
# the kernel statistics with the recognition network's forward chain on one stream: no two GEMM launches overlap (like for like with round 3)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/one_stream -o ks -- python3 $R/bench.py --enc-chunks 1 --steps 20 --warmup 3 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode graph > $OUT/one_stream.log 2>&1
s1=$(find $OUT/one_stream -name "*kernel_stats.csv" | head -1); [ -n "$s1" ] && cp $s1 $OUT/kernel_stats_one_stream.csv; rm -rf $OUT/one_stream
unset STOVE_BENCH_NO_PARITY
cd $R
timeout 200 python3 tools/loop_stamps.py > $OUT/loop_stamps.txt 2>&1
# what profiles/ holds of this run
for f in bench.json bench_avoidance.json bench_gravity.json bench_multibilliards.json bench_store_bw32.json bench_store_u8.json \
         kernel_stats.csv kernel_stats_one_stream.csv kernel_times.txt loop_stamps.txt pmc_sq.json pmc_traffic.json timeline.txt timeline_all.txt timeline_eager.txt \
         timeline_multibilliards.txt; do
  [ -s $OUT/$f ] && cp $OUT/$f $OUT/../${TAG}_final_$f
done
[ -s gpurun_out/parity_errors.json ] && cp gpurun_out/parity_errors.json $OUT/../${TAG}_final_parity_errors.json
