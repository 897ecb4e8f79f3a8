#!/bin/bash
# A/B of the fused LSTM path (STOVE_LSTM_FUSED) + kernel timeline of the fused step.  gpurun -- 'bash tools/lstm_ab.sh'
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/lstm; mkdir -p $OUT
cd $R
for f in 0 1 0 1; do STOVE_LSTM_FUSED=$f STOVE_BENCH_NO_PARITY=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused=$f', round(d['ms_per_step'],4), round(d['ms_per_step_p50'],4))"; done
cd /tmp && export TMPDIR=/tmp
export STOVE_BENCH_NO_PARITY=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode graph > $OUT/ks.log 2>&1
f=$(find $OUT/ks -name "*kernel_trace.csv" | head -1); python3 $R/tools/timeline.py $f 10 > $OUT/timeline_fused.txt
rm -rf $OUT/ks
head -60 $OUT/timeline_fused.txt
