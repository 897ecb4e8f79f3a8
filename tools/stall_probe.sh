#!/bin/bash
# Host-stall diagnosis: the driver's exact bench command on a fresh box, eager and graph step modes,
# with the watcher thread logging every thread's Python stack / wchan / syscall when a step overruns.
out=gpurun_out/stall; mkdir -p $out
export STOVE_BENCH_WATCH=1 STOVE_BENCH_GCLOG=1
common="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-variants"
for i in 1 2 3; do
  timeout 300 python3 bench.py $common --step-mode eager > $out/eager_$i.json 2> $out/eager_$i.err
  timeout 300 python3 bench.py $common --step-mode graph > $out/graph_$i.json 2> $out/graph_$i.err
done
HSA_ENABLE_INTERRUPT=0 timeout 300 python3 bench.py $common --step-mode eager > $out/eager_noint.json 2> $out/eager_noint.err
GPU_MAX_HW_QUEUES=2 timeout 300 python3 bench.py $common --step-mode eager > $out/eager_q2.json 2> $out/eager_q2.err
# long runs: does it recur?
timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline --no-variants --step-mode eager > $out/eager_200.json 2> $out/eager_200.err
timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline --no-variants --step-mode graph > $out/graph_200.json 2> $out/graph_200.err
# HIP API trace of the eager step: which runtime call blocks
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --hip-runtime-trace --stats -d $GRAFT_REPO_ROOT/$out/hiptrace -o t -- python3 $GRAFT_REPO_ROOT/bench.py $common --step-mode eager --profile-steps 0 > $GRAFT_REPO_ROOT/$out/hiptrace.json 2> $GRAFT_REPO_ROOT/$out/hiptrace.err
cd $GRAFT_REPO_ROOT
# keep only the summaries (the per-call csv can be large)
find $out/hiptrace -name '*hip_api_trace.csv' -size +20M -delete
grep -h "device ms\|host ms\|ms/step" $out/*.err | cut -c1-400
