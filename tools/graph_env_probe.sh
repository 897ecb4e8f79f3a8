#!/bin/bash
# graph-replay step time under different runtime queue settings
out=gpurun_out/genv; mkdir -p $out
common="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode graph"
run() { tag=$1; shift; env "$@" timeout 200 python3 bench.py $common > $out/$tag.json 2> $out/$tag.err; echo "$tag $(python3 -c "import json;d=json.load(open('$out/$tag.json'));print(d['ms_per_step_p50'], d['ms_per_step'])" 2>&1 | tail -1)"; }
run base A=1
for q in 2 3 6 8 12 16; do run gq$q DEBUG_HIP_FORCE_GRAPH_QUEUES=$q; done
for h in 8 16; do run hq$h GPU_MAX_HW_QUEUES=$h; run hq${h}_gq8 GPU_MAX_HW_QUEUES=$h DEBUG_HIP_FORCE_GRAPH_QUEUES=8;  run hq${h}_gq16 GPU_MAX_HW_QUEUES=$h DEBUG_HIP_FORCE_GRAPH_QUEUES=16; done
run nopc DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run nopc_h8g8 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 GPU_MAX_HW_QUEUES=8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run dynq DEBUG_HIP_DYNAMIC_QUEUES=1
