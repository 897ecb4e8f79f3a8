#!/bin/bash
# as ab_lib.sh, printing the per-kernel device time per step of the bench's own profiler next to the step time
R=$GRAFT_REPO_ROOT; cd $R
A=$1; B=$2; shift 2
cp stove_amd/libstove_hip.so /tmp/keep.so
for v in A B A B; do
  if [ $v = A ]; then cp stove_amd/$A stove_amd/libstove_hip.so; E="$STOVE_AB_ENV_A"; else cp /tmp/keep.so stove_amd/libstove_hip.so; [ "$B" != "-" ] && cp stove_amd/$B stove_amd/libstove_hip.so; E="$STOVE_AB_ENV_B"; fi
  env $E STOVE_BENCH_NO_PARITY=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']
print('$v', round(d['ms_per_step'],4), round(d['ms_per_step_p50'],4), {n: k[n] for n in ('gemm_bf16_k','dyn_loop_fwd_small_k','dyn_loop_bwd_small_k','objspn_tablegrad_under_k','enc_head_fwd_k','gnn_dw_small_k') if n in k}, 'gemm launches', d['roofline'].get('launches'), 'avg', d['roofline'].get('avg_ms'))"
done
cp /tmp/keep.so stove_amd/libstove_hip.so
