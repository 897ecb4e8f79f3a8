#!/bin/bash
# A/B of builds of the library on one box, alternating runs:  bash tools/ab_lib.sh "a.so b.so -" [bench args]
# Paths relative to stove_amd/; "-" = the library as shipped.  The builds are selected with STOVE_LIB (stove_amd/settings.py): the
# installed libstove_hip.so is never touched.  KERNELS=1 prints the per-kernel device time per step next to the step time.
R=$GRAFT_REPO_ROOT; cd $R
LIBS=$1; shift
PS=0; [ -n "$KERNELS" ] && PS=3
for rep in 1 2 3; do for L in $LIBS; do
  if [ "$L" = "-" ]; then E=""; else E="STOVE_LIB=$R/stove_amd/$L"; fi
  env $E STOVE_BENCH_NO_PARITY=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --profile-steps $PS "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; k=r.get('kernels_ms_per_step') or {}
print('$L', round(d['ms_per_step'],4), round(d['ms_per_step_p50'],4), {n: k[n] for n in ('gemm_bf16_k','dyn_loop_fwd_small_k','dyn_loop_bwd_small_k','objspn_tablegrad_under_k','gnn_dw_small_k','objspn_fwd_unit_k','bgspn_bwd_k','scene_pixtile_bwd_k') if n in k})"
done; done
