#!/bin/bash
# A/B of several builds of the library on one box, alternating runs: bash tools/ab_libs.sh "a.so b.so -" [bench args]
# (paths relative to stove_amd/; "-" = the library as shipped; each is copied over libstove_hip.so before its runs)
R=$GRAFT_REPO_ROOT; cd $R
LIBS=$1; shift
cp stove_amd/libstove_hip.so /tmp/keep.so
for rep in 1 2; do for L in $LIBS; do
  if [ "$L" = "-" ]; then cp /tmp/keep.so stove_amd/libstove_hip.so; else cp stove_amd/$L stove_amd/libstove_hip.so; fi
  env STOVE_BENCH_NO_PARITY=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L $*', round(d['ms_per_step'],4), round(d['ms_per_step_p50'],4))"
done; done
cp /tmp/keep.so stove_amd/libstove_hip.so
