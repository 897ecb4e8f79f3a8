#!/bin/bash
# A/B of two builds of the library on one box, alternating runs: bash tools/ab_lib.sh base.so new.so [bench args]
# (paths relative to stove_amd/; the file is copied over libstove_hip.so before each run; STOVE_AB_ENV_A / _B: extra environment)
R=$GRAFT_REPO_ROOT; cd $R
A=$1; B=$2; shift 2
cp stove_amd/libstove_hip.so /tmp/keep.so
for v in A B A B A B; do
  if [ $v = A ]; then cp stove_amd/$A stove_amd/libstove_hip.so; E="$STOVE_AB_ENV_A"; else cp /tmp/keep.so stove_amd/libstove_hip.so; [ "$B" != "-" ] && cp stove_amd/$B stove_amd/libstove_hip.so; E="$STOVE_AB_ENV_B"; fi
  env $E STOVE_BENCH_NO_PARITY=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['ms_per_step_p50'],4))"
done
cp /tmp/keep.so stove_amd/libstove_hip.so
