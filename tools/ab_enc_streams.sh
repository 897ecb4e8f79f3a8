R=$GRAFT_REPO_ROOT; cd $R
run() { env "$@" STOVE_BENCH_NO_PARITY=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --profile-steps 0 $WL 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '$WL', round(d['ms_per_step'],4), round(d['ms_per_step_p50'],4))"; }
for rep in 1 2; do
  run STOVE_ENC_CHUNKS=2 STOVE_ENC_STREAMS=2
  run STOVE_ENC_CHUNKS=3 STOVE_ENC_STREAMS=3
  run STOVE_ENC_CHUNKS=4 STOVE_ENC_STREAMS=4
  run STOVE_ENC_CHUNKS=4 STOVE_ENC_STREAMS=3
done
WL="--workload multibilliards"
for rep in 1 2; do
  run STOVE_ENC_CHUNKS=2 STOVE_ENC_STREAMS=2
  run STOVE_ENC_CHUNKS=3 STOVE_ENC_STREAMS=3
done
