#!/bin/bash
# Round-end measurement passes on the GPU box (run through gpurun): rocprofv3 kernel stats, then the two HBM
# PMC passes (separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes), then the default bench line.
# Usage: gpurun --timeout 3000 -- 'bash tools/profile_round.sh r01'
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export STOVE_BENCH_NO_PARITY=1      # the profiled passes hold the timed workload only (no golden-fixture leg)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode graph > $OUT/ks.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode eager > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode eager > $OUT/write.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/sq -o sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode eager > $OUT/sq.log 2>&1
cd $R
# summaries (the raw traces are tens of MB: only these are merged back)
f=$(find $OUT/ks -name "*kernel_trace.csv" | head -1); python3 tools/timeline.py $f 20 > $OUT/timeline.txt
s=$(find $OUT/ks -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp $s $OUT/kernel_stats.csv
python3 tools/pmc_summary.py $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $(find $OUT/write -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json
python3 tools/pmc_sq_summary.py $(find $OUT/sq -name "*counter_collection.csv" | head -1) $OUT/pmc_sq.json
rm -rf $OUT/ks $OUT/fetch $OUT/write $OUT/sq
# the bench line quotes roofline.traffic from profiles/*pmc_traffic.json (source-hash checked): make this run's summary the one it finds
cp $OUT/pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
unset STOVE_BENCH_NO_PARITY
python3 bench.py > $OUT/bench.json 2> $OUT/bench.log
tail -c 900 $OUT/bench.json
