#!/bin/bash
# Round-end measurement passes on the GPU box (run through gpurun): rocprofv3 kernel stats, then the two HBM
# PMC passes (separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes), then the default bench line.
# Usage: gpurun --timeout 3000 -- 'bash tools/profile_round.sh r01'
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0 > $OUT/ks.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 > $OUT/write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/sq -o sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 > $OUT/sq.log 2>&1
cd $R && python3 bench.py > $OUT/bench.json 2> $OUT/bench.log
find $OUT -name "*.csv" | head -20
tail -c 600 $OUT/bench.json
