#!/bin/bash
# What do the scene / SPN kernels wait for?  Three rocprofv3 --pmc passes (8 SQ slots each, --kernel-trace only) over the eager
# step of the default bench workload -> gpurun_out/$TAG/pmc_sq_scene.json (per-launch averages, all kernels).
# Usage: gpurun --timeout 1500 -- 'bash tools/pmc_scene.sh r06 [extra bench flags]'
TAG=${1:-r06}; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export STOVE_BENCH_NO_PARITY=1
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --profile-steps 0 --step-mode eager $@"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pa -o pa -- $B > $OUT/pa.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $OUT/pb -o pb -- $B > $OUT/pb.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/pc -o pc -- $B > $OUT/pc.log 2>&1
cd $R
python3 tools/pmc_sq_summary.py $(find $OUT/pa -name "*counter_collection.csv" | head -1) $OUT/pa.json
python3 tools/pmc_sq_summary.py $(find $OUT/pb -name "*counter_collection.csv" | head -1) $OUT/pb.json
python3 tools/pmc_sq_summary.py $(find $OUT/pc -name "*counter_collection.csv" | head -1) $OUT/pc.json
python3 - $OUT <<'PY'
import json, sys
o = sys.argv[1]
res = {}
for p in ('pa', 'pb', 'pc'):
    try:
        d = json.load(open(f'{o}/{p}.json'))
    except Exception as e:
        print(p, 'missing', e); continue
    for k, v in d.items():
        res.setdefault(k, {}).update(v)
json.dump(res, open(f'{o}/pmc_sq_scene.json', 'w'), indent=1, sort_keys=True)
for k in ('objspn_fwd_unit_k', 'bgspn_bwd_k', 'scene_pixtile_bwd_k', 'bgspn_mfma_fwd_k', 'scene_tile_fwd_k', 'objspn_tablegrad_under_k'):
    if k in res:
        print(k, {c: round(x) for c, x in res[k].items()})
PY
rm -rf $OUT/pa $OUT/pb $OUT/pc
tail -3 $OUT/pa.log
