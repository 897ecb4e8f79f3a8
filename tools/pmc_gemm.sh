#!/bin/bash
# PMC passes over the gx GEMM: memory side (TCC hits / misses, HBM fetch), then the SQ wait breakdown and LDS conflicts.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_gemm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# every pass under `timeout`: a counter set that does not fit the block's slots (TCC has 4, FETCH_SIZE takes 3) makes rocprofv3
# abort and then hang until the box limit
timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/mem -o mem -- python3 $R/tools/gemm_pmc.py $1 > $OUT/mem.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/mem2 -o mem2 -- python3 $R/tools/gemm_pmc.py $1 > $OUT/mem2.log 2>&1
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o sq -- python3 $R/tools/gemm_pmc.py $1 > $OUT/sq.log 2>&1
timeout 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $OUT/sq2 -o sq2 -- python3 $R/tools/gemm_pmc.py $1 > $OUT/sq2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in ('mem', 'mem2', 'sq', 'sq2'):
    for f in glob.glob('gpurun_out/pmc_gemm/%s/**/*counter_collection.csv' % tag, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm_bf16' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(tag, k, 'per launch: %.4g' % (sum(v) / len(v)), 'n=%d' % len(v))
PY
rm -rf $OUT/mem $OUT/mem2 $OUT/sq $OUT/sq2
