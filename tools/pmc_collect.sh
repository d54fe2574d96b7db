#!/bin/bash
# Collect PMC counters of one hot-path launch in separate passes (no trace domains besides the kernel list).
# usage (on the GPU box): tools/pmc_collect.sh OUTDIR NPOINTS
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/$1; N=${2:-16384}; CFG=${3:-cfg2_powerlaw_jI_aI}; SEL=${4:-}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SALU" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/one_batch.py $N $CFG $SEL > $OUT/log$i.txt 2>&1 || { tail -5 $OUT/log$i.txt; exit 1; }
done
python3 $R/tools/pmc_summary.py $OUT/summary.json $OUT/p* > /dev/null
grep "kernel ms" $OUT/log1.txt
