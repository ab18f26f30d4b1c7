#!/bin/bash
# PMC passes (MFMA busy / clock / waits / LDS conflicts) of the f16mx forward GEMM next to the bf16x2 lean tile on fc1's shape
# (tools/mx_gemm_ab.py).  Output: gpurun_out/r6b/mxpmc/pmc_mx.json
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6b/mxpmc
SHAPES=${1:-16384x4096x25088}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A PASS
PASS[C]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
PASS[D]="TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"
ARGS=""
for C in C D; do
  rm -rf /tmp/prof_$C && timeout 300 rocprofv3 --pmc ${PASS[$C]} --kernel-trace --output-format csv -d /tmp/prof_$C -o pmc -- python3 $ROOT/tools/mx_gemm_ab.py $SHAPES > $OUT/log_$C.txt 2>&1
  cp /tmp/prof_$C/*counter_collection.csv $OUT/pmc_$C.csv 2>/dev/null && ARGS="$ARGS $C=$OUT/pmc_$C.csv"
done
python3 $ROOT/tools/pmc_aggregate.py $OUT/pmc_mx.json $ARGS
rm -f $OUT/pmc_*.csv
