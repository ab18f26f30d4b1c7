#!/bin/bash
# on the GPU box: counter passes of the c64 probe (one shape)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export C64_ONLY=${C64_ONLY:-300,400,0}
declare -A PASS
PASS[A]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
PASS[B]="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
PASS[C]="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"
ARGS=""
for C in A B C; do
  rm -rf /tmp/prof_$C && timeout 300 rocprofv3 --pmc ${PASS[$C]} --kernel-trace --output-format csv -d /tmp/prof_$C -o pmc -- python3 $ROOT/tools/c64_probe.py > $OUT/log_$C.txt 2>&1
  cp /tmp/prof_$C/*counter_collection.csv $OUT/pmc_$C.csv 2>/dev/null && ARGS="$ARGS $C=$OUT/pmc_$C.csv"
done
python3 $ROOT/tools/pmc_aggregate.py $OUT/pmc_${TAG:-x}.json $ARGS
rm -f $OUT/pmc_*.csv
