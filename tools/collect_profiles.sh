#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the round's bench line, the rocprofv3 kernel-trace summary of the same command and
# the PMC passes behind roofline.traffic and the MFMA-utilisation table.  Counters are collected in their own runs with
# --kernel-trace only (no sys/hip/hsa trace domains).  Everything lands under gpurun_out/final/ (copy to profiles/).
#   gpurun --timeout 1800 -- 'bash tools/collect_profiles.sh'                     (the headline: parity_mx, 32 images / step)
#   gpurun --timeout 1800 -- 'PREC=bf16 bash tools/collect_profiles.sh'           (plain bf16, the side line)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PREC=${PREC:-parity_mx}
BATCH=${BATCH:-32}
TAG=b${BATCH}_${PREC}
OUT=$ROOT/gpurun_out/final
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--precision $PREC --batch $BATCH"
SHORT="$COMMON --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-side --no-parity"
FULL=""
if [ "$PREC" != "parity_mx" ]; then FULL="--no-side --no-parity --no-cpu-baseline"; fi
python3 "$ROOT/bench.py" $COMMON --steps 20 --warmup 5 $FULL > "$OUT/bench_$TAG.json" 2> "$OUT/bench_$TAG.err"
rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- \
    python3 "$ROOT/bench.py" $COMMON --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-parity > "$OUT/bench_${TAG}_under_rocprof.json" 2> "$OUT/rocprof_$TAG.err"
cp /tmp/prof_kt/*kernel_stats.csv "$OUT/kernel_stats_$TAG.csv"
declare -A PASS
PASS[FETCH_SIZE]="FETCH_SIZE"
PASS[WRITE_SIZE]="WRITE_SIZE"
PASS[SQ]="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
PASS[LDS]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum"
ARGS=""
for C in FETCH_SIZE WRITE_SIZE SQ LDS; do
  rm -rf /tmp/prof_$C && rocprofv3 --pmc ${PASS[$C]} --kernel-trace --output-format csv -d /tmp/prof_$C -o pmc -- \
      python3 "$ROOT/bench.py" $SHORT > "$OUT/pmc_$C.log" 2>&1
  cp /tmp/prof_$C/*counter_collection.csv "$OUT/pmc_$C.csv" 2>/dev/null && ARGS="$ARGS $C=$OUT/pmc_$C.csv"
done
python3 "$ROOT/tools/pmc_aggregate.py" "$OUT/pmc_$TAG.json" $ARGS
rm -f "$OUT"/pmc_*.csv
ls -la "$OUT"
