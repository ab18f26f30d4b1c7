#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the round's bench line, the rocprofv3 kernel-trace summary of the same command and
# the two PMC passes behind roofline.traffic.  Everything lands under gpurun_out/final/ (copy to profiles/ afterwards).
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --steps 20 --warmup 5 > "$OUT/bench_b32_bf16.json" 2> "$OUT/bench_b32_bf16.err"
python3 "$ROOT/bench.py" --steps 10 --warmup 3 --precision fp32 --no-cpu-baseline > "$OUT/bench_b32_fp32.json" 2>> "$OUT/bench_b32_bf16.err"
rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- \
    python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
cp /tmp/prof_kt/*kernel_stats.csv "$OUT/kernel_stats.csv"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -o pmc -- \
      python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > "$OUT/pmc_$C.log" 2>&1
  cp /tmp/prof_$C/*counter_collection.csv "$OUT/pmc_$C.csv"
done
python3 "$ROOT/tools/pmc_aggregate.py" "$OUT/pmc_traffic_b32_bf16.json" FETCH_SIZE="$OUT/pmc_FETCH_SIZE.csv" WRITE_SIZE="$OUT/pmc_WRITE_SIZE.csv"
rm -f "$OUT"/pmc_*.csv
ls -la "$OUT"
