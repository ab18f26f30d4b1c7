#!/bin/bash
# MFMA busy / clock of the f16mx tile and its timing ablations (tools/mx_abl.py) -> gpurun_out/r6b/ablpmc/<lib>.json
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6b/ablpmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in product "$@"; do
  if [ "$L" != product ]; then export WSOVOD_LIB=$ROOT/wsovod_amd/lib/abl/$L; fi
  rm -rf /tmp/prof_C && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/prof_C -o pmc -- python3 $ROOT/tools/mx_abl.py > $OUT/log_$L.txt 2>&1
  cp /tmp/prof_C/*counter_collection.csv $OUT/pmc_C.csv 2>/dev/null
  python3 $ROOT/tools/pmc_aggregate.py $OUT/$L.json C=$OUT/pmc_C.csv > /dev/null
  python3 - <<XX
import json
d = json.load(open("$OUT/$L.json"))
for k, v in (d.get("kernels") or d).items():
    if "gemm256_mx" in k:
        clk = v["GRBM_GUI_ACTIVE_avg"] / 8 / v["avg_duration_us"] / 1e3
        print("$L", "us", v["avg_duration_us"], "mfma_util", v["mfma_util"], "clock GHz %.3f" % clk, "wait_any", v["sq_wait_any_share_of_wave_cycles"])
XX
  rm -f $OUT/pmc_C.csv
done
