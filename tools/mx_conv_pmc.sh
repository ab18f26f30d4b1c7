#!/bin/bash
# MFMA busy / clock / waits of the f16mx conv next to the bf16x2 lean conv on the benchmark's res4 / res5 shapes
# (tools/mx_conv_ab.py under rocprofv3 --pmc) -> gpurun_out/r6b/convpmc/pmc_conv.json
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6b/convpmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_C && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/prof_C -o pmc -- python3 $ROOT/tools/mx_conv_ab.py > $OUT/log_C.txt 2>&1
cp /tmp/prof_C/*counter_collection.csv $OUT/pmc_C.csv
python3 $ROOT/tools/pmc_aggregate.py $OUT/pmc_conv.json C=$OUT/pmc_C.csv > /dev/null
python3 - <<XX
import json
d = json.load(open("$OUT/pmc_conv.json"))
for k, v in (d.get("kernels") or d).items():
    if "gemm256" in k:
        clk = v["GRBM_GUI_ACTIVE_avg"] / 8 / v["avg_duration_us"] / 1e3
        print(k[40:100], "us", v["avg_duration_us"], "mfma_util", v["mfma_util"], "clock GHz %.3f" % clk, "wait_any", v["sq_wait_any_share_of_wave_cycles"])
XX
rm -f $OUT/pmc_C.csv
