#!/bin/bash
# Runs ON THE GPU BOX: LDS counters of the bf16x2 64-channel conv per case (tools/x2_probe.py, X2_PROBE_C64_ONLY=1),
# one dispatch row per launch -> summary per (kernel, grid size).  gpurun --timeout 900 -- 'bash tools/c64x_lds_pmc.sh'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c64x_lds
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export X2_PROBE_C64_ONLY=1
for V in 1 0; do
  export WSOVOD_C64X_LEPI=$V
  rm -rf /tmp/prof_c64 && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL \
      --kernel-trace --output-format csv -d /tmp/prof_c64 -o pmc -- python3 "$ROOT/tools/x2_probe.py" 4 > "$OUT/run_$V.log" 2>&1
  python3 - "$V" <<'PY' > "$OUT/summary_lepi$V.txt"
import csv, glob, sys, collections
rows = list(csv.DictReader(open(glob.glob('/tmp/prof_c64/*counter_collection.csv')[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'c64' not in r['Kernel_Name']: continue
    key = (r['Kernel_Name'][:60], r['Grid_Size'])
    acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, 'launches', len(next(iter(v.values()))))
PY
done
cat "$OUT"/summary_lepi*.txt
