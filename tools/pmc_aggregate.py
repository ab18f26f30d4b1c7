"""Aggregate rocprofv3 --pmc counter_collection CSVs (one pass per counter) into the compact per-kernel traffic table
bench.py reads (profiles/rNN_pmc_traffic_b<batch>_bf16.json).

    python tools/pmc_aggregate.py OUT.json FETCH_SIZE=<csv> WRITE_SIZE=<csv>

Units / corrections (MI355X_MICROARCH.md, HBM + rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KB; on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads, so traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes
per launch.  Infinity-Cache hits are counted, not excluded.  Only this library's kernels are kept.
"""
import csv
import json
import sys
from collections import defaultdict


def load(path):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or row.get("Name")
            val = float(row.get("Counter_Value") or 0.0)
            a = acc[name]
            a[0] += val
            a[1] += 1
    return acc


def main():
    out = sys.argv[1]
    passes = dict(a.split("=", 1) for a in sys.argv[2:])
    tables = {k: load(v) for k, v in passes.items()}
    kernels = {}
    for name in sorted(set().union(*[set(t) for t in tables.values()])):
        if "wsovod" not in name and "_GLOBAL__N_" not in name and "(anonymous namespace)" not in name:
            continue
        f = tables.get("FETCH_SIZE", {}).get(name, [0.0, 0])
        w = tables.get("WRITE_SIZE", {}).get(name, [0.0, 0])
        n = max(f[1], w[1], 1)
        fk, wk = f[0] / max(f[1], 1), w[0] / max(w[1], 1)
        kernels[name] = {"launches_sampled": n, "FETCH_SIZE_KB_avg": round(fk, 3), "WRITE_SIZE_KB_avg": round(wk, 3),
                         "traffic_bytes_per_launch": round((2.0 * fk + wk) * 1024.0, 1)}
    doc = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py "
                  "--steps 3 --warmup 2 --no-cpu-baseline --no-roofline (16 images/GPU/step, bf16, 1x MI355X)",
        "unit_note": "counter values are KB; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x "
                     "(MI355X_MICROARCH.md, HBM), so traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; Infinity-Cache "
                     "hits are counted, not excluded",
        "kernels": kernels,
    }
    with open(out, "w") as fo:
        json.dump(doc, fo, indent=1)
    print(f"{out}: {len(kernels)} kernels")


if __name__ == "__main__":
    main()
