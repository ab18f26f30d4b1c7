"""Aggregate rocprofv3 --pmc counter_collection CSVs (one pass per counter group) into the compact per-kernel table
bench.py reads (profiles/rNN_pmc_traffic_b<batch>_bf16.json).

    python tools/pmc_aggregate.py OUT.json FETCH_SIZE=<csv> WRITE_SIZE=<csv> [SQ=<csv>] [LDS=<csv>]

Units / corrections (MI355X_MICROARCH.md, HBM + rocprofv3 sections):
* FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so
  traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch.  Infinity-Cache hits are counted, not excluded.
* SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (16 per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16);
  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
  mfma_util = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): the share of the kernel's SIMD-cycles in which the matrix
  pipe was busy; clock_ghz = GRBM_GUI_ACTIVE / 8 / kernel duration (reads high on dispatches under ~0.3 ms).
Only this library's kernels are kept; values are averages per launch.
"""
import csv
import json
import sys
from collections import defaultdict


def load(path):
    """-> {kernel: {counter: [sum, n]}, '_dur': [sum_ns, n]}"""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    seen = set()
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or row.get("Name")
            ctr = row.get("Counter_Name") or "value"
            val = float(row.get("Counter_Value") or 0.0)
            a = acc[name][ctr]
            a[0] += val
            a[1] += 1
            disp = row.get("Dispatch_Id")
            if disp is not None and (disp, name) not in seen and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                seen.add((disp, name))
                d = acc[name]["_dur_ns"]
                d[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                d[1] += 1
    return acc


def ours(name):
    return "wsovod" in name or "_GLOBAL__N_" in name or "(anonymous namespace)" in name


def main():
    out = sys.argv[1]
    passes = dict(a.split("=", 1) for a in sys.argv[2:])
    tables = {k: load(v) for k, v in passes.items()}
    kernels = {}
    for name in sorted(set().union(*[set(t) for t in tables.values()])):
        if not ours(name):
            continue
        rec = {}
        for t in tables.values():
            for ctr, (s, n) in t.get(name, {}).items():
                if ctr == "_dur_ns":
                    rec.setdefault("launches_sampled", n)
                    rec.setdefault("avg_duration_us", round(s / max(n, 1) / 1e3, 2))
                else:
                    rec[ctr + "_avg"] = round(s / max(n, 1), 3)
                    rec["launches_sampled"] = max(rec.get("launches_sampled", 0), n)
        fk, wk = rec.get("FETCH_SIZE_avg", 0.0), rec.get("WRITE_SIZE_avg", 0.0)
        if "FETCH_SIZE_avg" in rec or "WRITE_SIZE_avg" in rec:
            rec["FETCH_SIZE_KB_avg"], rec["WRITE_SIZE_KB_avg"] = fk, wk
            rec["traffic_bytes_per_launch"] = round((2.0 * fk + wk) * 1024.0, 1)
        gui = rec.get("GRBM_GUI_ACTIVE_avg")
        if gui:
            sq = tables.get("SQ", {}).get(name, {})
            dur = sq.get("_dur_ns")
            cyc = gui / 8.0
            if dur and dur[1]:
                rec["clock_ghz"] = round(cyc / (dur[0] / dur[1]), 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES_avg" in rec:
                rec["mfma_util"] = round(rec["SQ_VALU_MFMA_BUSY_CYCLES_avg"] / (cyc * 1024.0), 4)
            wc = rec.get("SQ_WAVE_CYCLES_avg")
            if wc:
                for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                    if k + "_avg" in rec:
                        rec[k.lower() + "_share_of_wave_cycles"] = round(rec[k + "_avg"] / wc, 4)
        h, m = rec.get("TCC_HIT_sum_avg"), rec.get("TCC_MISS_sum_avg")
        if h is not None and m is not None and h + m > 0:
            rec["l2_hit_rate"] = round(h / (h + m), 4)
        kernels[name] = rec
    doc = {
        "source": "rocprofv3 --pmc <group> --kernel-trace (one run per group: FETCH_SIZE | WRITE_SIZE | SQ_* + GRBM_GUI_ACTIVE | "
                  "LDS/TCC) -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-side --no-parity "
                  "(32 images/GPU/step, bf16, 1x MI355X)",
        "unit_note": __doc__.split("Units / corrections")[1].strip(),
        "kernels": kernels,
    }
    with open(out, "w") as fo:
        json.dump(doc, fo, indent=1)
    print(f"{out}: {len(kernels)} kernels")


if __name__ == "__main__":
    main()
