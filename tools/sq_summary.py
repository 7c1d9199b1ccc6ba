#!/usr/bin/env python3
"""Summarise a rocprofv3 SQ counter pass of bench.py into profiles/<tag>_sq_counters.json.

Usage: tools/sq_summary.py sq_counter_collection.csv OUT.json
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); wait fractions are of SQ_WAVE_CYCLES."""
import collections
import csv
import json
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(sys.argv[1])):
        k, c = r["Kernel_Name"], r["Counter_Name"]
        acc[k][c] += float(r["Counter_Value"])
        n[k][c] += 1
    out = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -- python3 bench.py "
                   "--steps 4 --warmup 1 --inflight 1; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); "
                   "wait fractions are of SQ_WAVE_CYCLES", "kernels": {}}
    for k in sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", 0.0)):
        a = acc[k]
        launches = max(n[k].values())
        gui = a.get("GRBM_GUI_ACTIVE", 0.0)
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        out["kernels"][k] = {
            "launches": launches, "gui_active_per_launch": round(gui / max(launches, 1)),
            "mfma_busy": round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024), 3) if gui else None,
            "wait_any": round(a.get("SQ_WAIT_ANY", 0.0) / wc, 3) if wc else None,
            "wait_inst_lds": round(a.get("SQ_WAIT_INST_LDS", 0.0) / wc, 3) if wc else None,
        }
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(len(out["kernels"]), "kernels")


if __name__ == "__main__":
    main()
