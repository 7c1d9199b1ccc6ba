#!/usr/bin/env python3
"""Condense the held-out parity rows of gpurun_out/parity_stats.jsonl (written by tests/test_gpu_heldout.py on MI355X) into the
small record bench.py quotes beside its two rates: profiles/<tag>_parity_heldout.json.
usage: tools/parity_record.py round4"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    rows = {}
    with open(os.path.join(ROOT, "gpurun_out", "parity_stats.jsonl")) as f:
        for line in f:
            try:
                d = json.loads(line)
            except ValueError:
                continue
            if str(d.get("test", "")).startswith("heldout_"):
                rows[d["test"]] = d                       # the last row of a name wins
    a, h = rows["heldout_cfg2_fp32_weights"], rows["heldout_cfg2_fp32_weights_precision_high"]
    rec = {
        "set": "64 held-out clips (seed 777, 1-30 s, %d frames), product Labeler vs the oracle at B = 1 on the checkpoint as given "
               "(tests/test_gpu_heldout.py)" % int(a["frames"]),
        "raw_mismatch_default": a["raw_mismatch_rate"], "raw_mismatch_high": h["raw_mismatch_rate"],
        "clips_identical_default": int(a["clips_with_identical_label_sequence"]),
        "clips_identical_high": int(h["clips_with_identical_label_sequence"]), "clips": int(a["clips"]),
        "graded_frac_default": a["graded_frac"], "graded_frac_high": h["graded_frac"],
        "tau_default": a["tau"], "tau_high": h["tau"],
        "graded_mismatches_default": int(a["graded_mismatches"]), "graded_mismatches_high": int(h["graded_mismatches"]),
        "max_boundary_shift_s_high": h["max_boundary_shift_s"],
    }
    out = os.path.join(ROOT, "profiles", tag + "_parity_heldout.json")
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")
    print(out, rec)


if __name__ == "__main__":
    main()
