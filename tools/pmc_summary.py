#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes of bench.py into profiles/<tag>_pmc_traffic.json.

Usage: tools/pmc_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv OUT.json
HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KB counters; on gfx950 FETCH_SIZE reports exactly half of the
bytes of a wide coalesced read — MI355X_MICROARCH.md §HBM — which the layernorm kernel confirms here: 11.76 MB
reported for a 23.5 MB read).  Separate passes, as the guide prescribes (FETCH_SIZE and WRITE_SIZE do not fit one)."""
import collections
import csv
import json
import sys


def agg(path, cname):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != cname:
            continue
        k = r["Kernel_Name"]
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return d


def main():
    F, W = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
    out = {"unit": "MB per launch", "correction": "hbm = 2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE x2)", "kernels": {}}
    g_n = g_b = 0
    for k in sorted(F, key=lambda k: -F[k][1]):
        n, v = F[k]
        wn, wv = W.get(k, [0, 0.0])
        fetch = 2 * v / n / 1024
        write = wv / max(wn, 1) / 1024
        out["kernels"][k] = {"launches": n, "fetch_mb": round(fetch, 2), "write_mb": round(write, 2), "hbm_mb": round(fetch + write, 2)}
        if k.startswith(("void gemm_bf16_kernel", "void gemm256_kernel", "void gemm_stream_kernel")):
            g_n += n
            g_b += n * (fetch + write)
    out["gemm_family_hbm_mb_per_launch"] = round(g_b / max(g_n, 1), 2)
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))


if __name__ == "__main__":
    main()
