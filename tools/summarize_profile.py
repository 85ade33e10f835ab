#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE passes) into a small
JSON the repo keeps under profiles/ and bench.py cites for `roofline.traffic`.

HBM traffic is priced exactly as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) says:
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide
coalesced (16 B/lane) read stream, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
Infinity-Cache hits are included in these fabric-side counters.
"""
import csv
import json
import statistics
import sys


def pmc(path, counter, kernel_substr):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]]
    return statistics.median(vals), len(vals)


def main():
    stats_csv, fetch_csv, write_csv, kernel_substr, out = sys.argv[1:6]
    ks = [r for r in csv.DictReader(open(stats_csv)) if kernel_substr in r["Name"]]
    k = max(ks, key=lambda r: float(r["TotalDurationNs"]))
    f, nf = pmc(fetch_csv, "FETCH_SIZE", kernel_substr)
    w, nw = pmc(write_csv, "WRITE_SIZE", kernel_substr)
    res = {
        "kernel": k["Name"], "calls": int(k["Calls"]), "avg_ns": float(k["AverageNs"]),
        "min_ns": float(k["MinNs"]), "max_ns": float(k["MaxNs"]), "pct_of_gpu_time": float(k["Percentage"]),
        "FETCH_SIZE_KiB_median": f, "FETCH_SIZE_launches": nf,
        "WRITE_SIZE_KiB_median": w, "WRITE_SIZE_launches": nw,
        "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0,
        "traffic_formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024  [gfx950: FETCH_SIZE counts half of wide reads]",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
