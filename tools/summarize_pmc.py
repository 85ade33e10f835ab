#!/usr/bin/env python3
"""Condense rocprofv3 --pmc passes (one directory per pass under DIR) into per-kernel averages.

usage: summarize_pmc.py DIR KERNEL_SUBSTRING  -> JSON on stdout: for every kernel whose name holds the substring,
the mean of each counter per dispatch (first dispatch of each kernel dropped as warm-up where there are more than
two), the kernel-trace average duration, and derived ratios.  Units as rocprofv3 reports them: SQ_*_CYCLES /
SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles summed over waves (or SIMDs for BUSY), GRBM_GUI_ACTIVE summed over
the 8 XCDs, FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE counts half of wide reads on gfx950).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(\w+)<(.*)>\(", name)
    return "%s<%s>" % (m.group(1), m.group(2).replace(" ", "")) if m else name.split("(")[0]


def main():
    root, sub = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if sub in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, counters in acc.items():
        ent = {}
        for c, v in sorted(counters.items()):
            v = v[1:] if len(v) > 2 else v
            ent[c] = sum(v) / len(v)
        out[k] = ent
    for path in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if sub in r["Name"] and short(r["Name"]) in out:
                out[short(r["Name"])].update(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]))
    for k, e in out.items():
        d = {}
        wc = e.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                      "SQ_ACTIVE_INST_LDS"):
                if c in e:
                    d[c.lower() + "_per_wave_cycle"] = e[c] / wc
        if e.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_fraction"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
        if "GRBM_GUI_ACTIVE" in e and "avg_ns" in e:
            d["gpu_clock_GHz"] = e["GRBM_GUI_ACTIVE"] / 8.0 / e["avg_ns"]
            cyc = e["GRBM_GUI_ACTIVE"] / 8.0
            if "SQ_ACTIVE_INST_VALU" in e:  # quad-cycles of VALU issue summed over waves / (cycles x 1024 SIMDs / 4)
                d["valu_busy_fraction_of_simd_time"] = e["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0)
            if "SQ_LDS_IDX_ACTIVE" in e:
                d["lds_busy_fraction_of_cu_time"] = e["SQ_LDS_IDX_ACTIVE"] / (cyc * 256.0)
        if "TCC_HIT_sum" in e:
            d["l2_hit_rate"] = e["TCC_HIT_sum"] / max(1.0, e["TCC_HIT_sum"] + e.get("TCC_MISS_sum", 0.0))
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            d["traffic_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
        e["derived"] = d
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
