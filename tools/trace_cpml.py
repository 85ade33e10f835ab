#!/usr/bin/env python3
"""Per-launch durations of one CPML time step from a rocprofv3 kernel trace (median over the steps of the first
sweep): which phase / axis of the border recursion costs what beside the step kernel.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/time_config.py --config cfg5 --nt 40 --rounds 1 --abc cpml
    python3 tools/trace_cpml.py OUT
"""
import collections
import csv
import glob
import os
import sys


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "pml_kernel" in r["Kernel_Name"] or "fwi::step" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seq = [("pml" if "pml_kernel" in r["Kernel_Name"] else "main", int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
            r["Kernel_Name"].split("(")[0][-40:]) for r in rows]
    # a forward step: [phase 1 x axes][step kernel][phase 3 x axes]; an adjoint step has phase 2 as well
    pats = collections.defaultdict(lambda: collections.defaultdict(list))
    i = 0
    while i < len(seq):
        if seq[i][0] != "pml":
            i += 1
            continue
        j = i
        while j < len(seq) and seq[j][0] == "pml":
            j += 1
        if j >= len(seq):
            break
        k = j + 1
        while k < len(seq) and seq[k][0] == "pml":
            k += 1
        key = (j - i, k - j - 1)
        for pos, g in enumerate(seq[i:k]):
            pats[key][pos].append(g[1])
        i = k
    for key, acc in pats.items():
        n = len(acc[0])
        tot = 0.0
        line = []
        for pos in sorted(acc):
            v = sorted(acc[pos])
            m = v[len(v) // 2] / 1e3
            tot += m
            line.append("%.1f" % m)
        print("step pattern %d slab launches + step kernel + %d slab launches (%d steps): %s  | sum %.1f us"
              % (key[0], key[1], n, " ".join(line), tot))


if __name__ == "__main__":
    main()
