#!/usr/bin/env python3
"""gpurun_out/r04/ (written by tools/collect_r04.sh on the GPU box) -> the committed summaries under profiles/:
r04_kernel_stats_<leg>.csv (rocprofv3 --kernel-trace --stats), r04_traffic.json (per leg: the dominant kernel's
median FETCH_SIZE / WRITE_SIZE per launch from the two separate --pmc passes and the fabric-side bytes they imply,
priced as MI355X_MICROARCH.md section HBM prescribes) and r04_bench.json (the bench line of the same session).
Legs of several kernels per time step (the CPML: step kernel + slab kernels) also get `traffic_bytes_per_step`."""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r04")
DST = os.path.join(ROOT, "profiles")
# algorithmic bytes per launch (SURVEY s.8d): 16 B/update forward; gradient shot 22 B averaged over the forward+store
# (20 B) and adjoint + paired-imaging (24 B) launches; the 2-D fused kernel advances 4 time steps per launch
# increment form: forward + store 24 B, adjoint + paired imaging 28 B (round 4; 32 B unpaired before); bf16 store: 18 B / 20 B; fp64: 32 B; the CPML legs add
# the memory variables (psi, zeta read + written: 16 B per border cell and axis) -- per launch of 4 steps in 2-D, per
# time step (step kernel + slab launches together) in 3-D
N3 = 256 ** 3
LEGS = {"headline": 16.0 * N3, "hbm": 16.0 * 512 ** 3, "gradient": 22.0 * N3, "gradient_increment": 26.0 * N3,
        "cfg2": 4 * 16.0 * 1024 ** 2, "cfg2_cpml": 4 * 16.0 * 1024 ** 2 + 16.0 * 2 * 2 * 40 * 1024,
        "cpml3d": 16.0 * N3 + 16.0 * 3 * 2 * 16 * 256 ** 2, "fp64": 32.0 * N3, "point": 16.0 * N3, "bf16": 19.0 * N3,
        # the CPML gradient sweeps (store 20 B + adjoint with paired imaging 24 B, averaged) and the HBM-regime CPML run
        "cpml3d_adjoint": 22.0 * N3 + 16.0 * 3 * 2 * 16 * 256 ** 2, "cpml512": 16.0 * 512 ** 3 + 16.0 * 3 * 2 * 16 * 512 ** 2}
PER_STEP = {"cpml3d", "cpml3d_adjoint", "cpml512"}  # legs whose unit is the whole time step (several kernels)


def one(pattern):
    """newest match: gpurun merges a call's outputs INTO gpurun_out/, so an earlier call's files may still be there"""
    hits = sorted(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)
    return hits[-1] if hits else None


def main():
    traffic = {}
    for leg in LEGS:
        stats = one("kt_%s/*/*_kernel_stats.csv" % leg)
        if not stats:
            continue
        shutil.copy(stats, os.path.join(DST, "r04_kernel_stats_%s.csv" % leg))
        rows = [r for r in csv.DictReader(open(stats)) if "fwi::step" in r["Name"] or "fwi::pml_kernel" in r["Name"] or "fwi::pml_line" in r["Name"]]
        per_kernel = {}
        for r in rows:
            ent = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct_of_gpu_time": float(r["Percentage"])}
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                f = one("pmc_%s_%s/*/*_counter_collection.csv" % (leg, c))
                if f:
                    v = [float(x["Counter_Value"]) for x in csv.DictReader(open(f))
                         if x["Counter_Name"] == c and x["Kernel_Name"] == r["Name"]]
                    if v:
                        ent[c + "_KiB_median"] = statistics.median(v)
                        ent[c + "_launches"] = len(v)
            if "FETCH_SIZE_KiB_median" in ent and "WRITE_SIZE_KiB_median" in ent:
                ent["traffic_bytes_per_launch"] = (2.0 * ent["FETCH_SIZE_KiB_median"] + ent["WRITE_SIZE_KiB_median"]) * 1024.0
            per_kernel[r["Name"].split("(")[0].replace("void fwi::", "")] = ent
        dom = max(per_kernel.values(), key=lambda e: e["calls"] * e["avg_ns"])
        if leg in PER_STEP and all("traffic_bytes_per_launch" in e for e in per_kernel.values()):
            # time steps of the leg = launches of its step kernels (a gradient leg runs several variants: store,
            # imaging, plain adjoint -- one per time step of either sweep)
            steps = sum(e["calls"] for k, e in per_kernel.items() if k.startswith("step"))
            dom = {"calls": steps, "avg_ns": sum(e["calls"] * e["avg_ns"] for e in per_kernel.values()) / steps,
                   "traffic_bytes_per_launch": sum(e["calls"] * e["traffic_bytes_per_launch"] for e in per_kernel.values()) / steps,
                   "note": "per TIME STEP: all kernels of the leg (step kernel + line / slab launches) / step-kernel calls"}
        elif len(per_kernel) > 1 and all("traffic_bytes_per_launch" in e for e in per_kernel.values()):
            # several step kernels share the leg (gradient: store / plain / paired-imaging launches): call-weighted mean
            n = sum(e["calls"] for e in per_kernel.values())
            dom = {"calls": n, "avg_ns": sum(e["calls"] * e["avg_ns"] for e in per_kernel.values()) / n,
                   "traffic_bytes_per_launch": sum(e["calls"] * e["traffic_bytes_per_launch"] for e in per_kernel.values()) / n,
                   "note": "call-weighted mean over the leg's step kernels"}
        traffic[leg] = dict(dom, kernels=per_kernel,
                            traffic_formula="(2*FETCH_SIZE + WRITE_SIZE)*1024  [gfx950: FETCH_SIZE counts half of wide reads]")
        if LEGS[leg] and "traffic_bytes_per_launch" in dom:
            traffic[leg]["algorithmic_bytes_per_launch"] = LEGS[leg]
            traffic[leg]["traffic_over_algorithmic"] = dom["traffic_bytes_per_launch"] / LEGS[leg]
    json.dump(traffic, open(os.path.join(DST, "r04_traffic.json"), "w"), indent=1)
    b = os.path.join(SRC, "bench.json")
    if os.path.exists(b):
        shutil.copy(b, os.path.join(DST, "r04_bench.json"))
    for leg, t in traffic.items():
        print(leg, {k: v for k, v in t.items() if k not in ("kernels", "traffic_formula")})


if __name__ == "__main__":
    main()
