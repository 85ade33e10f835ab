#!/usr/bin/env python3
"""Clocks and power of the GPU while a command runs: samples the amdgpu sysfs files (pp_dpm_sclk / pp_dpm_mclk current
level, hwmon power1_average / power1_input, freq1_input) every 20 ms in this process while the command runs as a child.
python tools/power_probe.py -- python bench.py --leg headline --steps 60"""
import glob
import json
import subprocess
import sys
import time


def read(p):
    try:
        return open(p).read()
    except OSError:
        return ""


def cur_level(txt):
    for l in txt.splitlines():
        if l.strip().endswith("*"):
            return l.split(":")[1].replace("*", "").strip()
    return None


def main():
    cmd = sys.argv[sys.argv.index("--") + 1:]
    devs = [c.rsplit("/", 1)[0] for c in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))]
    if not devs:
        print(json.dumps({"error": "no amdgpu sysfs"}))
        return subprocess.call(cmd)
    hws = [(glob.glob(d + "/hwmon/hwmon*") or [None])[0] for d in devs]
    p = subprocess.Popen(cmd)
    per = [[] for _ in devs]
    t0 = time.time()
    while p.poll() is None:
        for i, dev in enumerate(devs):  # every card: the one the child runs on is picked afterwards by its busy share
            row = {"t": round(time.time() - t0, 3), "sclk": cur_level(read(dev + "/pp_dpm_sclk")),
                   "mclk": cur_level(read(dev + "/pp_dpm_mclk")), "busy": int(read(dev + "/gpu_busy_percent").strip() or 0)}
            if hws[i]:
                for k in ("power1_average", "power1_input"):
                    v = read(hws[i] + "/" + k).strip()
                    if v:
                        row[k] = int(v)
            per[i].append(row)
        time.sleep(0.02)
    best = max(range(len(devs)), key=lambda i: sum(r["busy"] for r in per[i]))
    rows = per[best]
    print(json.dumps({"cards": len(devs), "picked": devs[best], "busy_sum_per_card": [sum(r["busy"] for r in q) for q in per]}))
    out = {"cmd": cmd, "n": len(rows), "rows": rows[::3]}
    print(json.dumps(out))
    return p.returncode


if __name__ == "__main__":
    sys.exit(main())
