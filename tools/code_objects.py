#!/usr/bin/env python3
"""Reads the gfx950 code objects inside the built .o files (csrc/*.o) and lists every kernel's resources from the
AMDGPU metadata notes: VGPRs, AGPRs, LDS bytes, scratch bytes per lane (private_segment_fixed_size) and spilled
registers.  `python tools/code_objects.py [--json] [--scratch-only]`.

tests/test_code_objects.py imports `kernels()` and fails on scratch in a default-path kernel (a scratch reload is
a vmcnt(0): DESIGN.md s.8 finding 0b) and on growth of the instantiation count.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "full_waveform_inversion_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FIELDS = ("agpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count",
          "vgpr_count", "vgpr_spill_count", "max_flat_workgroup_size")


def tools_present():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"))


def _notes(obj, tmp):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    if not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return ""
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=" + TARGET, "--output=" + co], stderr=subprocess.DEVNULL)
    return subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)


def kernels(objs=None):
    """[{obj, name (demangled), symbol, vgpr_count, agpr_count, private_segment_fixed_size, ...}] of every kernel."""
    if objs is None:
        objs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".o"))
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            cur = None
            for line in _notes(obj, tmp).splitlines():
                # a kernel's entry starts with "  - .<key>:" at two spaces; its own keys sit at four, its args deeper
                m = re.match(r"^(  - |    )\.(\w+):\s+(.*)$", line)
                if not m:
                    continue
                if m.group(1) == "  - ":
                    cur = {"obj": os.path.basename(obj)}
                    out.append(cur)
                if cur is None:
                    continue
                key, val = m.group(2), m.group(3).strip().strip("'")
                if key in FIELDS:
                    cur[key] = int(val)
                elif key == "name":
                    cur["symbol"] = val
    out = [k for k in out if "symbol" in k and "vgpr_count" in k]
    if out:
        try:
            names = subprocess.run(["c++filt"], input="\n".join(k["symbol"] for k in out), text=True, check=True,
                                   stdout=subprocess.PIPE).stdout.splitlines()
        except (OSError, subprocess.CalledProcessError):
            names = [k["symbol"] for k in out]
        for k, n in zip(out, names):
            k["name"] = n
    return out


def main():
    ks = kernels()
    if "--scratch-only" in sys.argv:
        ks = [k for k in ks if k.get("private_segment_fixed_size", 0) > 0]
    if "--json" in sys.argv:
        json.dump(ks, sys.stdout, indent=1)
        return
    per = {}
    for k in ks:
        per[k["obj"]] = per.get(k["obj"], 0) + 1
        print("%-18s v%-4d a%-4d lds %-7d scratch %-5d spill %-4d %s" % (
            k["obj"], k["vgpr_count"], k.get("agpr_count", 0), k.get("group_segment_fixed_size", 0),
            k.get("private_segment_fixed_size", 0), k.get("vgpr_spill_count", 0), k["name"][:200]))
    print("kernels per object:", per, "total", sum(per.values()))


if __name__ == "__main__":
    main()
