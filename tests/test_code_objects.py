"""The built gfx950 code objects (csrc/*.o): no kernel may use scratch memory, and the instantiation count is pinned.

A register spill is a scratch store + reload, and in the pipelined step kernels a scratch reload is a `vmcnt(0)` that
drains every prefetch in flight (DESIGN.md s.8 finding 0b).  Round 3 shipped twelve step3d_stream instantiations with
44-92 B of scratch per lane that no test reached; this reads the AMDGPU metadata notes of every object and fails on any
`private_segment_fixed_size > 0`.  No GPU needed (hipcc cross-compiles; the .o files come from `make` /
`__graft_entry__.build()`)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import code_objects as co  # noqa: E402

OBJS = ["fwi_kernels.o", "fwi_stream3d_f32_o8.o", "fwi_stream3d_f32_lo.o", "fwi_stream3d_f64.o", "fwi_fused2d.o",
        "fwi_fused2d_pml.o", "fwi_pml.o", "fwi_pair3d.o", "mc_kernels.o"]
# instantiations per object at the end of round 4 (round 3: fwi_kernels.o alone held 588 stream / tile / point kernels).
# A change here is deliberate: a new template flag doubles a family, and untested instantiations are where spills hide.
EXPECTED_MAX = {"fwi_kernels.o": 172, "fwi_stream3d_f32_o8.o": 208, "fwi_stream3d_f32_lo.o": 96, "fwi_stream3d_f64.o": 96,
                "fwi_fused2d.o": 162, "fwi_fused2d_pml.o": 18, "fwi_pml.o": 128, "fwi_pair3d.o": 3, "mc_kernels.o": 31}


@pytest.fixture(scope="module")
def kernels():
    paths = [os.path.join(co.CSRC, o) for o in OBJS]
    if not co.tools_present() or not all(os.path.exists(p) for p in paths):
        pytest.skip("ROCm LLVM tools or the built objects are missing (run `make -C full_waveform_inversion_amd/csrc`)")
    return co.kernels(paths)


def test_no_kernel_uses_scratch(kernels):
    bad = [(k["obj"], k["name"], k["private_segment_fixed_size"], k.get("vgpr_spill_count", 0)) for k in kernels
           if k.get("private_segment_fixed_size", 0) > 0 or k.get("vgpr_spill_count", 0) > 0]
    # (SGPR spills go to VGPR lanes, not to memory: sgpr_spill_count is not scratch)
    assert not bad, "kernels with scratch / spills:\n" + "\n".join("%s %s: %d B/lane, %d spilled VGPRs" % b for b in bad)


def test_instantiation_counts_do_not_grow_unnoticed(kernels):
    per = {}
    for k in kernels:
        per[k["obj"]] = per.get(k["obj"], 0) + 1
    print("kernels per object:", per, "total", sum(per.values()))
    assert set(per) == set(OBJS), per
    over = {o: (n, EXPECTED_MAX[o]) for o, n in per.items() if n > EXPECTED_MAX[o]}
    assert not over, over
    # the step / tile / point / helper kernels that made up round 3's fwi_kernels.o (588): 572 now, T-term and paired-increment-imaging variants included
    assert sum(per[o] for o in per if o.startswith("fwi_stream3d") or o == "fwi_kernels.o") < 588


def test_stream_kernel_resources_fit_their_launch_bounds(kernels):
    """8-row tiles are 512 threads = 2 waves per SIMD: at most 256 VGPRs (arch + acc) per lane; 4-row tiles 512."""
    for k in kernels:
        if "step3d_stream<" not in k["name"]:
            continue
        ty = int(k["name"].split("step3d_stream<")[1].split(",")[2])
        cap = 256 if ty == 8 else 512
        assert k["vgpr_count"] + k.get("agpr_count", 0) <= cap, k
        assert k["group_segment_fixed_size"] <= 64 * 1024, k
