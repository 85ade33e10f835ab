"""The C-ABI boundary without a GPU: header <-> library <-> binding agree, and
the product path fails loudly (no CPU fallback) when there is no device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from full_waveform_inversion_amd import Engine, FwiError, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fwi.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(fwi_[a-z0-9_]+)\s*\(", txt))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared()
    assert declared, "no declarations parsed from include/fwi.h"
    assert declared == set(_lib.SIGNATURES), "binding table and header disagree"
    for name in declared:
        assert hasattr(lib, name), "libfwi_hip.so does not export %s" % name


def test_abi_version_and_struct_size():
    lib = _lib.load()
    assert lib.fwi_abi_version() == _lib.ABI_VERSION
    # 18 int32 + 4 double, naturally aligned
    assert C.sizeof(_lib.Config) == 18 * 4 + 4 * 8


def test_product_imports_nothing_from_the_oracle():
    pkg = os.path.join(ROOT, "full_waveform_inversion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "libfwi_oracle" not in src, f


def test_create_rejects_bad_configs_before_touching_the_gpu():
    lib = _lib.load()
    ctx = C.c_void_p()
    good = dict(struct_size=C.sizeof(_lib.Config), ndim=3, nz=8, ny=8, nx=8, order=8, nt_max=4, npml=0,
                device=0, dtype=_lib.F32, kernel=_lib.KERNEL_AUTO, zchunk=0, ckpt_interval=0, image_stride=0,
                update_form=0, abc=0, store_dtype=0, launch_mode=0, pml_alpha_max=0.0, h=10.0,
                dt=1e-3,
                sigma_max=0.0)
    for bad in (dict(ndim=4), dict(order=6), dict(nz=0), dict(nt_max=0), dict(h=0.0), dict(dt=-1.0),
                dict(dtype=7), dict(struct_size=8), dict(npml=-1), dict(kernel=9), dict(ckpt_interval=-1), dict(image_stride=-1),
                dict(image_stride=4, ckpt_interval=8), dict(update_form=2), dict(abc=5), dict(store_dtype=3),
                dict(launch_mode=3), dict(store_dtype=1, dtype=_lib.F64), dict(pml_alpha_max=-1.0)):
        cfg = _lib.Config(**{**good, **bad})
        rc = lib.fwi_create(C.byref(cfg), C.byref(ctx))
        assert rc == 1, bad  # FWI_EINVAL
        assert b"fwi_create" in lib.fwi_last_error(None)
        assert not ctx.value


def test_no_device_means_loud_failure_not_fallback():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible; covered by the gpu tests")
    with pytest.raises(FwiError) as ei:  # npml = 0: the context is created eagerly
        Engine((8, 8, 8), 10.0, 1e-3, 4)
    assert ei.value.code == 2 and "no CPU fallback" in str(ei.value)
    e = Engine((8, 8, 8), 10.0, 1e-3, 4, npml=2)  # damping needs c_max: created at set_model
    with pytest.raises(FwiError) as ei:
        e.forward(np.full((8, 8, 8), 2000.0, np.float32), ([[4, 4, 4]], np.ones(4, np.float32)), [[1, 1, 1]])
    assert ei.value.code == 2 and "no CPU fallback" in str(ei.value)


def test_null_context_calls_return_einval():
    lib = _lib.load()
    assert lib.fwi_set_model(None, None) == 1
    assert lib.fwi_gradient(None, 0, None) == 1
    assert lib.fwi_forward(None, 1, 0, None, None, 0, None, 0, None) == 1
    lib.fwi_destroy(None)  # must be a no-op


def test_oracle_is_imported_only_by_test_infrastructure():
    """oracle/ may be used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
    import re
    offenders = []
    for base in ("full_waveform_inversion_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".sh")):
                    text = open(os.path.join(dirpath, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", text, re.M) or "oracle/" in text and f.endswith(".sh"):
                        offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
