"""ctypes binding of the C-ABI in include/fwi.h (libfwi_hip.so).

The reference has no FFI layer; this is the "thin ctypes
C-ABI shim" BASELINE.json's north_star asks for.  There is NO CPU fallback:
if the HIP library is missing or no GPU is visible the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# FWI_HIP_LIB: load another build of the same library (the stamped diagnostic variant, `make stamps`)
LIB_PATH = os.environ.get("FWI_HIP_LIB") or os.path.join(_HERE, "libfwi_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

ABI_VERSION = 13
F32, F64 = 0, 1
KERNEL_AUTO, KERNEL_POINT, KERNEL_STREAM = 0, 1, 2
WRT_VELOCITY, WRT_SLOWNESS2 = 0, 1
UPDATE_FORMS = {"standard": 0, "increment": 1}
ABCS = {"sponge": 0, "cpml": 1}
STORE_DTYPES = {"native": 0, "bf16": 1}
LAUNCH_MODES = {"auto": 0, "stream": 1, "graph": 2}
UNIQUE_ID_BYTES = 128
ERROR_NAMES = {1: "FWI_EINVAL", 2: "FWI_EHIP", 3: "FWI_ESTATE", 4: "FWI_ENOMEM", 5: "FWI_ECOMM"}


class FwiError(RuntimeError):
    """A C-ABI call returned non-zero (the library never exits the process)."""

    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (ERROR_NAMES.get(code, "FWI_E?"), code, message))
        self.code = code


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("ndim", C.c_int32), ("nz", C.c_int32),
                ("ny", C.c_int32), ("nx", C.c_int32), ("order", C.c_int32), ("nt_max", C.c_int32),
                ("npml", C.c_int32), ("device", C.c_int32), ("dtype", C.c_int32),
                ("kernel", C.c_int32), ("zchunk", C.c_int32), ("ckpt_interval", C.c_int32),
                ("image_stride", C.c_int32), ("update_form", C.c_int32), ("abc", C.c_int32),
                ("store_dtype", C.c_int32), ("launch_mode", C.c_int32), ("h", C.c_double), ("dt", C.c_double),
                ("sigma_max", C.c_double), ("pml_alpha_max", C.c_double)]


# name -> (restype, argtypes); every symbol include/fwi.h declares
_P, _I32, _I64, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_double
SIGNATURES = {
    "fwi_abi_version": (C.c_int, []),
    "fwi_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "fwi_destroy": (None, [_P]),
    "fwi_last_error": (C.c_char_p, [_P]),
    "fwi_set_model": (C.c_int, [_P, _P]),
    "fwi_forward": (C.c_int, [_P, _I32, _I32, _P, _P, _I32, _P, _I32, _P]),
    "fwi_forward_spread": (C.c_int, [_P, _I32, _I32, _I32, _P, _P, _P, _P, _I32, _I32, _P, _P, _P, _I32, _P]),
    "fwi_adjoint": (C.c_int, [_P, _P, _I32, _P]),
    "fwi_misfit_l2": (C.c_int, [_P, _P, C.POINTER(_D)]),
    "fwi_gradient": (C.c_int, [_P, _I32, _P]),
    "fwi_gradient_reset": (C.c_int, [_P]),
    "fwi_gradient_add": (C.c_int, [_P, _P]),
    "fwi_dot": (C.c_int, [_P, _P, _P, _I64, C.POINTER(_D)]),
    "fwi_comm_unique_id": (C.c_int, [_P]),
    "fwi_comm_init": (C.c_int, [_P, _I32, _I32, _P]),
    "fwi_allreduce_gradient": (C.c_int, [_P]),
    "fwi_allreduce_f64": (C.c_int, [_P, C.POINTER(_D), _I32]),
    "fwi_allreduce_f64_max": (C.c_int, [_P, C.POINTER(_D), _I32]),
    "fwi_comm_info": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I32)]),
    "fwi_comm_abort": (C.c_int, [_P]),
    "fwi_last_loop_ms": (C.c_int, [_P, C.POINTER(_D)]),
    "fwi_last_host_ms": (C.c_int, [_P, C.POINTER(_D), C.POINTER(_D)]),
    "fwi_placement_info": (C.c_int, [_P, C.POINTER(_D), C.POINTER(_D), C.POINTER(C.c_int64)]),
    "fwi_set_launch_mode": (C.c_int, [_P, _I32]),
    "fwi_synchronize": (C.c_int, [_P]),
    "fwi_check_padding": (C.c_int, [_P, C.POINTER(_I64)]),
    "fwi_kernel_name": (C.c_char_p, [_P]),
    "fwi_device_count": (C.c_int, [C.POINTER(_I32)]),
    "fwi_vec_create": (C.c_int, [_P, _I32]),
    "fwi_vec_upload": (C.c_int, [_P, _I32, _P]),
    "fwi_vec_download": (C.c_int, [_P, _I32, _P]),
    "fwi_vec_copy": (C.c_int, [_P, _I32, _I32]),
    "fwi_vec_axpby": (C.c_int, [_P, _I32, _D, _I32, _D]),
    "fwi_vec_dot": (C.c_int, [_P, _I32, _I32, C.POINTER(_D)]),
    "fwi_vec_absmax": (C.c_int, [_P, _I32, C.POINTER(_D)]),
    "fwi_vec_clip": (C.c_int, [_P, _I32, _D, _D]),
    "fwi_set_model_vec": (C.c_int, [_P, _I32]),
    "fwi_gradient_vec": (C.c_int, [_P, _I32, _I32]),
    "fwi_mc_score": (C.c_int, [_I32, _I32, _I32, _I32, _I64, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P,
                               C.POINTER(_D)]),
    "fwi_mc_forward": (C.c_int, [_I32, _I32, _I32, _I32, _I64, _P, _P, _P]),
    "fwi_mc_invert": (C.c_int, [_I32, _I32, C.c_uint64, _I64, _I64, _D, _I32, _I32, _I32, _P, _P, _I32, _I32, _I32,
                                _P, _P, _P, _P, _P, C.POINTER(_D)]),
    "fwi_mc_sample": (C.c_int, [_I32, _I32, C.c_uint64, _I64, _I64, _D, _P, _P]),
    "fwi_mc_plan_create": (C.c_int, [_I32, _I32, _I32, _I32, _P, _P, _I64, C.POINTER(C.c_void_p)]),
    "fwi_mc_plan_destroy": (None, [_P]),
    "fwi_mc_plan_invert": (C.c_int, [_P, _I32, C.c_uint64, _I64, _I64, _D, _I32, _I32, _I32, _P, _P, _P, _P,
                                     C.POINTER(_D), C.POINTER(_D)]),
    "fwi_mc_plan_score": (C.c_int, [_P, _I64, _P, _I32, _I32, _I32, _P, _P, C.POINTER(_D), C.POINTER(_D)]),
}
MC_INVERSION_TYPES = {"full_mt": 0, "DC": 1, "single_force": 2, "DC_single_force_couple": 3,
                      "DC_single_force_no_coupling": 4, "DC_crack_couple": 5, "single_force_crack_no_coupling": 6}
MC_METRICS = {"VR": 0, "CC": 1, "PCC": 2, "CC-shift": 3, "gau": 4}

_lib = None


def build(force=False):
    """Compile libfwi_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC_DIR, "-s", "-j4"]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return LIB_PATH


def load():
    """Load the library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libfwi_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C full_waveform_inversion_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(lib, name)  # AttributeError here = header/library mismatch
        f.restype = res
        f.argtypes = args
    if lib.fwi_abi_version() != ABI_VERSION:
        raise ImportError("libfwi_hip.so ABI %d != binding ABI %d" % (lib.fwi_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def device_count():
    n = _I32(0)
    load().fwi_device_count(C.byref(n))
    return int(n.value)


def check(ctx, code):
    if code:
        msg = load().fwi_last_error(ctx)
        raise FwiError(code, msg.decode() if msg else "")
