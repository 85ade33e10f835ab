"""Host-side engine: one context per GPU behind the C-ABI (include/fwi.h).

No reference counterpart; entry-point names and argument
meaning follow BASELINE.json's north_star: forward(model, src, rec),
adjoint(residual), gradient().  NumPy is used only to own host buffers.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib

_COEF_ABS_SUM = {2: 4.0, 4: 16.0 / 3.0, 8: 2048.0 / 315.0}


def cfl_dt(c_max, h, ndim, order):
    """Stability limit of the leapfrog scheme, 2 h / (c_max sqrt(D sum|a_k|))."""
    return 2.0 * h / (c_max * math.sqrt(ndim * _COEF_ABS_SUM[order]))


def default_sigma_max(c_max, h, npml, refl=1e-3):
    """Peak damping rate of the quadratic sponge for a target reflection coefficient."""
    return 0.0 if npml <= 0 else 3.0 * c_max * math.log(1.0 / refl) / (2.0 * npml * h)


def ricker(nt, dt, f0, t0=None, dtype=np.float32):
    t0 = 1.5 / f0 if t0 is None else t0
    a = (np.pi * f0 * (np.arange(nt) * dt - t0)) ** 2
    return ((1.0 - 2.0 * a) * np.exp(-a)).astype(dtype)


class Engine:
    """Acoustic forward / adjoint / gradient on one MI355X.

    ``shape`` is (nz, nx) or (nz, ny, nx); ``model`` arrays are velocities in
    m/s of that shape.  Stateful like the C-ABI: ``adjoint`` uses the last
    ``forward``; ``gradient`` returns the sum over all adjoint calls since
    ``reset_gradient``.  ``ckpt_interval = K > 0`` replaces the store of every step's imaging term
    (``nt_max`` model-sized arrays) by wavefield snapshots every K steps plus recomputation.
    ``image_stride = S > 1`` stores and correlates the imaging term every S-th step only (weight S): an
    approximation of the time integral that is accurate while ``S * dt`` still samples the wavelet's
    band, with a store S times smaller and less adjoint traffic.
    ``update_form="increment"`` carries the recursion as (u, v = u - u_prev): same mathematics, ~4x less fp32
    round-off growth, 25 % more traffic.  ``abc="cpml"`` replaces the sponge by a convolutional PML.
    ``store_dtype="bf16"`` halves the forward-term store of an fp32 engine.
    """

    def __init__(self, shape, h, dt, nt_max, order=8, npml=0, sigma_max=None, dtype="float32",
                 device=0, kernel="auto", zchunk=0, ckpt_interval=0, image_stride=1, update_form="standard",
                 abc="sponge", pml_alpha_max=0.0, store_dtype="native", launch_mode="auto"):
        shape = tuple(int(s) for s in shape)
        if len(shape) not in (2, 3):
            raise ValueError("shape must be (nz, nx) or (nz, ny, nx)")
        self.shape, self.ndim = shape, len(shape)
        self.h, self.dt, self.order, self.npml = float(h), float(dt), int(order), int(npml)
        self.nt_max = int(nt_max)
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("dtype must be float32 or float64")
        self.sigma_max = sigma_max
        self.device = int(device)
        self._kernel = {"auto": _lib.KERNEL_AUTO, "point": _lib.KERNEL_POINT,
                        "stream": _lib.KERNEL_STREAM}[kernel]
        self._zchunk = int(zchunk)
        self._ckpt = int(ckpt_interval)
        self._istride = int(image_stride)
        self._update_form = _lib.UPDATE_FORMS[update_form]
        self.update_form = update_form
        self._abc = _lib.ABCS[abc]
        self._store_dtype = _lib.STORE_DTYPES[store_dtype]
        # "graph": each sweep's time loop as one hipGraph; "auto": where that was measured to pay (fwi_config.launch_mode)
        self._launch_mode = _lib.LAUNCH_MODES[launch_mode]
        self.pml_alpha_max = float(pml_alpha_max)
        self._lib = _lib.load()
        self._ctx = None
        self._nsrc = self._nrec = self._nt = 0
        if self.npml == 0 or self.sigma_max is not None:
            self._create(0.0)  # damping does not depend on the model: create the context now

    # -- context ---------------------------------------------------------------
    def _create(self, c_max):
        if self.sigma_max is None:
            self.sigma_max = default_sigma_max(c_max, self.h, self.npml)
        nz, nx = self.shape[0], self.shape[-1]
        ny = self.shape[1] if self.ndim == 3 else 1
        cfg = _lib.Config(C.sizeof(_lib.Config), self.ndim, nz, ny, nx, self.order, self.nt_max,
                          self.npml, self.device,
                          _lib.F32 if self.dtype == np.float32 else _lib.F64, self._kernel,
                          self._zchunk, self._ckpt, self._istride, self._update_form, self._abc,
                          self._store_dtype, self._launch_mode, self.h, self.dt, float(self.sigma_max),
                          self.pml_alpha_max)
        ctx = C.c_void_p()
        _lib.check(None, self._lib.fwi_create(C.byref(cfg), C.byref(ctx)))
        self._ctx = ctx

    def close(self):
        if self._ctx is not None:
            self._lib.fwi_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, code):
        _lib.check(self._ctx, code)

    @property
    def _c(self):
        """The context handle, or a clear error when it does not exist yet (with ``npml > 0`` and no
        ``sigma_max`` the context is created by the first ``set_model``, which knows ``c_max``)."""
        if self._ctx is None:
            raise _lib.FwiError(3, "no context yet: call set_model() first (npml > 0 without sigma_max defers "
                                   "fwi_create until the model's maximum velocity is known)")
        return self._ctx

    def _host(self, a, shape=None):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if shape is not None and a.shape != tuple(shape):
            raise ValueError("expected shape %s, got %s" % (tuple(shape), a.shape))
        return a

    def _idx(self, idx):
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int32).reshape(-1, self.ndim))
        return idx

    @property
    def kernel_name(self):
        return self._lib.fwi_kernel_name(self._ctx).decode() if self._ctx is not None else ""

    # -- the three entry points ------------------------------------------------
    def set_model(self, model):
        model = self._host(model, self.shape)
        if self._ctx is None:
            self._create(float(model.max()))
        self._chk(self._lib.fwi_set_model(self._ctx, model.ctypes.data_as(C.c_void_p)))

    def forward(self, model, src, rec, save=True):
        """Seismograms ``(nt, nrec)``.  ``src = (src_idx (nsrc, ndim), wavelet (nt[, nsrc]))``."""
        if model is not None:
            self.set_model(model)
        if self._ctx is None:
            raise _lib.FwiError(3, "forward: no model set")
        src_idx, wavelet = src
        src_idx = self._idx(src_idx)
        rec_idx = self._idx(rec)
        wavelet = np.asarray(wavelet)
        if wavelet.ndim == 1:
            wavelet = wavelet[:, None]
        wavelet = self._host(wavelet)
        nt = wavelet.shape[0]
        if wavelet.shape[1] != len(src_idx):
            raise ValueError("wavelet must be (nt, nsrc)")
        seis = np.zeros((nt, len(rec_idx)), self.dtype)
        self._chk(self._lib.fwi_forward(
            self._ctx, nt, len(src_idx), src_idx.ctypes.data_as(C.c_void_p),
            wavelet.ctypes.data_as(C.c_void_p), len(rec_idx), rec_idx.ctypes.data_as(C.c_void_p),
            int(bool(save)), seis.ctypes.data_as(C.c_void_p)))
        self._nt, self._nsrc, self._nrec = nt, len(src_idx), len(rec_idx)
        return seis

    def forward_at(self, model, src, rec, save=True):
        """``forward`` with sources / receivers at fractional grid coordinates (cells; multilinear interpolation):
        ``src = (src_xyz (nsrc, ndim) | points.Spread, wavelet (nt[, nsrc]))``, ``rec = rec_xyz | Spread``.
        The per-point series are spread onto / gathered from the grid nodes ON THE DEVICE (``fwi_forward_spread``);
        the following ``adjoint`` / ``misfit_l2`` then work per point as well."""
        from .points import Spread
        if model is not None:
            self.set_model(model)
        if self._ctx is None:
            raise _lib.FwiError(3, "forward_at: no model set")
        S, wavelet = src
        S = S if isinstance(S, Spread) else Spread(S, self.shape)
        R = rec if isinstance(rec, Spread) else Spread(rec, self.shape)
        wavelet = np.asarray(wavelet)
        if wavelet.ndim == 1:
            wavelet = wavelet[:, None]
        wavelet = self._host(wavelet)
        nt = wavelet.shape[0]
        if wavelet.shape[1] != S.n:
            raise ValueError("wavelet must be (nt, nsrc points)")
        sw, rw = self._host(S.weights), self._host(R.weights)
        seis = np.zeros((nt, R.n), self.dtype)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        self._chk(self._lib.fwi_forward_spread(
            self._ctx, nt, S.n, len(S.idx), vp(S.idx), vp(S.pt_start), vp(sw), vp(wavelet), R.n, len(R.idx),
            vp(R.idx), vp(R.pt_start), vp(rw), int(bool(save)), vp(seis)))
        self._nt, self._nsrc, self._nrec = nt, S.n, R.n
        return seis

    def adjoint(self, residual, image=True):
        """Back-propagate ``residual (nt, nrec)``; returns ``F^T residual`` as ``(nt, nsrc)``.
        ``residual=None``: the residual :meth:`misfit_l2` formed on the device."""
        if self._ctx is None:
            raise _lib.FwiError(3, "adjoint: no forward run to adjoin")
        rp = None
        if residual is not None:
            residual = self._host(residual, (self._nt, self._nrec))
            rp = residual.ctypes.data_as(C.c_void_p)
        out = np.zeros((self._nt, self._nsrc), self.dtype)
        self._chk(self._lib.fwi_adjoint(self._ctx, rp, int(bool(image)), out.ctypes.data_as(C.c_void_p)))
        return out

    def misfit_l2(self, d_obs):
        """``J = 1/2 ||d_syn - d_obs||^2`` for the last forward's seismograms, residual formed and reduced on
        the device and kept there for ``adjoint(None)``."""
        d_obs = self._host(d_obs, (self._nt, self._nrec))
        J = C.c_double(0.0)
        self._chk(self._lib.fwi_misfit_l2(self._c, d_obs.ctypes.data_as(C.c_void_p), C.byref(J)))
        return J.value

    def gradient(self, wrt="velocity"):
        if self._ctx is None:
            raise _lib.FwiError(3, "gradient: no model set")
        g = np.zeros(self.shape, self.dtype)
        w = {"velocity": _lib.WRT_VELOCITY, "slowness2": _lib.WRT_SLOWNESS2}[wrt]
        self._chk(self._lib.fwi_gradient(self._ctx, w, g.ctypes.data_as(C.c_void_p)))
        return g

    def reset_gradient(self):
        if self._ctx is not None:
            self._chk(self._lib.fwi_gradient_reset(self._ctx))

    def gradient_add_from(self, other):
        """This engine's gradient accumulator += ``other``'s (same shape, same GPU)."""
        self._chk(self._lib.fwi_gradient_add(self._ctx, other._ctx))

    # -- reductions, exchange, measurement --------------------------------------
    def dot(self, a, b):
        a, b = self._host(a).ravel(), self._host(b).ravel()
        if a.size != b.size:
            raise ValueError("dot: size mismatch")
        out = C.c_double(0.0)
        self._chk(self._lib.fwi_dot(self._c, a.ctypes.data_as(C.c_void_p),
                                    b.ctypes.data_as(C.c_void_p), a.size, C.byref(out)))
        return out.value

    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        _lib.check(None, _lib.load().fwi_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank, nranks, unique_id):
        if len(unique_id) != _lib.UNIQUE_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % _lib.UNIQUE_ID_BYTES)
        self._chk(self._lib.fwi_comm_init(self._c, rank, nranks, C.c_char_p(unique_id)))

    def allreduce_gradient(self):
        self._chk(self._lib.fwi_allreduce_gradient(self._c))

    def allreduce_f64(self, vals, op="sum"):
        arr = (C.c_double * len(vals))(*vals)
        fn = {"sum": self._lib.fwi_allreduce_f64, "max": self._lib.fwi_allreduce_f64_max}[op]
        self._chk(fn(self._c, arr, len(vals)))
        return list(arr)

    def comm_info(self):
        """(nranks, rank) as RCCL reports them for this context's communicator."""
        n, r = C.c_int32(0), C.c_int32(-1)
        self._chk(self._lib.fwi_comm_info(self._c, C.byref(n), C.byref(r)))
        return int(n.value), int(r.value)

    def comm_abort(self):
        if self._ctx is not None:
            self._chk(self._lib.fwi_comm_abort(self._ctx))

    # -- device-resident model-shaped vectors (optimiser state) ------------------------
    def vec_create(self, count):
        self._chk(self._lib.fwi_vec_create(self._c, int(count)))

    def vec_upload(self, slot, a):
        a = self._host(a, self.shape)
        self._chk(self._lib.fwi_vec_upload(self._c, slot, a.ctypes.data_as(C.c_void_p)))

    def vec_download(self, slot):
        out = np.empty(self.shape, self.dtype)
        self._chk(self._lib.fwi_vec_download(self._c, slot, out.ctypes.data_as(C.c_void_p)))
        return out

    def vec_copy(self, dst, src):
        self._chk(self._lib.fwi_vec_copy(self._c, dst, src))

    def vec_axpby(self, y, a, x, b=1.0):
        """y = a * x + b * y on the device."""
        self._chk(self._lib.fwi_vec_axpby(self._c, y, float(a), x, float(b)))

    def vec_dot(self, x, y):
        out = C.c_double(0.0)
        self._chk(self._lib.fwi_vec_dot(self._c, x, y, C.byref(out)))
        return out.value

    def vec_absmax(self, x):
        out = C.c_double(0.0)
        self._chk(self._lib.fwi_vec_absmax(self._c, x, C.byref(out)))
        return out.value

    def vec_clip(self, x, lo, hi):
        self._chk(self._lib.fwi_vec_clip(self._c, x, float(lo), float(hi)))

    def set_model_vec(self, slot):
        self._chk(self._lib.fwi_set_model_vec(self._c, slot))

    def gradient_vec(self, slot, wrt="velocity"):
        w = {"velocity": _lib.WRT_VELOCITY, "slowness2": _lib.WRT_SLOWNESS2}[wrt]
        self._chk(self._lib.fwi_gradient_vec(self._c, w, slot))

    def last_loop_ms(self):
        ms = C.c_double(0.0)
        self._chk(self._lib.fwi_last_loop_ms(self._c, C.byref(ms)))
        return ms.value

    def last_host_ms(self):
        """(submit_ms, graph_build_ms): host wall time spent forming and submitting the last time loop's launches, and
        the hipGraph capture + instantiation share of it (0 with ``launch_mode="stream"``)."""
        a, b = C.c_double(0.0), C.c_double(0.0)
        self._chk(self._lib.fwi_last_host_ms(self._c, C.byref(a), C.byref(b)))
        return a.value, b.value

    def placement_info(self):
        """(us_before, us_after, shifts) of the placement search a 3-D CPML or increment-form context runs at creation
        (include/fwi.h fwi_placement_info; shifts = bytes of the movable arrays in search order, then zeros); all zeros
        when the context ran none."""
        a, b = C.c_double(0.0), C.c_double(0.0)
        sh = (C.c_int64 * 8)()
        self._chk(self._lib.fwi_placement_info(self._c, C.byref(a), C.byref(b), sh))
        return a.value, b.value, tuple(int(v) for v in sh)

    def set_launch_mode(self, mode):
        """"auto" / "stream" / "graph" from the next sweep on (an A/B on one context: same buffers, same cache state)."""
        self._chk(self._lib.fwi_set_launch_mode(self._c, _lib.LAUNCH_MODES[mode]))

    def synchronize(self):
        self._chk(self._lib.fwi_synchronize(self._c))

    def dirty_padding(self):
        """Cells of the padded device fields outside the grid's interior that are not exactly zero (must be 0)."""
        n = C.c_int64(-1)
        self._chk(self._lib.fwi_check_padding(self._c, C.byref(n)))
        return int(n.value)
