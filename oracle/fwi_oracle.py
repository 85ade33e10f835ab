"""CPU oracle for the acoustic forward / adjoint / gradient path (NumPy, fp64).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker.  The product path
(``full_waveform_inversion_amd``) never imports it and has no CPU fallback.

PARITY UNPINNED.  The reference (/root/reference/full_waveform_inversion.py)
holds no wave-propagation code at all: no grid, stencil, absorbing boundary,
adjoint, imaging condition or gradient (its only "forward"
is ``forward_model`` at full_waveform_inversion.py:253-264, a contraction of
precomputed Green's functions with a <=9-vector).  There is therefore no
reference file:line this restatement can follow, no golden vector and no
known-answer test to pin it.  The scheme below is BUILD-DEFINED (SURVEY.md
section 7.0) and is validated by its own ladder in tests/test_oracle.py:
analytic point-source solutions, the adjoint dot-product identity to 1e-12 and
finite-difference gradient checks.  Reports must say "vs the build's own NumPy
oracle", never "matches the reference".

Scheme (constant density acoustic, 2nd order in time, order 2r in space)
-----------------------------------------------------------------------
Grid ``(nz, nx)`` or ``(nz, ny, nx)``, C order, x fastest, spacing ``h``;
field is zero outside the grid (Dirichlet halo).  Model = velocity ``c``;
``C = dt^2 c^2``.  Damping ``d = sigma dt / 2 >= 0`` (sum of per-axis quadratic
ramps over an ``npml``-cell border), ``A = 1/(1+d)``, ``B = 1-d``.

    q^n     = C * (L u^n + P w^n / h^D)            (L: undivided star / h^2)
    u^{n+1} = A * (2 u^n - B u^{n-1} + q^n),       u^0 = u^{-1} = 0
    d^n     = R u^{n+1},                           n = 0 .. nt-1

Adjoint (exact transpose of the above), j = nt-1 .. 0:

    mu^{j+1} = A * (2 mu^{j+2} - B mu^{j+3} + C * (L mu^{j+2} + R^T r^j))
    a^j      = P^T mu^{j+1} / h^D                  (= (F^T r)^j)

Second absorbing boundary, ``abc="cpml"`` (convolutional PML, memory variables psi_d, zeta_d per axis d, non-zero
in that axis' ``npml`` border only; the sponge is off: A = B = 1).  With D_d the centred first difference of the
same order and E_d the second-difference star along d (L = sum_d E_d), per-axis 1-D coefficients
``b = exp(-(sigma + alpha) dt)``, ``a = sigma / (sigma + alpha) (b - 1)``, ``sigma = sigma_max (dist/npml)^2``,
``alpha = alpha_max (1 - dist/npml)``:

    psi_d  <- b psi_d  + a D_d u^n
    zeta_d <- b zeta_d + a (E_d u^n + D_d psi_d)
    q^n     = C * (L u^n + sum_d (D_d psi_d + zeta_d) + P w^n / h^D),   u^{n+1} = 2 u^n - u^{n-1} + q^n

Its exact transpose, in the scaled adjoint variable mu = C * (adjoint of u) that makes the main recursion the
forward one again (D_d is antisymmetric, E_d symmetric on the zero-extended grid), j = nt-1 .. 0:

    zt_d <- b zt_d + mu^{j+2},                         alpha_d = a zt_d
    pt_d <- b pt_d - D_d mu^{j+2} - D_d alpha_d,       beta_d  = a pt_d
    mu^{j+1} = 2 mu^{j+2} - mu^{j+3} + C * (L mu^{j+2} + sum_d (E_d alpha_d - D_d beta_d) + R^T r^j)

The gradient formula below is unchanged (the model enters through C only; q^n includes the PML terms).

Gradient of ``J = 1/2 sum (d - d_obs)^2`` w.r.t. squared slowness m = 1/c^2
(zero-lag correlation of the adjoint field with the stored forward term q):

    g_m = -(1/dt^2) sum_n mu^{n+1} q^n ,     g_c = g_m * (-2 / c^3)
"""
from __future__ import annotations

import numpy as np

# Centred second-derivative weights a_0, a_1 .. a_r for order 2r.
COEFFS = {
    2: (-2.0, 1.0),
    4: (-5.0 / 2.0, 4.0 / 3.0, -1.0 / 12.0),
    8: (-205.0 / 72.0, 8.0 / 5.0, -1.0 / 5.0, 8.0 / 315.0, -1.0 / 560.0),
}


def ricker(nt, dt, f0, t0=None, dtype=np.float64):
    """Ricker wavelet with peak frequency ``f0`` delayed by ``t0`` (default 1.5/f0)."""
    if t0 is None:
        t0 = 1.5 / f0
    t = np.arange(nt, dtype=np.float64) * dt - t0
    a = (np.pi * f0 * t) ** 2
    return ((1.0 - 2.0 * a) * np.exp(-a)).astype(dtype)


def cfl_dt(c_max, h, ndim, order):
    """Largest stable dt of the leapfrog scheme: 2 / (c_max sqrt(D sum|a|) / h)."""
    s = sum(abs(a) for a in COEFFS[order]) + sum(abs(a) for a in COEFFS[order][1:])
    return 2.0 * h / (c_max * np.sqrt(ndim * s))


def default_sigma_max(c_max, h, npml, refl=1e-3):
    """Peak damping rate (1/s) of the quadratic sponge for a target reflection."""
    if npml <= 0:
        return 0.0
    return 3.0 * c_max * np.log(1.0 / refl) / (2.0 * npml * h)


def damping_profiles(shape, npml, sigma_max, dt):
    """Per-axis 1-D profiles of d = sigma dt / 2; the field profile is their sum."""
    out = []
    for n in shape:
        i = np.arange(n, dtype=np.float64)
        if npml > 0:
            dist = np.maximum(0.0, np.maximum(npml - i, i - (n - 1 - npml)))
            out.append(0.5 * dt * sigma_max * (dist / npml) ** 2)
        else:
            out.append(np.zeros(n))
    return out


# Centred first-difference weights d_1 .. d_r for order 2r (D u = sum_k d_k (u[+k] - u[-k]) / h).
DCOEFFS = {
    2: (1.0 / 2.0,),
    4: (2.0 / 3.0, -1.0 / 12.0),
    8: (4.0 / 5.0, -1.0 / 5.0, 4.0 / 105.0, -1.0 / 280.0),
}


def bf16_round(x):
    """fp32 -> bf16 (round to nearest even) -> back, element-wise; returned in the input's dtype."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return u.view(np.float32).astype(np.asarray(x).dtype)


def cpml_profiles(shape, npml, sigma_max, alpha_max, dt):
    """Per-axis 1-D CPML coefficients (a, b); a = 0 outside the npml border."""
    out = []
    for n in shape:
        i = np.arange(n, dtype=np.float64)
        dist = np.maximum(0.0, np.maximum(npml - i, i - (n - 1 - npml))) if npml > 0 else np.zeros(n)
        x = dist / max(npml, 1)
        sig = sigma_max * x ** 2
        alp = alpha_max * (1.0 - x)
        b = np.exp(-(sig + alp) * dt)
        with np.errstate(invalid="ignore", divide="ignore"):
            a = np.where(sig > 0.0, sig / (sig + alp) * (b - 1.0), 0.0)
        out.append((a, b))
    return out


def _ravel_idx(idx, shape):
    idx = np.asarray(idx, dtype=np.int64).reshape(-1, len(shape))
    for a, n in enumerate(shape):
        if idx.size and (idx[:, a].min() < 0 or idx[:, a].max() >= n):
            raise ValueError("index outside the grid")
    return np.ravel_multi_index(tuple(idx.T), shape) if idx.size else np.zeros(0, np.int64)


class Propagator:
    """fp64 (or any ``dtype``) restatement of one shot's forward/adjoint/gradient."""

    def __init__(self, c, h, dt, order=8, npml=0, sigma_max=None, dtype=np.float64, image_stride=1,
                 abc="sponge", pml_alpha_max=0.0, store_dtype="native"):
        c = np.asarray(c, dtype=np.float64)
        # image_stride S > 1: the imaging condition is a Riemann sum over every S-th step,
        # img = S * sum_{n % S == 0} mu^{n+1} q^n  (the engine's fwi_config.image_stride)
        self.image_stride = max(1, int(image_stride))
        if c.ndim not in (2, 3):
            raise ValueError("model must be 2-D (nz,nx) or 3-D (nz,ny,nx)")
        if order not in COEFFS:
            raise ValueError("order must be one of %s" % sorted(COEFFS))
        self.c = c
        self.shape = c.shape
        self.ndim = c.ndim
        self.h = float(h)
        self.dt = float(dt)
        self.order = order
        self.r = order // 2
        self.npml = int(npml)
        self.dtype = np.dtype(dtype)
        if sigma_max is None:
            sigma_max = default_sigma_max(c.max(), h, npml)
        self.sigma_max = float(sigma_max)
        # store_dtype="bf16" restates the engine's fwi_config.store_dtype: the stored forward term is C L u rounded
        # to bf16 (from fp32, nearest even); the source's own share of it is kept exact
        self.store_bf16 = store_dtype == "bf16"
        if abc not in ("sponge", "cpml"):
            raise ValueError("abc must be 'sponge' or 'cpml'")
        self.abc = abc
        self.pml_alpha_max = float(pml_alpha_max)
        prof = damping_profiles(self.shape, self.npml if abc == "sponge" else 0, self.sigma_max, self.dt)
        self.cpml = None
        if abc == "cpml" and self.npml > 0:
            self.cpml = []
            for ax, (a, b) in enumerate(cpml_profiles(self.shape, self.npml, self.sigma_max, self.pml_alpha_max,
                                                      self.dt)):
                sh = [1] * self.ndim
                sh[ax] = -1
                self.cpml.append((a.reshape(sh).astype(self.dtype), b.reshape(sh).astype(self.dtype)))
            self.dcoef = [self.dtype.type(dk / self.h) for dk in DCOEFFS[order]]
        d = np.zeros(self.shape)
        for a, p in enumerate(prof):
            sh = [1] * self.ndim
            sh[a] = -1
            d = d + p.reshape(sh)
        self.profiles = prof
        self.A = (1.0 / (1.0 + d)).astype(self.dtype)
        self.B = (1.0 - d).astype(self.dtype)
        self.C = (self.dt ** 2 * c ** 2).astype(self.dtype)
        self.coef = [self.dtype.type(a / self.h ** 2) for a in COEFFS[order]]
        self.q_store = None
        self._last = None

    # -- the stencil ---------------------------------------------------------
    def laplacian(self, u):
        r = self.r
        p = np.pad(u, r)
        core = tuple(slice(r, r + n) for n in self.shape)
        lap = (self.ndim * self.coef[0]) * u
        for k in range(1, r + 1):
            acc = None
            for a in range(self.ndim):
                lo = list(core)
                hi = list(core)
                lo[a] = slice(r - k, r - k + self.shape[a])
                hi[a] = slice(r + k, r + k + self.shape[a])
                t = p[tuple(lo)] + p[tuple(hi)]
                acc = t if acc is None else acc + t
            lap = lap + self.coef[k] * acc
        return lap

    def _d1(self, u, ax):
        """Centred first difference along axis ``ax`` (zero outside the grid): antisymmetric."""
        r = self.r
        pad = [(0, 0)] * self.ndim
        pad[ax] = (r, r)
        p = np.pad(u, pad)
        n = self.shape[ax]
        out = np.zeros(self.shape, self.dtype)
        for k in range(1, r + 1):
            hi = [slice(None)] * self.ndim
            lo = [slice(None)] * self.ndim
            hi[ax] = slice(r + k, r + k + n)
            lo[ax] = slice(r - k, r - k + n)
            out = out + self.dcoef[k - 1] * (p[tuple(hi)] - p[tuple(lo)])
        return out

    def _d2(self, u, ax):
        """Second-difference star along axis ``ax`` (the Laplacian is the sum of these over the axes)."""
        r = self.r
        pad = [(0, 0)] * self.ndim
        pad[ax] = (r, r)
        p = np.pad(u, pad)
        n = self.shape[ax]
        out = self.coef[0] * u
        for k in range(1, r + 1):
            hi = [slice(None)] * self.ndim
            lo = [slice(None)] * self.ndim
            hi[ax] = slice(r + k, r + k + n)
            lo[ax] = slice(r - k, r - k + n)
            out = out + self.coef[k] * (p[tuple(hi)] + p[tuple(lo)])
        return out

    def _cpml_term(self, u, aux, reverse):
        """Advance the memory variables one step with the newest field ``u`` and return their contribution to
        the bracket of q (forward recursion, or its transpose for the adjoint sweep)."""
        term = np.zeros(self.shape, self.dtype)
        for ax, (a, b) in enumerate(self.cpml):
            p_, z_ = aux[ax]
            if not reverse:
                p_ = b * p_ + a * self._d1(u, ax)
                z_ = b * z_ + a * (self._d2(u, ax) + self._d1(p_, ax))
                term = term + self._d1(p_, ax) + z_
            else:
                z_ = b * z_ + u
                al = a * z_
                p_ = b * p_ - self._d1(u, ax) - self._d1(al, ax)
                term = term + self._d2(al, ax) - self._d1(a * p_, ax)
            aux[ax] = (p_, z_)
        return term

    def _propagate(self, inj_flat, inj_amp, inj_scale, rec_flat, rec_scale, nt,
                   reverse, save_q=False, image_q=None):
        """Shared time loop: inject ``inj_amp[n]`` at ``inj_flat``, record at ``rec_flat``."""
        dt_ = self.dtype
        u_prev = np.zeros(self.shape, dt_)
        u_cur = np.zeros(self.shape, dt_)
        rec = np.zeros((nt, len(rec_flat)), dt_)
        qs = np.zeros((nt,) + self.shape, dt_) if save_q else None
        img = np.zeros(self.shape, np.float64) if image_q is not None else None
        steps = range(nt - 1, -1, -1) if reverse else range(nt)
        aux = [(np.zeros(self.shape, dt_), np.zeros(self.shape, dt_)) for _ in range(self.ndim)] if self.cpml else None
        for n in steps:
            src = np.zeros(self.shape, dt_)
            if len(inj_flat):
                np.add.at(src.reshape(-1), inj_flat, inj_amp[n] * dt_.type(inj_scale))
            extra = self._cpml_term(u_cur, aux, reverse) if self.cpml else 0.0
            lap = self.laplacian(u_cur) + extra
            q = self.C * (lap + src)
            u_next = self.A * (2 * u_cur - self.B * u_prev + q)
            if save_q:
                qs[n] = (bf16_round(self.C * lap) + self.C * src) if self.store_bf16 else q
            if img is not None and n % self.image_stride == 0:
                img += self.image_stride * (u_next.astype(np.float64) * image_q[n])
            rec[n] = u_next.reshape(-1)[rec_flat] * dt_.type(rec_scale)
            u_prev, u_cur = u_cur, u_next
        self._last = (u_prev, u_cur)
        return rec, qs, img

    # -- entry points --------------------------------------------------------
    def forward(self, src_idx, wavelet, rec_idx, save=True):
        """Seismograms ``(nt, nrec)`` for sources at ``src_idx`` with ``wavelet (nt, nsrc)``."""
        wavelet = np.asarray(wavelet, self.dtype)
        if wavelet.ndim == 1:
            wavelet = wavelet[:, None]
        self.src_flat = _ravel_idx(src_idx, self.shape)
        self.rec_flat = _ravel_idx(rec_idx, self.shape)
        if wavelet.shape[1] != len(self.src_flat):
            raise ValueError("wavelet must be (nt, nsrc)")
        self.nt = wavelet.shape[0]
        d, qs, _ = self._propagate(self.src_flat, wavelet, 1.0 / self.h ** self.ndim,
                                   self.rec_flat, 1.0, self.nt, False, save_q=save)
        self.q_store = qs
        return d

    def adjoint(self, residual, image=True):
        """Back-propagate ``residual (nt, nrec)``; returns ``F^T r`` as ``(nt, nsrc)``.

        With ``image`` the zero-lag correlation with the stored forward term is
        accumulated for :meth:`gradient`.
        """
        residual = np.asarray(residual, self.dtype)
        if residual.shape != (self.nt, len(self.rec_flat)):
            raise ValueError("residual must be (nt, nrec) of the last forward")
        if image and self.q_store is None:
            raise RuntimeError("forward(..., save=True) must precede adjoint(image=True)")
        a, _, img = self._propagate(self.rec_flat, residual, 1.0, self.src_flat,
                                    1.0 / self.h ** self.ndim, self.nt, True,
                                    image_q=self.q_store if image else None)
        self._img = img
        return a

    def gradient(self, wrt="velocity"):
        """dJ/dc (``wrt='velocity'``) or dJ/dm, m = 1/c^2 (``wrt='slowness2'``)."""
        g_m = -self._img / self.dt ** 2
        if wrt == "slowness2":
            return g_m
        if wrt == "velocity":
            return g_m * (-2.0 / self.c ** 3)
        raise ValueError("wrt must be 'velocity' or 'slowness2'")


def misfit_and_gradient(c, h, dt, order, npml, src_idx, wavelet, rec_idx, d_obs,
                        sigma_max=None, wrt="velocity"):
    """J = 1/2 ||F(c) - d_obs||^2 and its gradient for one shot."""
    p = Propagator(c, h, dt, order, npml, sigma_max)
    d = p.forward(src_idx, wavelet, rec_idx, save=True)
    r = d - d_obs
    p.adjoint(r)
    return 0.5 * float(np.sum(r * r)), p.gradient(wrt), d
