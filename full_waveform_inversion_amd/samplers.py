"""The reference's seven random source samplers (full_waveform_inversion.py:282-510), batched.

Each sampler of the reference draws a handful of random deviates and then applies a fixed,
deterministic map to them.  Here the two halves are separated:

* ``*_from_deviates`` functions are the deterministic maps, vectorised over ``N`` samples
  (``(N, ...)`` deviates in, ``(n, N)`` source vectors out in the reference's ``MTs[:, i]`` layout).
  They keep the reference's arithmetic, including its quirks: the
  "divide by norm**-1" followed by a re-normalisation, ``arctan2`` angles everywhere except the
  single-force-crack sampler, which uses ``arccos`` (azimuth only in [0, pi]), and the crack lune
  angle built from ``phi in {0, pi/3}`` plus a quadrant shuffle.
* ``draw_deviates`` produces the deviates either from a ``numpy.random.Generator`` (fast, bulk) or --
  ``reference_stream=True`` -- from the global ``numpy.random`` / ``random`` generators in exactly
  the order the reference's per-sample loop consumes them, so that after the same
  ``np.random.seed(s); random.seed(s)`` the samples equal the reference's to round-off
  (tests/test_samplers.py against tests/golden/ref_samplers.npz).

``draw(inversion_type, N, ...)`` combines both and returns ``(MTs (n, N), amp_frac (N,) or None)``.
"""
from __future__ import annotations

import random as _stdlib_random

import numpy as np

INVERSION_TYPES = ("full_mt", "DC", "single_force", "DC_single_force_couple", "DC_single_force_no_coupling",
                   "DC_crack_couple", "single_force_crack_no_coupling")
# types whose sampler also returns an amplitude fraction, appended to MTs as an extra row (:852-853)
COUPLED_TYPES = INVERSION_TYPES[3:]
# number of source components (rows of the Green's function array) per type
NUM_COMPONENTS = {"full_mt": 6, "DC": 6, "single_force": 3, "DC_single_force_couple": 9,
                  "DC_single_force_no_coupling": 9, "DC_crack_couple": 6, "single_force_crack_no_coupling": 9}

_SQRT2 = np.sqrt(2.0)
_DC_MT = np.array([[0.0, 0.0, 1.0], [0.0, 0.0, 0.0], [1.0, 0.0, 0.0]])  # :299


def _unit_rows(a):
    """The reference's two-step normalisation of each row of ``a`` (:288-290 and the same lines in
    every sampler): ``a / (sum a^2)**-0.5`` (a multiplication by the norm), then division by the
    norm of the result."""
    a = np.asarray(a, dtype=float)
    a1 = a / (np.sum(a ** 2, axis=1, keepdims=True) ** -0.5)
    return a1 / (np.sum(a1 ** 2, axis=1, keepdims=True) ** 0.5)


def _angles_arctan2(a):
    """theta, phi of unit vectors as :308-309 / :350-351 / :436-437."""
    x, y, z = a[:, 0], a[:, 1], a[:, 2]
    return np.arctan2(np.sqrt(x ** 2 + y ** 2), z), np.arctan2(y, x)


def _angles_arccos(a):
    """theta, phi as the single-force-crack sampler takes them (:497-498): phi only spans [0, pi]."""
    x, z = a[:, 0], a[:, 2]
    theta = np.arccos(z)
    with np.errstate(invalid="ignore", divide="ignore"):
        phi = np.arccos(x / np.sin(theta))
    return theta, phi


def _rotations(theta, phi):
    """(N, 3, 3) matrices R_theta (about Y) and R_phi (about Z) of rot_mt_by_theta_phi (:226-232)."""
    n = theta.shape[0]
    ct, st, cp, sp = np.cos(theta), np.sin(theta), np.cos(phi), np.sin(phi)
    rt = np.zeros((n, 3, 3))
    rt[:, 0, 0], rt[:, 0, 2], rt[:, 1, 1], rt[:, 2, 0], rt[:, 2, 2] = ct, st, 1.0, -1.0 * st, ct
    rp = np.zeros((n, 3, 3))
    rp[:, 0, 0], rp[:, 0, 1], rp[:, 1, 0], rp[:, 1, 1], rp[:, 2, 2] = cp, -1.0 * sp, sp, cp, 1.0
    return rt, rp


def _rot_mt(full_mt, theta, phi):
    """rot_mt_by_theta_phi (:226-232) for (N, 3, 3) or one (3, 3) tensor."""
    rt, rp = _rotations(theta, phi)
    first = rt @ (full_mt @ np.transpose(rt, (0, 2, 1)))
    return rp @ (first @ np.transpose(rp, (0, 2, 1)))


def _rot_force(vec, theta, phi):
    """rot_single_force_by_theta_phi (:234-240) for one 3-vector, N rotations -> (N, 3)."""
    rt, rp = _rotations(theta, phi)
    return np.einsum("nij,nj->ni", rp, np.einsum("nij,j->ni", rt, vec))


def _six_from_full(full):
    """get_six_MT_from_full_MT_array (:206-208), (N, 3, 3) -> (N, 6)."""
    return np.stack([full[:, 0, 0], full[:, 1, 1], full[:, 2, 2], _SQRT2 * full[:, 0, 1], _SQRT2 * full[:, 0, 2],
                     _SQRT2 * full[:, 1, 2]], axis=1)


def _normalise_six(six):
    return six / (np.sum(six ** 2, axis=1, keepdims=True) ** 0.5)


def _crack_tensor(u_theta, r_phi, r_quadrant):
    """Crack moment tensor on the lune boundary (:393-424 and :457-488): ``u_theta`` is the
    ``np.random.uniform(-1, 1)`` draw, ``r_phi`` / ``r_quadrant`` the two ``random.random()`` draws."""
    theta_l = np.asarray(u_theta, dtype=float) * np.pi / 2.0
    phi_l = np.where(np.asarray(r_phi) <= 0.5, 0.0, np.pi / 3)
    with np.errstate(invalid="ignore", divide="ignore"):
        ang = np.arctan(np.sin(phi_l) / np.sin(theta_l))
    r = np.asarray(r_quadrant, dtype=float)
    ang = np.where((r > 0.25) & (r <= 0.5), ang + np.pi, ang)
    ang = np.where((r > 0.5) & (r <= 0.75), ang + np.pi / 2, ang)
    ang = np.where((r > 0.75) & (r <= 1.0), ang + 3 * np.pi / 2, ang)
    s, c = np.sin(ang), np.cos(ang)
    scale = (((4 * (s ** 2)) + (c ** 2)) ** -0.5) / np.sqrt(3.0)
    crack = np.zeros((ang.shape[0], 3, 3))
    crack[:, 0, 0] = crack[:, 1, 1] = c - (np.sqrt(2) * s)
    crack[:, 2, 2] = c + (2.0 * np.sqrt(2) * s)
    return scale[:, None, None] * crack


# ---------------------------------------------------------------------------------------------
# deterministic maps: deviates -> samples
# ---------------------------------------------------------------------------------------------
def full_mt_from_deviates(z6):
    """generate_random_MT (:282-293): ``z6 (N, 6)`` normal deviates -> ``(6, N)`` unit tensors."""
    return np.ascontiguousarray(_unit_rows(z6).T)


def single_force_from_deviates(z3):
    """generate_random_single_force_vector (:320-331): ``z3 (N, 3)`` -> ``(3, N)``."""
    return np.ascontiguousarray(_unit_rows(z3).T)


def dc_from_deviates(z3):
    """generate_random_DC_MT (:295-317): a double couple rotated to a random orientation."""
    theta, phi = _angles_arctan2(_unit_rows(z3))
    six = _six_from_full(_rot_mt(_DC_MT, theta, phi))
    return np.ascontiguousarray(_normalise_six(six).T)


def dc_single_force_coupled_from_deviates(z3, frac):
    """generate_random_DC_single_force_coupled_tensor (:333-367): force along the DC slip vector."""
    frac = np.asarray(frac, dtype=float)
    theta, phi = _angles_arctan2(_unit_rows(z3))
    six = _normalise_six(_six_from_full(_rot_mt(_DC_MT, theta, phi)))
    ned = _rot_force(np.array([1.0, 0.0, 0.0]), theta, phi)
    end = np.stack([ned[:, 1], ned[:, 0], ned[:, 2]], axis=1)  # NED -> END (:358)
    out = np.concatenate([six * frac[:, None], end * (1.0 - frac)[:, None]], axis=1)
    return np.ascontiguousarray(out.T), frac


def dc_single_force_uncoupled_from_deviates(z3_dc, z3_sf, frac):
    """generate_random_DC_single_force_uncoupled_tensor (:369-382)."""
    frac = np.asarray(frac, dtype=float)
    dc, sf = dc_from_deviates(z3_dc), single_force_from_deviates(z3_sf)
    return np.ascontiguousarray(np.concatenate([dc * frac, sf * (1.0 - frac)], axis=0)), frac


def dc_crack_coupled_from_deviates(u_theta, r_phi, r_quadrant, frac, z3):
    """generate_random_DC_crack_coupled_tensor (:384-446)."""
    frac = np.asarray(frac, dtype=float)
    crack = _crack_tensor(u_theta, r_phi, r_quadrant)
    mixed = frac[:, None, None] * _DC_MT + (1.0 - frac)[:, None, None] * crack
    theta, phi = _angles_arctan2(_unit_rows(z3))
    six = _normalise_six(_six_from_full(_rot_mt(mixed, theta, phi)))
    return np.ascontiguousarray(six.T), frac


def single_force_crack_uncoupled_from_deviates(z3_sf, u_theta, r_phi, r_quadrant, z3_rot, frac):
    """generate_random_single_force_crack_uncoupled_tensor (:448-510); the crack part is NOT
    re-normalised and its rotation angles come from arccos (:497-498), as in the reference."""
    frac = np.asarray(frac, dtype=float)
    sf = single_force_from_deviates(z3_sf)
    crack = _crack_tensor(u_theta, r_phi, r_quadrant)
    theta, phi = _angles_arccos(_unit_rows(z3_rot))
    six = _six_from_full(_rot_mt(crack, theta, phi))
    return np.ascontiguousarray(np.concatenate([six.T * (1.0 - frac), sf * frac], axis=0)), frac


# ---------------------------------------------------------------------------------------------
# deviates
# ---------------------------------------------------------------------------------------------
def draw_deviates(inversion_type, num_samples, rng=None, reference_stream=False):
    """Random deviates for ``num_samples`` samples of ``inversion_type`` as a dict of arrays.

    ``reference_stream=True`` consumes the GLOBAL ``numpy.random`` and ``random`` generators in the
    reference's per-sample order (seed them with ``np.random.seed`` / ``random.seed`` first);
    otherwise ``rng`` (a ``numpy.random.Generator``, default ``default_rng()``) is used in bulk.
    """
    if inversion_type not in INVERSION_TYPES:
        raise ValueError("inversion_type must be one of %s" % (INVERSION_TYPES,))
    n = int(num_samples)
    if reference_stream:
        # the legacy global generator fills arrays with the same draws, in the same order, as
        # repeated scalar np.random.normal() calls, so samplers that only draw normals go in bulk;
        # the crack samplers interleave np.random.uniform with the normals and are drawn per sample
        normal, std = np.random.normal, _stdlib_random.random
        if inversion_type == "full_mt":
            return {"z6": normal(size=(n, 6))}
        if inversion_type in ("DC", "single_force"):
            return {"z3": normal(size=(n, 3))}
        if inversion_type == "DC_single_force_couple":
            return {"z3": normal(size=(n, 3)), "frac": np.array([std() for _ in range(n)])}
        if inversion_type == "DC_single_force_no_coupling":
            z = normal(size=(n, 6))  # per sample: 3 for the DC orientation, then 3 for the force
            return {"z3_dc": z[:, :3], "z3_sf": z[:, 3:], "frac": np.array([std() for _ in range(n)])}
        out = {k: np.empty(n) for k in ("u_theta", "r_phi", "r_quadrant", "frac")}
        if inversion_type == "DC_crack_couple":
            out["z3"] = np.empty((n, 3))
            for i in range(n):  # :393, :394, :402, :425, :430
                out["u_theta"][i] = np.random.uniform(-1.0, 1.0)
                out["r_phi"][i], out["r_quadrant"][i], out["frac"][i] = std(), std(), std()
                out["z3"][i] = normal(size=3)
            return out
        out["z3_sf"], out["z3_rot"] = np.empty((n, 3)), np.empty((n, 3))
        for i in range(n):  # :454, :457, :458, :466, :491, :507
            out["z3_sf"][i] = normal(size=3)
            out["u_theta"][i] = np.random.uniform(-1.0, 1.0)
            out["r_phi"][i], out["r_quadrant"][i] = std(), std()
            out["z3_rot"][i] = normal(size=3)
            out["frac"][i] = std()
        return out
    rng = np.random.default_rng() if rng is None else rng
    if inversion_type == "full_mt":
        return {"z6": rng.standard_normal((n, 6))}
    if inversion_type in ("DC", "single_force"):
        return {"z3": rng.standard_normal((n, 3))}
    if inversion_type == "DC_single_force_couple":
        return {"z3": rng.standard_normal((n, 3)), "frac": rng.random(n)}
    if inversion_type == "DC_single_force_no_coupling":
        return {"z3_dc": rng.standard_normal((n, 3)), "z3_sf": rng.standard_normal((n, 3)), "frac": rng.random(n)}
    crack = {"u_theta": rng.uniform(-1.0, 1.0, n), "r_phi": rng.random(n), "r_quadrant": rng.random(n),
             "frac": rng.random(n)}
    if inversion_type == "DC_crack_couple":
        crack["z3"] = rng.standard_normal((n, 3))
        return crack
    crack["z3_sf"], crack["z3_rot"] = rng.standard_normal((n, 3)), rng.standard_normal((n, 3))
    return crack


def from_deviates(inversion_type, dev):
    """Samples of ``inversion_type`` from a ``draw_deviates`` dict: ``(MTs (n, N), amp_frac or None)``."""
    if inversion_type == "full_mt":
        return full_mt_from_deviates(dev["z6"]), None
    if inversion_type == "DC":
        return dc_from_deviates(dev["z3"]), None
    if inversion_type == "single_force":
        return single_force_from_deviates(dev["z3"]), None
    if inversion_type == "DC_single_force_couple":
        return dc_single_force_coupled_from_deviates(dev["z3"], dev["frac"])
    if inversion_type == "DC_single_force_no_coupling":
        return dc_single_force_uncoupled_from_deviates(dev["z3_dc"], dev["z3_sf"], dev["frac"])
    if inversion_type == "DC_crack_couple":
        return dc_crack_coupled_from_deviates(dev["u_theta"], dev["r_phi"], dev["r_quadrant"], dev["frac"],
                                              dev["z3"])
    if inversion_type == "single_force_crack_no_coupling":
        return single_force_crack_uncoupled_from_deviates(dev["z3_sf"], dev["u_theta"], dev["r_phi"],
                                                          dev["r_quadrant"], dev["z3_rot"], dev["frac"])
    raise ValueError("inversion_type must be one of %s" % (INVERSION_TYPES,))


def draw(inversion_type, num_samples, rng=None, reference_stream=False):
    """``num_samples`` random sources of ``inversion_type``: ``(MTs (n, N), amp_frac (N,) or None)``."""
    return from_deviates(inversion_type, draw_deviates(inversion_type, num_samples, rng, reference_stream))
