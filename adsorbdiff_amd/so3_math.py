"""Constant tables of the EquiformerV2 denoiser's SO(3) machinery, computed on the host in float64 at engine creation
and handed to the library through ``adf_eqv2_set_constants`` (no per-step host work).

What the reference gets from ``e3nn==0.4.4`` and from the vendored ``Jd.pt`` (models/equiformer_v2/so3.py:509-531,
566-613; wigner.py:8,16-40) is derived here from the published definitions:

* real spherical harmonics in e3nn's convention — Y is the polar axis, integral normalisation, (l, m) order with
  m = -l..l, ``sqrt2 sin(|m| a)`` for m < 0 and ``sqrt2 cos(m a)`` for m > 0, no Condon-Shortley phase, a (-1)^l factor;
* ``J_l`` = Wigner matrix of the half-turn about (x + y)/sqrt2, the rotation that conjugates rotations about Y into
  rotations about X, so that D_l(Ry(a) Rx(b) Ry(c)) = Z(a) J Z(b) J Z(c) with Z the Y-rotation matrices of wigner.py:31-40.
  Solved from Y_l(R x) = J_l Y_l(x) on a fixed point set;
* the S2 grid (res_beta x res_alpha, Kostelec-Rockmore quadrature weights, "component" normalisation) transforms
  ``to_grid`` / ``from_grid`` with the reference's m-truncation rescale (so3.py:566-613).

Layouts handed to the device ("m-major reduced" order of the |m| <= mmax coefficients: m = 0 for l = 0..L, then for
m = 1..mmax the +m entries for l = m..L followed by the -m entries): see ``reduced_order``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np


def lm_pairs(lmax: int, mmax: int) -> List[Tuple[int, int]]:
    """(l, m) with |m| <= min(l, mmax), degree-major — the reference's coefficient order (so3.py:56-72)."""
    return [(l, m) for l in range(lmax + 1) for m in range(-min(l, mmax), min(l, mmax) + 1)]


def reduced_order(lmax: int, mmax: int) -> List[Tuple[int, int]]:
    """m-major order of the reduced coefficients (so3.py:83-103): what an SO(2) convolution consumes block by block."""
    out = [(l, 0) for l in range(lmax + 1)]
    for m in range(1, mmax + 1):
        out += [(l, m) for l in range(m, lmax + 1)]
        out += [(l, -m) for l in range(m, lmax + 1)]
    return out


def _legendre_part(lmax: int, z: np.ndarray, y: np.ndarray) -> np.ndarray:
    """[..., (lmax+1)^2] polar-angle factor of every (l, m): z = cos(beta), y = sin(beta) >= 0."""
    P = np.polynomial.polynomial
    cols = []
    for l in range(lmax + 1):
        base = P.polypow([-1.0, 0.0, 1.0], l)  # (z^2 - 1)^l
        by_m = {}
        for m in range(l + 1):
            der = P.polyder(base, l + m)
            val = P.polyval(z, der)
            norm = (-1.0) ** l * math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - m) / math.factorial(l + m))
            by_m[m] = norm * y**m * val / (2.0**l * math.factorial(l))
        cols += [by_m[abs(m)] for m in range(-l, l + 1)]
    return np.stack(cols, axis=-1)


def _azimuth_part(lmax: int, alpha: np.ndarray) -> np.ndarray:
    """[..., 2 lmax + 1] for m = -lmax..lmax."""
    a = alpha[..., None]
    neg = math.sqrt(2.0) * np.sin(np.arange(lmax, 0, -1) * a)
    pos = math.sqrt(2.0) * np.cos(np.arange(1, lmax + 1) * a)
    return np.concatenate([neg, np.ones_like(a), pos], axis=-1)


def real_sh(lmax: int, xyz: np.ndarray) -> np.ndarray:
    """Integral-normalised real harmonics at unit vectors xyz [..., 3] -> [..., (lmax+1)^2] (float64)."""
    xyz = np.asarray(xyz, dtype=np.float64)
    xyz = xyz / np.linalg.norm(xyz, axis=-1, keepdims=True)
    alpha = np.arctan2(xyz[..., 0], xyz[..., 2])
    cb = np.clip(xyz[..., 1], -1.0, 1.0)
    sb = np.sqrt(np.maximum(0.0, 1.0 - cb * cb))
    leg = _legendre_part(lmax, cb, sb)
    az = _azimuth_part(lmax, alpha)
    out = np.empty_like(leg)
    i = 0
    for l in range(lmax + 1):
        for m in range(-l, l + 1):
            out[..., i] = leg[..., i] * az[..., lmax + m]
            i += 1
    return out


def wigner_from_matrix(lmax: int, R: np.ndarray) -> List[np.ndarray]:
    """[D_0, ..., D_lmax] with Y_l(R x) = D_l Y_l(x), solved on a fixed generic point set (float64)."""
    rng = np.random.default_rng(20240607)
    pts = rng.standard_normal((6 * (lmax + 1) ** 2, 3))
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    Y = real_sh(lmax, pts)
    Yr = real_sh(lmax, pts @ np.asarray(R, dtype=np.float64).T)
    out = []
    for l in range(lmax + 1):
        a, b = l * l, (l + 1) ** 2
        Dt, *_ = np.linalg.lstsq(Y[:, a:b], Yr[:, a:b], rcond=None)  # Y D^T = Yr
        out.append(Dt.T)
    return out


_RJ = np.array([[0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])


def j_matrices(lmax: int) -> List[np.ndarray]:
    """J_l, l = 0..lmax (float64): symmetric, orthogonal, J_l^2 = 1.  The (-1)^l factor (the half-turn composed with
    the inversion) is e3nn's sign choice for its tabulated J; J enters every D_l twice, so D_l does not depend on it."""
    return [(-1.0) ** l * d for l, d in enumerate(wigner_from_matrix(lmax, _RJ))]


def y_rotation_matrix(l: int, angle: float) -> np.ndarray:
    """Z_l(angle) = D_l(Ry(angle)) in the form of wigner.py:31-40: cos(f_i a) on the diagonal, sin(f_i a) on the
    anti-diagonal, f_i = l - i."""
    n = 2 * l + 1
    M = np.zeros((n, n))
    f = np.arange(l, -l - 1, -1, dtype=np.float64)
    idx = np.arange(n)
    M[idx, n - 1 - idx] = np.sin(f * angle)
    M[idx, idx] = np.cos(f * angle)
    return M


def _quadrature_weights(b: int) -> np.ndarray:
    """Kostelec-Rockmore weights of the 2b latitudes beta_j = (j + 1/2) pi / 2b."""
    k = np.arange(b, dtype=np.float64)
    w = np.array([
        (2.0 / b) * math.sin(math.pi * (2 * j + 1) / (4.0 * b))
        * float(np.sum(np.sin((2 * j + 1) * (2 * k + 1) * math.pi / (4.0 * b)) / (2 * k + 1)))
        for j in range(2 * b)])
    return w / (2.0 * (2 * b) ** 2)


def s2_grid_matrices(lmax: int, mmax: int, res: int) -> Tuple[np.ndarray, np.ndarray]:
    """(to_grid, from_grid), each [res*res, n_coeff] float64 with grid point p = beta_index * res + alpha_index and the
    degree-major coefficients with |m| <= mmax; "component" normalisation and the m-truncation rescale
    sqrt((2l+1)/(2 mmax+1)) on degrees l > mmax when mmax < lmax (so3.py:566-613)."""
    if res % 2:
        raise ValueError("grid_resolution must be even")
    betas = (np.arange(res) + 0.5) / res * math.pi
    alphas = np.arange(res) / res * 2.0 * math.pi
    leg = _legendre_part(lmax, np.cos(betas), np.abs(np.sin(betas)))  # [b, i]
    az = _azimuth_part(lmax, alphas)                                   # [a, m]
    qw = _quadrature_weights(res // 2) * res**2 / res                  # [b]
    full = lm_pairs(lmax, lmax)
    to = np.empty((res, res, len(full)))
    fr = np.empty((res, res, len(full)))
    for i, (l, m) in enumerate(full):
        n_to = math.sqrt(4 * math.pi) / math.sqrt(2 * l + 1) / math.sqrt(lmax + 1)
        n_fr = math.sqrt(4 * math.pi) * math.sqrt(2 * l + 1) * math.sqrt(lmax + 1)
        resc = math.sqrt((2 * l + 1) / (2 * mmax + 1)) if (mmax != lmax and l > mmax) else 1.0
        to[:, :, i] = np.outer(leg[:, i] * n_to, az[:, lmax + m]) * resc
        fr[:, :, i] = np.outer(leg[:, i] * n_fr * qw, az[:, lmax + m]) * resc
    keep = [i for i, (l, m) in enumerate(full) if abs(m) <= mmax]
    return to[:, :, keep].reshape(res * res, -1), fr[:, :, keep].reshape(res * res, -1)


def device_tables(lmax: int, mmax: int, res: int) -> Dict[str, np.ndarray]:
    """float32 arrays in the layouts of adf_eqv2_set_constants (include/adsorbdiff_hip.h):
      jd        concatenated J_l, row-major, l = 0..lmax                       [sum (2l+1)^2]
      to_red    [G, S_r]  columns in m-major reduced order                     (separable S2 activation of the attention)
      from_red  [G, S_r]
      to_full   [G, S]    degree-major                                         (grid MLP of the feed-forward network)
      from_full [G, S]
    """
    jd = np.concatenate([j.reshape(-1) for j in j_matrices(lmax)])
    to_r, fr_r = s2_grid_matrices(lmax, mmax, res)
    red = lm_pairs(lmax, mmax)
    perm = [red.index(p) for p in reduced_order(lmax, mmax)]
    to_f, fr_f = s2_grid_matrices(lmax, lmax, res)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return {"jd": f32(jd), "to_red": f32(to_r[:, perm]), "from_red": f32(fr_r[:, perm]), "to_full": f32(to_f),
            "from_full": f32(fr_f)}
