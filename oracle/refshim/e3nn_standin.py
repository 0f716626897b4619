"""TEST INFRASTRUCTURE (build container only) — stand-in for the parts of ``e3nn==0.4.4`` that the reference's
EquiformerV2 touches (requirements.txt:3; call sites: models/equiformer_v2/so3.py:20-21,513-516,566-599,
equiformer_v2_denoising.py:8,124, drop.py:10,81).  e3nn itself cannot be installed here (no network), and no part of it
is vendored in /root/reference except wigner.py + Jd.pt.

PARITY STATUS — UNPINNED: everything below is a restatement of e3nn 0.4.4's PUBLISHED algorithm (o3/_rotation.py,
o3/_s2grid.py, o3/_legendre) from its documentation and paper conventions, not checked against the real package.
What this file CAN check, and does in ``self_check()``:
  * angles_to_matrix / xyz_to_angles are mutually consistent (Y-polar-axis convention, R = Ry(a) Rx(b) Ry(c));
  * the real spherical harmonics implied by (sha, shb) transform under the reference's own VENDORED Wigner-D
    (wigner.py + Jd.pt):  Y(R x) = D(R) Y(x)  for l <= 6 — this pins the basis up to one sign per degree l;
  * FromS2Grid o ToS2Grid = identity on band-limited coefficients (normalisation constants are mutually consistent).
And, in tests/test_oracle_golden.py::test_e3nn_standin_harmonics_equal_scipy_orthonormal_harmonics: the harmonics built
from (_legendre, _sh_alpha) equal scipy's orthonormal spherical harmonics for every (l, m), l <= 6, up to one sign per
(l, m) - the amplitude of the Legendre factor is pinned against an independent implementation.
What it CANNOT check: e3nn's own constant for normalization="component" (sqrt(4 pi) / sqrt(2l+1) / sqrt(lmax+1) on the
way to the grid, restated from its published source) — it changes the amplitude seen by the point-wise non-linearity of
the S2 activation — and the per-l sign (harmless on an inversion-symmetric grid such as the shipped 18 x 18 one).
"""
from __future__ import annotations

import math
import types

import numpy as np
import torch


# ---- rotations (e3nn o3/_rotation.py conventions: Y is the polar axis) ------------------------------------------
def matrix_x(angle):
    c, s, o, z = angle.cos(), angle.sin(), torch.ones_like(angle), torch.zeros_like(angle)
    return torch.stack([torch.stack([o, z, z], -1), torch.stack([z, c, -s], -1), torch.stack([z, s, c], -1)], -2)


def matrix_y(angle):
    c, s, o, z = angle.cos(), angle.sin(), torch.ones_like(angle), torch.zeros_like(angle)
    return torch.stack([torch.stack([c, z, s], -1), torch.stack([z, o, z], -1), torch.stack([-s, z, c], -1)], -2)


def angles_to_matrix(alpha, beta, gamma):
    alpha, beta, gamma = torch.broadcast_tensors(alpha, beta, gamma)
    return matrix_y(alpha) @ matrix_x(beta) @ matrix_y(gamma)


def xyz_to_angles(xyz):
    xyz = torch.nn.functional.normalize(xyz, p=2, dim=-1).clamp(-1, 1)
    return torch.atan2(xyz[..., 0], xyz[..., 2]), torch.acos(xyz[..., 1])


# ---- spherical harmonics on the (beta, alpha) product grid ------------------------------------------------------
def _legendre(lmax: int, z: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """[..., (lmax+1)^2]: for every (l, m) the beta-dependent factor
    (-1)^l sqrt((2l+1)/(4 pi) (l-|m|)!/(l+|m|)!) y^|m| / (2^l l!) d^{l+|m|}/dz^{l+|m|} (z^2-1)^l   (integral-normalised,
    no Condon-Shortley phase; z = cos(beta), y = sin(beta))."""
    out = []
    zd = z.double()
    yd = y.double()
    for l in range(lmax + 1):
        base = np.polynomial.polynomial.polypow([-1.0, 0.0, 1.0], l)  # (z^2 - 1)^l, ascending powers
        row = {}
        for m in range(l + 1):
            der = np.polynomial.polynomial.polyder(base, l + m)
            val = torch.zeros_like(zd)
            for k, ck in enumerate(der):
                val = val + float(ck) * zd**k
            norm = (-1) ** l * math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - m) / math.factorial(l + m))
            row[m] = norm * yd**m * val / (2**l * math.factorial(l))
        for m in range(-l, l + 1):
            out.append(row[abs(m)])
    return torch.stack(out, -1).to(z.dtype)


def _sh_alpha(lmax: int, alpha: torch.Tensor) -> torch.Tensor:
    """[..., 2 lmax + 1]: sqrt2 sin(|m| a) for m < 0, 1, sqrt2 cos(m a) for m > 0; m = -lmax..lmax."""
    a = alpha.unsqueeze(-1)
    m_pos = torch.arange(1, lmax + 1, dtype=alpha.dtype)
    m_neg = torch.arange(lmax, 0, -1, dtype=alpha.dtype)
    return torch.cat([math.sqrt(2) * torch.sin(m_neg * a), torch.ones_like(a), math.sqrt(2) * torch.cos(m_pos * a)], -1)


def _expand_matrix(lmax: int) -> torch.Tensor:
    """[l, m, i]: flat (l, m) index i <-> (l, m + lmax) matrix position."""
    mat = torch.zeros(lmax + 1, 2 * lmax + 1, (lmax + 1) ** 2)
    i = 0
    for l in range(lmax + 1):
        mat[l, lmax - l : lmax + l + 1, i : i + 2 * l + 1] = torch.eye(2 * l + 1)
        i += 2 * l + 1
    return mat


def s2_grid(res_beta: int, res_alpha: int):
    betas = (torch.arange(res_beta, dtype=torch.get_default_dtype()) + 0.5) / res_beta * math.pi
    alphas = torch.arange(res_alpha, dtype=torch.get_default_dtype()) / res_alpha * 2 * math.pi
    return betas, alphas


def _quadrature_weights(b: int) -> torch.Tensor:
    """Kostelec & Rockmore weights for the 2b latitudes (beta_j = (j + 1/2) pi / 2b), as in lie_learn / e3nn."""
    k = torch.arange(b, dtype=torch.float64)
    w = torch.tensor([
        (2.0 / b) * math.sin(math.pi * (2.0 * j + 1.0) / (4.0 * b))
        * float(((1.0 / (2 * k + 1)) * torch.sin((2 * j + 1) * (2 * k + 1) * math.pi / (4.0 * b))).sum())
        for j in range(2 * b)], dtype=torch.float64)
    return (w / (2.0 * (2 * b) ** 2)).to(torch.get_default_dtype())


def real_sh(lmax: int, xyz: torch.Tensor) -> torch.Tensor:
    """Integral-normalised real spherical harmonics at points xyz [..., 3] -> [..., (lmax+1)^2] (used by self_check)."""
    alpha, beta = xyz_to_angles(xyz)
    shb = _legendre(lmax, beta.cos(), beta.sin().abs())
    sha = _sh_alpha(lmax, alpha)
    m = _expand_matrix(lmax)
    return torch.einsum("lmi,...i,...m->...i", m, shb, sha)


class ToS2Grid(torch.nn.Module):
    def __init__(self, lmax=None, res=None, normalization="component", dtype=None, device=None):
        super().__init__()
        res_beta, res_alpha = res
        betas, alphas = s2_grid(res_beta, res_alpha)
        shb = _legendre(lmax, betas.cos(), betas.sin().abs())  # [b, i]
        sha = _sh_alpha(lmax, alphas)                          # [a, m]
        if normalization == "component":
            n = math.sqrt(4 * math.pi) * torch.tensor([1 / math.sqrt(2 * l + 1) for l in range(lmax + 1)]) / math.sqrt(lmax + 1)
        elif normalization == "norm":
            n = math.sqrt(4 * math.pi) * torch.ones(lmax + 1) / math.sqrt(lmax + 1)
        elif normalization == "integral":
            n = torch.ones(lmax + 1)
        else:
            raise ValueError(normalization)
        m = _expand_matrix(lmax)
        self.register_buffer("alphas", alphas)
        self.register_buffer("betas", betas)
        self.register_buffer("sha", sha)
        self.register_buffer("shb", torch.einsum("lmj,bj,lmi,l->mbi", m, shb, m, n))


class FromS2Grid(torch.nn.Module):
    def __init__(self, res=None, lmax=None, normalization="component", lmax_in=None, dtype=None, device=None):
        super().__init__()
        res_beta, res_alpha = res
        if lmax_in is None:
            lmax_in = lmax
        betas, alphas = s2_grid(res_beta, res_alpha)
        shb = _legendre(lmax, betas.cos(), betas.sin().abs())
        sha = _sh_alpha(lmax, alphas)
        if normalization == "component":
            n = math.sqrt(4 * math.pi) * torch.tensor([math.sqrt(2 * l + 1) for l in range(lmax + 1)]) * math.sqrt(lmax_in + 1)
        elif normalization == "norm":
            n = math.sqrt(4 * math.pi) * torch.ones(lmax + 1) * math.sqrt(lmax_in + 1)
        elif normalization == "integral":
            n = 4 * math.pi * torch.ones(lmax + 1)
        else:
            raise ValueError(normalization)
        m = _expand_matrix(lmax)
        assert res_beta % 2 == 0
        qw = _quadrature_weights(res_beta // 2) * res_beta**2 / res_alpha
        self.register_buffer("alphas", alphas)
        self.register_buffer("betas", betas)
        self.register_buffer("sha", sha)
        self.register_buffer("shb", torch.einsum("lmj,bj,lmi,l,b->mbi", m, shb, m, n, qw))


class Irreps:
    """Only what the reference reads: ``Irreps.spherical_harmonics(lmax, p)`` (stored, unused), ``num_irreps``, ``dim``."""

    def __init__(self, spec="") -> None:
        self.spec = spec
        self.num_irreps = 0
        self.dim = 0

    @classmethod
    def spherical_harmonics(cls, lmax, p=-1):
        out = cls("sh%d" % lmax)
        out.num_irreps, out.dim = lmax + 1, (lmax + 1) ** 2
        return out


class ElementwiseTensorProduct(torch.nn.Module):
    def __init__(self, *a, **k) -> None:
        super().__init__()

    def forward(self, x, mask):
        raise NotImplementedError("equivariant dropout is training-only; the goldens run in eval mode")


def install(sys_modules) -> None:
    e3nn = types.ModuleType("e3nn")
    o3 = types.ModuleType("e3nn.o3")
    for name, obj in (("angles_to_matrix", angles_to_matrix), ("xyz_to_angles", xyz_to_angles), ("ToS2Grid", ToS2Grid),
                      ("FromS2Grid", FromS2Grid), ("Irreps", Irreps), ("ElementwiseTensorProduct", ElementwiseTensorProduct),
                      ("matrix_x", matrix_x), ("matrix_y", matrix_y)):
        setattr(o3, name, obj)
    e3nn.o3 = o3
    e3nn.__version__ = "0.4.4-standin"
    sys_modules["e3nn"] = e3nn
    sys_modules["e3nn.o3"] = o3


def self_check(wigner_D, lmax: int = 6) -> dict:
    """Consistency of the stand-in with the reference's vendored Wigner-D (see the module docstring)."""
    g = torch.Generator().manual_seed(0)
    out = {}
    # (1) Y(R x) = D(R) Y(x)
    ang = torch.rand(16, 3, generator=g, dtype=torch.float64) * torch.tensor([2 * math.pi, math.pi, 2 * math.pi]).double()
    x = torch.nn.functional.normalize(torch.randn(16, 3, generator=g, dtype=torch.float64), dim=-1)
    R = angles_to_matrix(ang[:, 0], ang[:, 1], ang[:, 2])
    Y = real_sh(lmax, x)
    YR = real_sh(lmax, torch.einsum("nij,nj->ni", R, x))
    worst = 0.0
    i = 0
    for l in range(lmax + 1):
        D = wigner_D(l, ang[:, 0], ang[:, 1], ang[:, 2]).double()
        worst = max(worst, float((torch.einsum("nij,nj->ni", D, Y[:, i : i + 2 * l + 1]) - YR[:, i : i + 2 * l + 1]).abs().max()))
        i += 2 * l + 1
    out["equivariance_max_abs"] = worst
    # (2) FromS2Grid o ToS2Grid = identity (component normalisation, the model's resolution and a default one)
    for lm, res in ((4, (18, 18)), (6, (18, 18)), (6, (14, 15))):
        to, fr = ToS2Grid(lm, res, normalization="component"), FromS2Grid(res, lm, normalization="component")
        tg = torch.einsum("mbi,am->bai", to.shb, to.sha)
        fg = torch.einsum("am,mbi->bai", fr.sha, fr.shb)
        eye = torch.einsum("bai,baj->ij", fg, tg)
        out["roundtrip_l%d_%dx%d" % (lm, res[0], res[1])] = float((eye - torch.eye((lm + 1) ** 2)).abs().max())
    return out
