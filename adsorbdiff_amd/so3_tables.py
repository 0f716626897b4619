"""IGSO(3) tables of the rotation noise: sampling CDF, score magnitude and expected score norm on the reference's
(eps, omega) grid — what ``adsorbdiff/utils/rot_utils.py:140-264`` precomputes at import time (~6 min of numpy there).

Here the series  f(w) = sum_l (2l+1) exp(-l(l+1) eps^2) sin((l+1/2) w) / sin(w/2)  (L = 2000 terms, 1000 eps values x
2000 angles) is evaluated in fp64 torch ops on the ROCm device when there is one (a few seconds), otherwise on the host,
and cached as ``adsorbdiff_amd/_cache/igso3_v1.npz``.  Look-ups (``sample_vec``, ``score_vec``, ``score_norm``) follow the
reference's nearest-row / linear-interpolation rules and consume numpy's global random stream in the same order.
Pinned against the reference's own tables by tests/golden/igso3_tables.npz (oracle/make_golden.py section 6).
"""
from __future__ import annotations

import os
import zipfile
from pathlib import Path
from typing import Optional

import numpy as np
import torch

MIN_EPS, MAX_EPS, N_EPS = 0.01, 2, 1000
X_N = 2000
L_TERMS = 2000
_CACHE = Path(__file__).resolve().parent / "_cache" / "igso3_v1.npz"


def compute_tables(device: Optional[torch.device] = None, rows=None) -> dict:
    """cdf [n, X_N], score [n, X_N], exp_score_norm [n] for the listed eps rows (default: all N_EPS)."""
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    f64 = torch.float64
    rows = np.arange(N_EPS) if rows is None else np.asarray(rows).reshape(-1)
    eps_all = 10 ** np.linspace(np.log10(MIN_EPS), np.log10(MAX_EPS), N_EPS)
    om = torch.from_numpy(np.linspace(0, np.pi, X_N + 1)[1:]).to(device, f64)
    l = torch.arange(L_TERMS, dtype=f64, device=device)[:, None]
    arg = om[None, :] * (l + 0.5)
    hi, dhi = torch.sin(arg), (l + 0.5) * torch.cos(arg)
    lo, dlo = torch.sin(om / 2), 0.5 * torch.cos(om / 2)
    ratio = hi / lo[None, :]
    dratio = (lo[None, :] * dhi - hi * dlo[None, :]) / lo[None, :] ** 2
    cdf = np.empty((len(rows), X_N))
    score = np.empty((len(rows), X_N))
    esn = np.empty(len(rows))
    marg = (1 - torch.cos(om)) / np.pi
    for i, r in enumerate(rows):
        w = (2 * l + 1) * torch.exp(-l * (l + 1) * float(eps_all[int(r)]) ** 2)
        expansion = (w * ratio).sum(0)
        # Far in the tail of a narrow distribution the alternating series is pure cancellation noise (|f| ~ 1e-13 of its
        # peak); a device reduction order can land on exactly 0 there, which the host order never does.  Such angles
        # carry no probability mass: a non-finite ratio is replaced by 0 so it cannot poison the expectation below.
        sc = torch.nan_to_num((w * dratio).sum(0) / expansion, nan=0.0, posinf=0.0, neginf=0.0)
        pdf = expansion * marg
        cdf[i] = (pdf.cumsum(0) / X_N * np.pi).cpu().numpy()
        score[i] = sc.cpu().numpy()
        esn[i] = float(torch.sqrt((sc**2 * pdf).sum() / pdf.sum() / np.pi))
    return {"omegas": om.cpu().numpy(), "cdf": cdf, "score": score, "exp_score_norm": esn, "rows": rows}


class Igso3Tables:
    _shared = None

    def __init__(self, omegas, cdf, score, exp_score_norm) -> None:
        self.omegas = np.asarray(omegas)
        self.cdf = cdf                    # [N_EPS, X_N] or None (only needed by sample / sample_vec)
        self.score = score                # [N_EPS, X_N] or None (only needed by score_vec)
        self.exp_score_norm = np.asarray(exp_score_norm)

    @classmethod
    def shared(cls) -> "Igso3Tables":
        """The full tables: loaded from the cache file, computed (and cached) on first use."""
        if cls._shared is None:
            t = cls._load_cache()
            if t is None:
                t = compute_tables()
                cls._store_cache(t)
            cls._shared = cls(t["omegas"], t["cdf"], t["score"], t["exp_score_norm"])
        return cls._shared

    @staticmethod
    def _load_cache():
        """The cached tables, or None when the file is missing, unreadable (another rank is still writing an older,
        non-atomic version of it) or of another shape."""
        try:
            with np.load(_CACHE) as z:
                t = {k: z[k] for k in ("omegas", "cdf", "score", "exp_score_norm")}
        except (OSError, ValueError, KeyError, EOFError, zipfile.BadZipFile):
            return None
        ok = t["omegas"].shape == (X_N,) and t["cdf"].shape == (N_EPS, X_N) and t["score"].shape == (N_EPS, X_N) and \
            t["exp_score_norm"].shape == (N_EPS,)
        return t if ok else None

    @staticmethod
    def _store_cache(t) -> None:
        """Several ranks may get here at once: each writes its own temporary file and renames it into place (atomic; the
        contents are identical).  A read-only install just skips the cache."""
        try:
            _CACHE.parent.mkdir(exist_ok=True)
            tmp = _CACHE.with_name(f"{_CACHE.stem}.{os.getpid()}.tmp.npz")
            np.savez(tmp, omegas=t["omegas"], cdf=t["cdf"], score=t["score"], exp_score_norm=t["exp_score_norm"])
            os.replace(tmp, _CACHE)
        except OSError:
            pass

    @staticmethod
    def eps_index(eps):
        idx = (np.log10(eps) - np.log10(MIN_EPS)) / (np.log10(MAX_EPS) - np.log10(MIN_EPS)) * N_EPS
        return np.clip(np.around(idx).astype(int), a_min=0, a_max=N_EPS - 1)

    def sample(self, eps) -> float:
        return float(np.interp(np.random.rand(), self.cdf[self.eps_index(eps)], self.omegas))

    def sample_vec(self, eps) -> np.ndarray:
        x = np.random.randn(3)
        x /= np.linalg.norm(x)
        return x * self.sample(eps)

    def score_vec(self, eps, vec) -> np.ndarray:
        om = np.linalg.norm(vec)
        return np.interp(om, self.omegas, self.score[self.eps_index(eps)]) * vec / om

    @staticmethod
    def _lerp(x, x0, x1, f0, f1):
        """numpy's own interpolation arithmetic (np.interp): slope * (x - x0) + f0."""
        with np.errstate(divide="ignore", invalid="ignore"):
            return (f1 - f0) / (x1 - x0) * (x - x0) + f0

    def sample_and_score_vecs(self, eps: np.ndarray):
        """``sample_vec`` + ``score_vec`` for many systems: the global numpy stream is consumed system by system exactly
        as a loop over them does (three normals, then one uniform), the table look-ups run for all systems at once
        (same arithmetic as np.interp; 10 -> 1.5 ms for 256 systems).  Returns (vectors [B,3], scores [B,3]): the vectors
        equal the loop's bit for bit, the scores to ~1e-14 relative (the norm of the vector is summed in another order)."""
        eps = np.asarray(eps, dtype=np.float64).reshape(-1)
        B, n = eps.shape[0], self.omegas.shape[0]
        idx = self.eps_index(eps)
        x = np.empty((B, 3))
        nrm = np.empty(B)
        j = np.empty(B, dtype=np.int64)
        u = np.empty(B)
        for b in range(B):
            xb = np.random.randn(3)
            ub = np.random.rand()
            x[b] = xb
            nrm[b] = np.sqrt(xb.dot(xb))                                   # np.linalg.norm of a vector
            u[b] = ub
            j[b] = np.searchsorted(self.cdf[idx[b]], ub, side="right") - 1  # last k with cdf[k] <= u
        rows = np.arange(B)
        jc = np.clip(j, 0, n - 2)
        cdf0, cdf1 = self.cdf[idx, jc], self.cdf[idx, jc + 1]
        omega = self._lerp(u, cdf0, cdf1, self.omegas[jc], self.omegas[jc + 1])
        omega = np.where(j < 0, self.omegas[0], np.where(u >= self.cdf[idx, n - 1], self.omegas[n - 1], omega))
        vec = (x / nrm[:, None]) * omega[:, None]
        om = np.sqrt(np.einsum("ij,ij->i", vec, vec))
        k = np.searchsorted(self.omegas, om, side="right") - 1
        kc = np.clip(k, 0, n - 2)
        sc = self._lerp(om, self.omegas[kc], self.omegas[kc + 1], self.score[idx, kc], self.score[idx, kc + 1])
        sc = np.where(k < 0, self.score[idx, 0], np.where(om >= self.omegas[n - 1], self.score[idx, n - 1], sc))
        del rows
        return vec, sc[:, None] * vec / om[:, None]

    def score_norm(self, eps: torch.Tensor) -> torch.Tensor:
        return torch.from_numpy(self.exp_score_norm[self.eps_index(eps.detach().cpu().numpy())]).float()
