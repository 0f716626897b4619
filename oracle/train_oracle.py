"""TEST INFRASTRUCTURE — CPU restatement of the score-matching TRAINING step of AdsorbDiff.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.  Pinned: oracle/make_golden.py
executes the reference's own functions (sde_denoising_trainer.py:45-135, 675-728; rot_utils.py:140-264) and asserts
equality with the functions below before it writes tests/golden/train_small.npz and igso3_tables.npz.

  igso3_rows              IGSO(3) series tables              adsorbdiff/utils/rot_utils.py:140-223
  Igso3.sample/score_*    table look-ups                     rot_utils.py:226-264
  pbc_correction          minimum-image wrap of a vector     trainers/sde_denoising_trainer.py:45-64
  tr_so3_schedule         forward noising of a batch         trainers/sde_denoising_trainer.py:67-135
  score_matching_loss     DenoisingTrainer._compute_loss     trainers/sde_denoising_trainer.py:675-728
Gradients come from torch.autograd through oracle/painn_oracle.py's forward (plain PyTorch ops).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .painn_oracle import ads_mean, axis_angle_to_matrix

MIN_EPS, MAX_EPS, N_EPS = 0.01, 2, 1000  # rot_utils.py:9
X_N = 2000                               # rot_utils.py:10
L_TERMS = 2000                           # default L of _expansion / _score


def igso3_rows(eps_rows) -> Dict[str, np.ndarray]:
    """cdf [n, X_N], score [n, X_N], exp_score_norm [n] for the listed rows of the eps grid (rot_utils.py:189-217).
    density f(w) = sum_l (2l+1) exp(-l(l+1) eps^2) sin((l+1/2) w) / sin(w/2); marginal pdf = f (1 - cos w) / pi."""
    eps_array = 10 ** np.linspace(np.log10(MIN_EPS), np.log10(MAX_EPS), N_EPS)
    om = np.linspace(0, np.pi, X_N + 1)[1:]
    l = np.arange(L_TERMS, dtype=np.float64)[:, None]
    lo, dlo = np.sin(om / 2), 0.5 * np.cos(om / 2)
    cdf, score, esn = [], [], []
    for r in np.asarray(eps_rows).reshape(-1):
        eps = eps_array[int(r)]
        w = (2 * l + 1) * np.exp(-l * (l + 1) * eps**2)
        hi = np.sin(om[None, :] * (l + 0.5))
        dhi = (l + 0.5) * np.cos(om[None, :] * (l + 0.5))
        expansion = (w * hi / lo[None, :]).sum(0)
        dsigma = (w * (lo[None, :] * dhi - hi * dlo[None, :]) / lo[None, :] ** 2).sum(0)
        pdf = expansion * (1 - np.cos(om)) / np.pi
        sc = dsigma / expansion
        cdf.append(pdf.cumsum() / X_N * np.pi)
        score.append(sc)
        esn.append(np.sqrt(np.sum(sc**2 * pdf) / np.sum(pdf) / np.pi))
    return {"cdf": np.asarray(cdf), "score": np.asarray(score), "exp_score_norm": np.asarray(esn), "omegas": om}


class Igso3:
    """Look-ups on the tables (rot_utils.py:226-264).  `tables` is anything with the reference module's array names
    (_omegas_array, _cdf_vals, _score_norms, _exp_score_norms): the reference module itself in make_golden.py, or a
    dict built from igso3_rows for the eps values a test uses."""

    def __init__(self, tables) -> None:
        get = (lambda k: tables[k]) if isinstance(tables, dict) else (lambda k: getattr(tables, k))
        self.omegas, self.cdf = get("_omegas_array"), get("_cdf_vals")
        self.score_norms, self.exp_score_norms = get("_score_norms"), get("_exp_score_norms")

    @staticmethod
    def eps_index(eps):
        idx = (np.log10(eps) - np.log10(MIN_EPS)) / (np.log10(MAX_EPS) - np.log10(MIN_EPS)) * N_EPS
        return np.clip(np.around(idx).astype(int), a_min=0, a_max=N_EPS - 1)

    def sample(self, eps):  # :226-235, consumes one np.random.rand()
        return np.interp(np.random.rand(), self.cdf[self.eps_index(eps)], self.omegas)

    def sample_vec(self, eps):  # :238-241, consumes np.random.randn(3) then np.random.rand()
        x = np.random.randn(3)
        x /= np.linalg.norm(x)
        return x * self.sample(eps)

    def score_vec(self, eps, vec):  # :244-253
        om = np.linalg.norm(vec)
        return np.interp(om, self.omegas, self.score_norms[self.eps_index(eps)]) * vec / om

    def score_norm(self, eps: torch.Tensor) -> torch.Tensor:  # :256-264
        return torch.from_numpy(np.asarray(self.exp_score_norms)[self.eps_index(eps.numpy())]).float()


@torch.no_grad()
def pbc_correction(noise_vec: torch.Tensor, cell: torch.Tensor) -> torch.Tensor:
    """Per system: fractional = solve(cell^T, v) in fp64, wrapped into (-0.5, 0.5], back with the ROWS of cell
    (sde_denoising_trainer.py:45-64; noise_vec is [B,3], one vector per system)."""
    out = torch.zeros_like(noise_vec)
    for b in range(cell.shape[0]):
        frac = torch.linalg.solve(cell[b].t().double(), noise_vec[b].reshape(1, 3).t().double()).t()
        frac %= 1.0
        frac %= 1.0
        frac[frac > 0.5] -= 1
        out[b] = torch.matmul(frac.float(), cell[b].float())
    return out


def tr_so3_schedule(pos, cell, tags, batch, natoms, params: dict, igso: Igso3, draws: Optional[dict] = None) -> dict:
    """Forward noising (sde_denoising_trainer.py:67-135).  Random streams in the reference's order: torch.rand(B) for
    t, torch normal_ [B,3] for the COM noise, then per system np.random.randn(3) + np.random.rand() for the rotation.
    `draws` (t, com_noise, rot_update) replaces the streams when given."""
    B = int(natoms.shape[0])
    lo, hi = params["ads_std_low"], params["ads_std_high"]
    rlo, rhi = params["rot_std_low"], params["rot_std_high"]
    t = draws["t"] if draws else torch.rand(size=(B,))
    tr_sigma = lo ** (1 - t) * hi**t
    rot_sigma = rlo ** (1 - t) * rhi**t
    m = tags == 2
    center = ads_mean(pos, tags, batch, B)
    noise = draws["com_noise"].clone() if draws else torch.zeros(center.shape).normal_()
    noise = noise * tr_sigma[:, None]
    noise = pbc_correction(noise, cell)
    noise[:, -1] = 0
    ads_pos = pos[m]
    bm = batch[m]
    rot_scores, new_ads = [], []
    for b in range(B):
        rot_update = draws["rot_update"][b].numpy() if draws else igso.sample_vec(eps=rot_sigma[b].item())
        R = axis_angle_to_matrix(torch.tensor(rot_update)).float()
        rot_scores.append(torch.from_numpy(igso.score_vec(vec=rot_update, eps=rot_sigma[b].item())).float().unsqueeze(0))
        new_ads.append((ads_pos[bm == b] - center[b]) @ R.T + noise[b] + center[b])
    new_ads = torch.cat(new_ads)
    new_ads[:, -1] += 1  # "move the adsorbate up by roughly 1 A" (:127)
    new_pos = pos.clone()
    new_pos[m] = new_ads
    return {"pos": new_pos, "tr_sigma": tr_sigma[:, None], "rot_sigma": rot_sigma[:, None],
            "rot_score": torch.cat(rot_scores), "ads_center_noise_vec": noise, "tr_score": -noise / tr_sigma[:, None] ** 2}


def score_matching_loss(out1, out2, tags, batch, noised: dict, igso: Igso3, pos_coefficient: float = 1.0):
    """DenoisingTrainer._compute_loss with so3_denoising (sde_denoising_trainer.py:675-728).  Returns (loss, [terms])."""
    B = noised["tr_sigma"].shape[0]
    p = ads_mean(out1, tags, batch, B) / noised["tr_sigma"]
    p = torch.cat([p[:, :2], torch.zeros_like(p[:, 2:])], dim=1)  # out["positions"][:, -1] = 0
    l_tr = ((p - noised["tr_score"]) ** 2 * noised["tr_sigma"] ** 2).mean()
    r = ads_mean(out2, tags, batch, B) / noised["rot_sigma"]
    norm = igso.score_norm(noised["rot_sigma"].cpu())
    l_rot = (((r - noised["rot_score"]) / norm) ** 2).mean()
    return l_tr + l_rot, [l_tr, l_rot]
