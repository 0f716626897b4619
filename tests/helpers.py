"""Shared helpers for the parity tests."""
from pathlib import Path

import numpy as np
import torch

from adsorbdiff_amd.data import Batch

GOLDEN = Path(__file__).resolve().parent / "golden"


def load_npz(name):
    with np.load(GOLDEN / name, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def batch_from_fixture(fx, pos_key="pos", device="cpu"):
    b = Batch()
    b.pos = torch.from_numpy(fx[pos_key]).float().clone()
    b.atomic_numbers = torch.from_numpy(fx["atomic_numbers"]).float()
    b.tags = torch.from_numpy(fx["tags"]).long()
    b.fixed = torch.from_numpy(fx["fixed"]).long()
    b.cell = torch.from_numpy(fx["cell"]).float()
    b.natoms = torch.from_numpy(fx["natoms"]).long()
    b.batch = torch.from_numpy(fx["batch"]).long()
    b.sid = [str(i) for i in range(len(b.natoms))]
    return b.to(device)


def state_dict_from_fixture(fx):
    return {k[4:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("sd::")}


def canon_edges(src, dst, dist, vec):
    """Canonical order for an edge multiset: by (dst, src, dist, vec)."""
    src, dst = np.asarray(src, np.int64), np.asarray(dst, np.int64)
    dist, vec = np.asarray(dist, np.float64), np.asarray(vec, np.float64)
    order = np.lexsort((np.round(vec[:, 2], 4), np.round(vec[:, 1], 4), np.round(vec[:, 0], 4), np.round(dist, 4), src, dst))
    return src[order], dst[order], dist[order], vec[order]


def rel_err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def row_rel_err(a, b, floor=1e-7):
    """max over rows of |a_i - b_i| / max(|b_i|, floor): every system / atom bounded against its OWN magnitude (the
    Frobenius ratio of rel_err lets large rows hide small ones)."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    return float(((a - b).norm(dim=1) / b.norm(dim=1).clamp(min=floor)).max())


def max_abs_err_rel_to_max(a, b):
    """max |a - b| / max row norm of b: the per-atom heads bounded element-wise against the largest atom."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.reshape(b.shape[0], -1).norm(dim=1).max().clamp(min=1e-30))
