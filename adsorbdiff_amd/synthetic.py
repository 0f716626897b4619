"""Synthetic OC20-Dense-shaped adsorbate+slab systems.

No dataset is reachable offline, so the benchmark and the parity tests use
this generator.  Per system: ``n_slab`` slab atoms on a jittered lattice in
z in [7, 17.5] A inside a skewed cell ``[a,0,0],[0.3a,0.95a,0],[0,0,35]`` with
a ~ U(14,16) A, plus a 4-atom adsorbate (C,H,O,H) around (a/2, a/2, 19 A).
Atoms are ordered slab-then-adsorbate (matters for the src<dst symmetrisation
rule, reference: adsorbdiff/models/painn/painn_denoising.py:264).
"""
from __future__ import annotations

import math

import torch

from .data import Batch, Data

ADS_Z = (6, 1, 8, 1)


def _lattice_dims(n_slab: int):
    """Pick (nx, ny, nz) with nx*ny*nz >= n_slab, close to the 7x7x4 of the 196-atom slab."""
    nz = 4
    nxy = max(1, math.ceil(math.sqrt(n_slab / nz)))
    return nxy, nxy, nz


def make_system(gen: torch.Generator, n_slab: int = 196, n_ads: int = 4, sid: str = "0") -> Data:
    nx, ny, nz = _lattice_dims(n_slab)
    # a ~ U(14,16) A for the 7x7x4 benchmark slab; smaller slabs keep the same
    # ~2.1 A lateral spacing, so their cells need several periodic images
    a = (14.0 + 2.0 * torch.rand((), generator=gen).item()) * nx / 7.0
    cell = torch.tensor([[a, 0.0, 0.0], [0.3 * a, 0.95 * a, 0.0], [0.0, 0.0, 35.0]], dtype=torch.float32)
    ix, iy, iz = torch.meshgrid(torch.arange(nx), torch.arange(ny), torch.arange(nz), indexing="ij")
    frac = torch.stack(
        [(ix.reshape(-1) + 0.5) / nx, (iy.reshape(-1) + 0.5) / ny, torch.zeros(nx * ny * nz)], dim=1
    )[:n_slab].float()
    zlayer = iz.reshape(-1)[:n_slab].float()
    z = 7.0 + (zlayer + 0.5) * (10.5 / nz)
    slab = frac @ cell
    slab[:, 2] = z
    slab = slab + (torch.rand(n_slab, 3, generator=gen) - 0.5) * 0.5
    slab_Z = torch.randint(20, 80, (n_slab,), generator=gen)
    centre = torch.tensor([a / 2, a / 2, 19.0])
    ads = centre + 0.7 * torch.randn(n_ads, 3, generator=gen)
    ads_Z = torch.tensor([ADS_Z[i % len(ADS_Z)] for i in range(n_ads)])
    pos = torch.cat([slab, ads]).float()
    Z = torch.cat([slab_Z, ads_Z]).float()
    zmid = 7.0 + 10.5 / 2
    tags = torch.cat([(slab[:, 2] > zmid).long(), torch.full((n_ads,), 2, dtype=torch.long)])
    fixed = (tags == 0).long()
    return Data(
        pos=pos,
        atomic_numbers=Z,
        tags=tags,
        fixed=fixed,
        cell=cell.reshape(1, 3, 3),
        natoms=torch.tensor([n_slab + n_ads]),
        sid=sid,
    )


def make_batch(num_systems: int, n_slab: int = 196, n_ads: int = 4, seed: int = 1000, sid_offset: int = 0) -> Batch:
    """``seed`` = 1000 + shard in the benchmark."""
    gen = torch.Generator().manual_seed(seed)
    systems = [make_system(gen, n_slab, n_ads, sid=str(sid_offset + i)) for i in range(num_systems)]
    return Batch.from_data_list(systems)
