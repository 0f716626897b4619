"""Multi-GPU plumbing of the sampler: shard independent systems over ranks, gather the sampled sites.

Systems never interact (block-diagonal graph, per-system update), so the path shards with no
data-path collective; the only exchange is one all_gather of the sampled adsorbate sites at the
end (RCCL over xGMI on the GPU box, gloo in the CPU tests).  This replaces the reference's
per-rank ``.npz`` files + barrier + rank-0 merge (adsorbdiff/trainers/sde_denoising_trainer.py:
862-909); the greedy partition mirrors its load balancing by atom count
(adsorbdiff/datasets/data_parallel.py:32-48).
"""
from __future__ import annotations

from typing import List, Sequence

import torch


def balanced_partition(sizes: Sequence[int], num_parts: int) -> List[List[int]]:
    """Greedy: largest system first, always onto the currently lightest rank."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    loads = [0] * num_parts
    parts: List[List[int]] = [[] for _ in range(num_parts)]
    for i in order:
        r = min(range(num_parts), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += int(sizes[i])
    return [sorted(p) for p in parts]


def shard_batch(batch, rank: int, world: int):
    """This rank's share of ``batch`` (systems dealt by atom count); returns (sub_batch, system ids)."""
    from .data import Batch

    parts = balanced_partition(batch.natoms.tolist(), world)
    mine = parts[rank]
    data = batch.to_data_list()
    return Batch.from_data_list([data[i] for i in mine]), mine


def adsorbate_sites(batch) -> torch.Tensor:
    """[B, A, 3] positions of each system's adsorbate (tag==2) atoms, NaN-padded to the largest
    adsorbate in the batch."""
    tags, bidx = batch.tags, batch.batch
    B = int(batch.natoms.shape[0])
    m = tags == 2
    idx_b = bidx[m]
    counts = torch.bincount(idx_b, minlength=B)
    A = int(counts.max().item()) if counts.numel() else 0
    start = torch.cumsum(counts, 0) - counts
    within = torch.arange(idx_b.shape[0], device=idx_b.device) - start[idx_b]
    out = torch.full((B, max(A, 1), 3), float("nan"), dtype=batch.pos.dtype, device=batch.pos.device)
    out[idx_b, within] = batch.pos[m]
    return out


def gather_sites(batch, world: int) -> torch.Tensor:
    """All ranks' adsorbate sites, rank-major: [sum_r B_r, A_max, 3] (NaN padded)."""
    local = adsorbate_sites(batch)
    if world <= 1:
        return local
    import torch.distributed as dist

    if dist.get_backend() == "gloo" and local.is_cuda:  # test configuration: several ranks on one GPU
        local = local.cpu()
    dev = local.device
    meta = torch.tensor([local.shape[0], local.shape[1]], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    Bmax = int(max(int(m[0]) for m in metas))
    Amax = int(max(int(m[1]) for m in metas))
    padded = torch.full((Bmax, Amax, 3), float("nan"), dtype=local.dtype, device=dev)
    padded[: local.shape[0], : local.shape[1]] = local
    outs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(outs, padded)
    return torch.cat([o[: int(m[0])] for o, m in zip(outs, metas)], dim=0)
