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


class RcclGather:
    """The exchange step through the library's own C-ABI (``adf_allgather_sites``, csrc/collect.hip): one RCCL
    communicator per rank, created once; the 128-byte unique id travels through the already initialised
    ``torch.distributed`` group.  Needs one GPU per rank (RCCL refuses two ranks on one device)."""

    _instance = None

    def __init__(self, device) -> None:
        import ctypes as C

        import torch.distributed as dist

        from . import lib as _lib

        self.lib, self.device = _lib.load(), torch.device(device)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        buf = (C.c_uint8 * 128)()
        if self.rank == 0:
            _lib.check(self.lib.adf_comm_unique_id(buf))
        on_dev = dist.get_backend() == "nccl"
        t = torch.tensor(list(buf), dtype=torch.uint8, device=self.device if on_dev else "cpu")
        dist.broadcast(t, src=0)
        buf = (C.c_uint8 * 128)(*t.cpu().tolist())
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_comm_create(buf, self.rank, self.world, C.byref(self.handle)))

    @classmethod
    def get(cls, device):
        if cls._instance is None or cls._instance.device != torch.device(device):
            cls._instance = cls(device)
        return cls._instance

    def all_gather(self, local: torch.Tensor) -> torch.Tensor:
        """[world, *local.shape] from equally shaped, contiguous device tensors."""
        import ctypes as C

        from . import lib as _lib

        assert local.is_cuda and local.is_contiguous()
        out = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        stream = C.c_void_p(torch.cuda.current_stream(local.device).cuda_stream)
        with torch.cuda.device(local.device):
            _lib.check(self.lib.adf_allgather_sites(self.handle, local.data_ptr(), local.numel() * local.element_size(),
                                                   out.data_ptr(), stream))
        return out

    def close(self) -> None:
        if self.handle:
            self.lib.adf_comm_destroy(self.handle)
            self.handle = None


def gather_sites(batch, world: int, via: str = "torch", system_ids=None) -> torch.Tensor:
    """All ranks' adsorbate sites: [sum_r B_r, A_max, 3] (NaN padded), rank-major — or, when every rank passes the
    global ids of its systems (``system_ids``, what ``shard_batch`` returns), in global system order.

    via="torch": ``torch.distributed.all_gather`` (backend nccl = RCCL over xGMI; gloo in the CPU tests);
    via="rccl":  the library's C-ABI entry ``adf_allgather_sites`` (one GPU per rank)."""
    local = adsorbate_sites(batch)
    if world <= 1:
        return local
    import torch.distributed as dist

    ids = torch.as_tensor(system_ids if system_ids is not None else [], dtype=torch.int64)
    if via == "rccl":
        g = RcclGather.get(local.device)
        meta = torch.tensor([local.shape[0], local.shape[1]], dtype=torch.int64, device=local.device)
        metas = g.all_gather(meta).cpu()
        Bmax, Amax = int(metas[:, 0].max()), int(metas[:, 1].max())
        padded = torch.full((Bmax, Amax, 3), float("nan"), dtype=local.dtype, device=local.device)
        padded[: local.shape[0], : local.shape[1]] = local
        outs = g.all_gather(padded)
        id_pad = torch.full((Bmax,), -1, dtype=torch.int64, device=local.device)
        id_pad[: ids.numel()] = ids.to(local.device)
        all_ids = g.all_gather(id_pad) if system_ids is not None else None
        counts = [int(m[0]) for m in metas]
    else:
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
        all_ids = None
        if system_ids is not None:
            id_pad = torch.full((Bmax,), -1, dtype=torch.int64, device=dev)
            id_pad[: ids.numel()] = ids.to(dev)
            all_ids = [torch.empty_like(id_pad) for _ in range(world)]
            dist.all_gather(all_ids, id_pad)
        counts = [int(m[0]) for m in metas]
    sites = torch.cat([outs[r][: counts[r]] for r in range(world)], dim=0)
    if all_ids is None:
        return sites
    gid = torch.cat([all_ids[r][: counts[r]] for r in range(world)], dim=0)
    assert bool((gid >= 0).all()), "every rank must pass one id per local system"
    order = torch.argsort(gid)
    return sites[order]
