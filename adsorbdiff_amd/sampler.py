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
    """This rank's share of ``batch`` (systems dealt by atom count); returns (sub_batch, system ids) - (None, []) for a rank
    that is dealt nothing (more ranks than systems): it skips the sampling and only joins ``gather_sites``."""
    from .data import Batch

    parts = balanced_partition(batch.natoms.tolist(), world)
    mine = parts[rank]
    if not mine:
        return None, []
    data = batch.to_data_list()
    return Batch.from_data_list([data[i] for i in mine]), mine


def _exchange_device(via: str = "torch") -> torch.device:
    """Where a rank WITHOUT systems must put its (empty) message so that the collective accepts it: the current ROCm device
    under backend nccl (= RCCL) or ``via="rccl"`` - every other rank is already waiting in the all-gather with a device
    tensor, and a CPU tensor there raises on this rank while the others block until the NCCL time-out -, the host
    otherwise (gloo)."""
    on_dev = via == "rccl"
    if not on_dev:
        import torch.distributed as dist

        on_dev = dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
    if on_dev and torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def adsorbate_sites(batch, device=None) -> torch.Tensor:
    """[B, A, 3] positions of each system's adsorbate (tag==2) atoms, NaN-padded to the largest
    adsorbate in the batch.  A rank that was dealt no system (more ranks than systems) passes None or an empty batch and
    gets a [0, 1, 3] tensor on ``device`` (default: where the exchange of this process group runs, ``_exchange_device``):
    it still takes part in the exchange with an all-padding message."""
    if batch is None or not hasattr(batch, "pos"):
        return torch.empty(0, 1, 3, dtype=torch.float32, device=device if device is not None else _exchange_device())
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


def shard_bounds(batch, world: int):
    """(B_max, A_max) over all shards of ``shard_batch(batch, r, world)``: the largest number of systems on a rank and the
    largest adsorbate.  Every rank computes it locally from the global batch (the partition is deterministic), so the
    exchange buffer of ``gather_sites`` can be sized without a collective."""
    parts = balanced_partition(batch.natoms.tolist(), world)
    B = int(batch.natoms.shape[0])
    counts = torch.bincount(batch.batch[batch.tags == 2], minlength=B)
    return max(len(p) for p in parts), max(int(counts.max().item()), 1)


def gather_sites(batch, world: int, via: str = "torch", system_ids=None, bounds=None, local=None, device=None) -> torch.Tensor:
    """All ranks' adsorbate sites: [sum_r B_r, A_max, 3] (NaN padded), rank-major — or, when every rank passes the
    global ids of its systems (``system_ids``, what ``shard_batch`` returns), in global system order.

    ONE collective when ``bounds = (B_max, A_max)`` is given (``shard_bounds``: every rank derives it locally): each rank
    packs [B_max, 1 + 3 A_max] floats — the system id (bit pattern of an int32; -1 = padding row) and the NaN-padded
    sites — and one all_gather moves them.  Without bounds the shapes are agreed on first (one more small all_gather).

    via="torch": ``torch.distributed.all_gather`` (backend nccl = RCCL over xGMI; gloo in the CPU tests);
    via="rccl":  the library's C-ABI entry ``adf_allgather_sites`` (one GPU per rank)."""
    if local is None:   # (``local``: the sites already extracted)
        # a rank without systems (batch None): an empty message on ``device``, or where this group's collective runs
        local = adsorbate_sites(batch, device if device is not None else _exchange_device(via))
    if world <= 1:
        return local
    import torch.distributed as dist

    host_bounce = via != "rccl" and dist.get_backend() == "gloo" and local.is_cuda  # test configuration
    if host_bounce:
        local = local.cpu()
    dev = local.device

    def all_gather(t):
        if via == "rccl":
            return RcclGather.get(dev).all_gather(t)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return torch.stack(outs)

    if bounds is None:
        meta = all_gather(torch.tensor([local.shape[0], local.shape[1]], dtype=torch.int64, device=dev)).cpu()
        bounds = (int(meta[:, 0].max()), int(meta[:, 1].max()))
    Bmax, Amax = int(bounds[0]), int(bounds[1])
    packed = pack_sites(local, system_ids, (Bmax, Amax))
    everything = all_gather(packed.contiguous())                       # the one exchange: [world, B_max, 1 + 3 A_max]
    return merge_packed_sites(everything, Amax, ordered=system_ids is not None).to(local.dtype)


def pack_sites(local: torch.Tensor, system_ids, bounds) -> torch.Tensor:
    """One rank's message of the exchange: ``[B_max, 1 + 3 A_max]`` INT32 words - the system id as it is (-1 = padding row)
    and the sites as their float32 bit patterns (a float -> int reinterpretation is lossless on every copy path, whereas ids
    stored as float bits would be denormals / NaN payloads that a flush-to-zero or NaN-canonicalising copy could alter);
    padding sites are NaN bit patterns."""
    Bmax, Amax = int(bounds[0]), int(bounds[1])
    dev = local.device
    if local.shape[0] > Bmax or local.shape[1] > Amax:
        raise ValueError(f"gather_sites: local sites {tuple(local.shape)} exceed the bounds {(Bmax, Amax)}")
    ids = torch.arange(local.shape[0], dtype=torch.int32) if system_ids is None else torch.as_tensor(system_ids, dtype=torch.int32)
    if ids.numel() != local.shape[0]:
        raise ValueError("every rank must pass one id per local system")
    nan_bits = int(torch.tensor([float("nan")], dtype=torch.float32).view(torch.int32)[0])
    packed = torch.full((Bmax, 1 + 3 * Amax), nan_bits, dtype=torch.int32, device=dev)
    packed[:, 0] = -1
    packed[: local.shape[0], 0] = ids.to(dev)
    packed[: local.shape[0], 1 : 1 + 3 * local.shape[1]] = (
        local.reshape(local.shape[0], 3 * local.shape[1]).to(torch.float32).contiguous().view(torch.int32))   # (a rank may be empty)
    return packed


def merge_packed_sites(everything: torch.Tensor, Amax: int, ordered: bool = True) -> torch.Tensor:
    """All ranks' messages ``[world, B_max, 1 + 3 A_max]`` -> sites ``[B, A_max, 3]`` (float32), in global system order when
    the ranks sent global ids (``ordered``), rank-major otherwise."""
    world, Bmax = everything.shape[0], everything.shape[1]
    gid = everything[:, :, 0]
    keep = gid >= 0
    sites = everything[:, :, 1:].contiguous().view(torch.float32).reshape(world, Bmax, Amax, 3)[keep]
    if not ordered:
        return sites                                                   # rank-major
    return sites[torch.argsort(gid[keep].to(torch.int64))]
