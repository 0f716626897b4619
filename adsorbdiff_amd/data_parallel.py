"""Input side of the sharded runs: per-step batches for one process per GPU, balanced by atom count.

Mirror of ``BalancedBatchSampler`` + ``OCPCollater`` (adsorbdiff/datasets/data_parallel.py:23-200) and of the dataset
protocol of ``LmdbDataset`` (adsorbdiff/datasets/lmdb_dataset.py:30-263) as far as the sampling / training paths use
them: ``len``, ``[i] -> Data``, optional ``metadata`` sizes.

Design difference: the reference all-gathers every rank's indices and sizes each step before it partitions them
(``distutils.all_gather``, :186-190).  The per-rank index streams of a ``DistributedSampler`` are a pure function of
(seed, epoch, rank), so here every rank derives ALL ranks' indices for the step locally and runs the same greedy
partition — the same batches with no collective on the input path (xGMI stays free for the gradient all-reduce).
"""
from __future__ import annotations

import math
import pickle
from pathlib import Path
from typing import Iterator, List, Optional, Sequence

import numpy as np
import torch

from .data import Batch, Data, data_list_collater


def balanced_partition_ref(sizes, num_parts: int) -> List[List[int]]:
    """The reference's partition with the reference's OUTPUT ORDER (datasets/data_parallel.py:32-48): descending sizes by
    ``np.argsort(-sizes)``, a heap of (load, members) entries — ties between equal loads are broken by comparing the
    member lists — and the parts returned in heap-array order, members in insertion order; rank r takes entry r.
    (``sampler.balanced_partition`` deals sampling shards with a deterministic index tie-break and sorted members; which
    rank gets which shard does not change any result there.)"""
    import heapq

    sizes = np.asarray(sizes)
    order = np.argsort(-sizes)
    heap = [(sizes[i], [i]) for i in order[:num_parts]]
    heapq.heapify(heap)
    for i in order[num_parts:]:
        load, members = heapq.heappop(heap)
        heapq.heappush(heap, (load + sizes[i], members + [i]))
    return [[int(j) for j in members] for _, members in heap]


class OCPCollater:
    def __init__(self, otf_graph: bool = True) -> None:
        self.otf_graph = otf_graph

    def __call__(self, data_list: List[Data]) -> Batch:
        return data_list_collater(data_list, otf_graph=self.otf_graph)


def distributed_indices(n: int, num_replicas: int, rank: int, shuffle: bool, seed: int, epoch: int,
                        drop_last: bool = False) -> List[int]:
    """torch.utils.data.DistributedSampler's index stream (same generator use, padding and striding)."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    if drop_last and n % num_replicas != 0:
        num_samples = math.ceil((n - num_replicas) / num_replicas)
    else:
        num_samples = math.ceil(n / num_replicas)
    total = num_samples * num_replicas
    if not drop_last:
        pad = total - len(idx)
        if pad <= len(idx):
            idx += idx[:pad]
        else:
            idx += (idx * math.ceil(pad / len(idx)))[:pad]
    else:
        idx = idx[:total]
    return idx[rank:total:num_replicas]


class BalancedBatchSampler:
    """Yields, per step, this rank's share of the step's global batch (``batch_size`` samples per rank), dealt by size
    (atoms, or any per-sample cost in ``sizes``) so that every rank gets about the same work."""

    def __init__(self, sizes: Sequence[int], batch_size: int, num_replicas: int, rank: int, mode="atoms",
                 shuffle: bool = True, drop_last: bool = False, seed: int = 0) -> None:
        if mode is True:
            mode = "atoms"
        if isinstance(mode, str) and mode.lower() not in ("atoms", "neighbors"):
            raise ValueError(f"Invalid mode {mode}. Must be one of 'atoms', 'neighbors', or a boolean.")
        self.sizes = np.asarray(sizes, dtype=np.int64)
        self.batch_size, self.num_replicas, self.rank = batch_size, num_replicas, rank
        self.shuffle, self.drop_last, self.seed, self.epoch = shuffle, drop_last, seed, 0
        self.balance_batches = num_replicas > 1 and mode is not False

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def _rank_batches(self, rank: int) -> List[List[int]]:
        idx = distributed_indices(len(self.sizes), self.num_replicas, rank, self.shuffle, self.seed, self.epoch, self.drop_last)
        out = [idx[i : i + self.batch_size] for i in range(0, len(idx), self.batch_size)]
        if self.drop_last and out and len(out[-1]) < self.batch_size:
            out.pop()
        return out

    def __len__(self) -> int:
        return len(self._rank_batches(self.rank))

    def __iter__(self) -> Iterator[List[int]]:
        mine = self._rank_batches(self.rank)
        if not self.balance_batches:
            yield from mine
            return
        per_rank = [self._rank_batches(r) for r in range(self.num_replicas)]
        for step in range(len(mine)):
            idx_all = [i for r in range(self.num_replicas) for i in per_rank[r][step]]
            parts = balanced_partition_ref(np.array([int(self.sizes[i]) for i in idx_all]), self.num_replicas)
            yield [idx_all[j] for j in parts[self.rank]]


class RecordDataset:
    """Systems stored by ``handoff.write_final_frames`` (``.npz`` records) or in a reference-style LMDB of pickled
    objects with attributes pos / cell / atomic_numbers / natoms / tags / fixed / sid (needs the ``lmdb`` package)."""

    def __init__(self, path) -> None:
        self.path = Path(path)
        self._env = None
        if self.path.suffix == ".npz":
            self._z = np.load(self.path, allow_pickle=False)
            self._n = int(self._z["length"])
        else:  # pragma: no cover - lmdb is not installed in the build image
            import lmdb  # type: ignore

            self._env = lmdb.open(str(self.path), subdir=False, readonly=True, lock=False, readahead=True, meminit=False,
                                  max_readers=1)
            with self._env.begin() as txn:
                length = txn.get(b"length")
                self._n = pickle.loads(length) if length is not None else self._env.stat()["entries"]

    def __len__(self) -> int:
        return self._n

    @property
    def natoms(self) -> np.ndarray:
        return np.array([int(self[i].natoms) for i in range(len(self))])

    def __getitem__(self, i: int) -> Data:
        if self._env is None:
            g = lambda k: self._z[f"{i}/{k}"]  # noqa: E731
            return Data(pos=torch.from_numpy(g("pos")).float(), atomic_numbers=torch.from_numpy(g("atomic_numbers")).float(),
                        tags=torch.from_numpy(g("tags")).long(), fixed=torch.from_numpy(g("fixed")).long(),
                        cell=torch.from_numpy(g("cell")).float().reshape(1, 3, 3), natoms=torch.tensor([int(g("natoms"))]),
                        sid=str(g("sid")))
        with self._env.begin() as txn:  # pragma: no cover
            rec = pickle.loads(txn.get(f"{i}".encode("ascii")))
        get = (lambda k: rec[k]) if isinstance(rec, dict) else (lambda k: getattr(rec, k))
        return Data(pos=torch.as_tensor(get("pos")).float(), atomic_numbers=torch.as_tensor(get("atomic_numbers")).float(),
                    tags=torch.as_tensor(get("tags")).long(), fixed=torch.as_tensor(get("fixed")).long(),
                    cell=torch.as_tensor(get("cell")).float().reshape(1, 3, 3),
                    natoms=torch.tensor([int(torch.as_tensor(get("natoms")).reshape(-1)[0])]), sid=str(get("sid")))
