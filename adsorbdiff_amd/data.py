"""Attribute-bag ``Data`` / ``Batch`` containers for adsorbate+slab systems.

The reference consumes ``torch_geometric.data.Batch`` objects produced by
``data_list_collater`` (reference: adsorbdiff/datasets/lmdb_dataset.py:233-263).
torch_geometric is not a dependency of this package; everything on the
sampling path only touches attributes (``pos, atomic_numbers, tags, batch,
natoms, cell, fixed, sid``), so a PyG ``Batch`` and this ``Batch`` are
interchangeable for every entry point here (duck typing).
"""
from __future__ import annotations

from typing import Iterable, List

import torch

_TENSOR_NODE_KEYS = ("pos", "atomic_numbers", "tags", "fixed", "force")
_TENSOR_GRAPH_KEYS = ("cell", "natoms", "y", "energy", "pbc")


class Data:
    """One adsorbate+slab system (reference schema: SURVEY.md §3.3)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __contains__(self, key) -> bool:
        return key in self.__dict__

    def keys(self):
        return list(self.__dict__.keys())

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                self.__dict__[k] = v.to(device)
        return self

    def clone(self):
        out = self.__class__()
        for k, v in self.__dict__.items():
            out.__dict__[k] = v.clone() if torch.is_tensor(v) else (list(v) if isinstance(v, list) else v)
        return out


class Batch(Data):
    """A collated batch: node tensors concatenated, ``batch`` = system index per atom."""

    @property
    def num_graphs(self) -> int:
        return int(self.natoms.shape[0])

    @staticmethod
    def from_data_list(data_list: Iterable[Data]) -> "Batch":
        data_list = list(data_list)
        # ml_diffuse hands back a list of already-collated batches
        # (reference: adsorbdiff/relaxation/ml_relaxation.py:167).
        flat: List[Data] = []
        for d in data_list:
            if isinstance(d, Batch):
                flat.extend(d.to_data_list())
            else:
                flat.append(d)
        out = Batch()
        node: dict = {}
        graph: dict = {}
        sids = []
        for d in flat:
            for k in _TENSOR_NODE_KEYS:
                if k in d:
                    node.setdefault(k, []).append(getattr(d, k))
            for k in _TENSOR_GRAPH_KEYS:
                if k in d:
                    v = getattr(d, k)
                    if k == "cell":
                        v = v.reshape(-1, 3, 3)
                    elif k == "pbc":
                        v = torch.atleast_2d(v)
                    else:
                        v = v.reshape(-1)
                    graph.setdefault(k, []).append(v)
            if "sid" in d:
                sids.append(d.sid)
        for k, v in node.items():
            setattr(out, k, torch.cat(v, dim=0))
        for k, v in graph.items():
            setattr(out, k, torch.cat(v, dim=0))
        natoms = torch.tensor([int(d.pos.shape[0]) for d in flat], dtype=torch.long)
        out.natoms = natoms.to(out.pos.device)
        out.batch = torch.repeat_interleave(
            torch.arange(len(flat), device=out.pos.device), out.natoms
        )
        out.sid = sids
        return out

    def to_data_list(self) -> List[Data]:
        sizes = self.natoms.tolist()
        out = []
        start = 0
        for b, n in enumerate(sizes):
            d = Data()
            for k in _TENSOR_NODE_KEYS:
                if k in self:
                    setattr(d, k, getattr(self, k)[start : start + n])
            for k in _TENSOR_GRAPH_KEYS:
                if k in self and k != "natoms":
                    setattr(d, k, getattr(self, k)[b : b + 1])
            d.natoms = torch.tensor([n], dtype=torch.long, device=self.pos.device)
            if "sid" in self:
                d.sid = self.sid[b]
            out.append(d)
            start += n
        return out


def data_list_collater(data_list, otf_graph: bool = True) -> Batch:
    """Reference: adsorbdiff/datasets/lmdb_dataset.py:233-263 (otf_graph only)."""
    return Batch.from_data_list(data_list)
