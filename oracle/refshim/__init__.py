"""TEST INFRASTRUCTURE — container-only import harness for the *reference* AdsorbDiff code.

The reference (``/root/reference``) depends on packages that cannot be installed
here (torch_scatter, torch_geometric, ase, lmdb).  ``install()`` pre-seeds
``sys.modules`` with minimal stand-ins so that the reference's own PaiNN
denoiser (adsorbdiff/models/painn/painn_denoising.py) and reverse-SDE stepper
(adsorbdiff/relaxation/diffusers/denoising_torch.py) can be imported and run on
CPU to generate golden vectors (``oracle/make_golden.py``).

Nothing here travels to the GPU box in a way that matters: ``/root/reference``
does not exist there, so ``install()`` is only ever called by
``oracle/make_golden.py`` in the build container.

Stand-in semantics (these ARE the definition used for the goldens, since the
reference holds no tests that pin them):
  * torch_scatter.scatter / segment_coo / segment_csr: plain indexed
    sum / mean / min / max on top of torch.scatter_add_ / scatter_reduce_.
  * torch_geometric.nn.MessagePassing.propagate: ``*_j`` arguments are gathered
    with edge_index[0], ``*_i`` with edge_index[1]; aggregation index is
    edge_index[1] (flow source_to_target, node_dim=0), as documented by PyG.
  * ase: I/O sink only (no arithmetic).
"""
from __future__ import annotations

import inspect
import os
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def _scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    assert dim == 0 and out is None
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    shape = (int(dim_size),) + tuple(src.shape[1:])
    idx = index.reshape(-1, *([1] * (src.dim() - 1))).expand_as(src)
    if reduce in ("sum", "add"):
        return torch.zeros(shape, dtype=src.dtype).scatter_add_(0, idx, src)
    if reduce == "mean":
        total = torch.zeros(shape, dtype=src.dtype).scatter_add_(0, idx, src)
        cnt = torch.zeros(int(dim_size), dtype=src.dtype).scatter_add_(
            0, index, torch.ones_like(index, dtype=src.dtype)
        )
        cnt = cnt.clamp(min=1)
        return total / cnt.reshape(-1, *([1] * (src.dim() - 1)))
    if reduce == "min":
        return torch.zeros(shape, dtype=src.dtype).scatter_reduce_(0, idx, src, "amin", include_self=False)
    if reduce == "max":
        return torch.zeros(shape, dtype=src.dtype).scatter_reduce_(0, idx, src, "amax", include_self=False)
    raise NotImplementedError(reduce)


def _segment_coo(src, index, out=None, dim_size=None, reduce="sum"):
    return _scatter(src, index, 0, dim_size=dim_size, reduce=reduce)


def _segment_csr(src, indptr, out=None, reduce="sum"):
    n = indptr.numel() - 1
    counts = indptr[1:] - indptr[:-1]
    index = torch.repeat_interleave(torch.arange(n), counts)
    return _scatter(src[int(indptr[0]) : int(indptr[-1])], index, 0, dim_size=n, reduce=reduce)


class _MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", node_dim=0, **kw):
        super().__init__()
        self.aggr = aggr
        self.node_dim = node_dim

    def jittable(self):
        return self

    def propagate(self, edge_index, size=None, **kwargs):
        src, dst = edge_index[0], edge_index[1]
        n_nodes = None
        msg_kwargs = {}
        for name in inspect.signature(self.message).parameters:
            if name.endswith("_j"):
                t = kwargs[name[:-2]]
                n_nodes = t.shape[0]
                msg_kwargs[name] = t.index_select(0, src)
            elif name.endswith("_i"):
                t = kwargs[name[:-2]]
                n_nodes = t.shape[0]
                msg_kwargs[name] = t.index_select(0, dst)
            else:
                msg_kwargs[name] = kwargs[name]
        out = self.message(**msg_kwargs)
        out = self.aggregate(out, dst, None, n_nodes)
        return self.update(out)


def install():
    """Pre-seed sys.modules; returns nothing.  Idempotent."""
    if "torch_scatter" in sys.modules and getattr(sys.modules["torch_scatter"], "_adf_standin", False):
        return
    sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference

    from adsorbdiff_amd.data import Batch, Data

    ts = types.ModuleType("torch_scatter")
    ts._adf_standin = True
    ts.scatter, ts.segment_coo, ts.segment_csr = _scatter, _segment_coo, _segment_csr
    sys.modules["torch_scatter"] = ts

    tg = types.ModuleType("torch_geometric")
    tg.__version__ = "2.4.0"
    tg_nn = types.ModuleType("torch_geometric.nn")
    tg_nn.MessagePassing = _MessagePassing
    tg_nn.radius_graph = None
    tg_data = types.ModuleType("torch_geometric.data")
    tg_data.Data, tg_data.Batch = Data, Batch
    tg_dd = types.ModuleType("torch_geometric.data.data")
    tg_dd.BaseData = Data
    tg_utils = types.ModuleType("torch_geometric.utils")
    tg_utils.remove_self_loops = None

    def _segment_softmax(src, index, ptr=None, num_nodes=None, dim=0):
        """torch_geometric.utils.softmax: softmax of src over the entries that share an index (EqV2 attention over the
        incoming edges of a node, transformer_block.py:340).  Documented PyG semantics: subtract the group max, exp,
        divide by the group sum (+1e-16)."""
        assert dim == 0
        n = int(index.max()) + 1 if num_nodes is None else num_nodes
        idx = index.reshape(-1, *([1] * (src.dim() - 1))).expand_as(src)
        mx = torch.full((n,) + tuple(src.shape[1:]), float("-inf"), dtype=src.dtype, device=src.device)
        mx = mx.scatter_reduce(0, idx, src, reduce="amax", include_self=True)
        ex = (src - mx.gather(0, idx)).exp()
        den = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device).scatter_add_(0, idx, ex)
        return ex / (den.gather(0, idx) + 1e-16)

    tg_utils.softmax = _segment_softmax
    tg.nn, tg.data, tg.utils = tg_nn, tg_data, tg_utils
    for name, mod in (
        ("torch_geometric", tg),
        ("torch_geometric.nn", tg_nn),
        ("torch_geometric.data", tg_data),
        ("torch_geometric.data.data", tg_dd),
        ("torch_geometric.utils", tg_utils),
    ):
        sys.modules[name] = mod

    # ase: trajectory sink only
    ase = types.ModuleType("ase")

    class Atoms:
        def __init__(self, **kw):
            self.__dict__.update(kw)

        def set_calculator(self, calc):
            self.calc = calc

    class Trajectory:
        def __init__(self, path, mode="r"):
            self.path, self.frames = path, []

        def write(self, atoms):
            self.frames.append(atoms)

        def close(self):
            with open(self.path, "w") as f:
                f.write(str(len(self.frames)))

    ase.Atoms = Atoms
    ase_io = types.ModuleType("ase.io")
    ase_io.Trajectory = Trajectory
    ase.io = ase_io
    ase_calc = types.ModuleType("ase.calculators")
    ase_sp = types.ModuleType("ase.calculators.singlepoint")
    ase_sp.SinglePointCalculator = lambda **kw: kw
    ase_con = types.ModuleType("ase.constraints")
    ase_con.FixAtoms = lambda mask=None: mask
    for name, mod in (
        ("ase", ase),
        ("ase.io", ase_io),
        ("ase.calculators", ase_calc),
        ("ase.calculators.singlepoint", ase_sp),
        ("ase.constraints", ase_con),
    ):
        sys.modules[name] = mod
    lmdb = types.ModuleType("lmdb")
    lmdb.Environment = object
    sys.modules["lmdb"] = lmdb

    # bypass adsorbdiff/__init__.py (pulls ase/pymatgen/lmdb via relaxation/calculator.py)
    pkg = types.ModuleType("adsorbdiff")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "adsorbdiff")]
    sys.modules["adsorbdiff"] = pkg


class FakeTrainer:
    """The 3 things DiffTorchCalc / Denoiser touch on a trainer
    (reference: adsorbdiff/relaxation/diffusers/denoising_torch.py:38,491-500)."""

    def __init__(self, model):
        self.model = model
        self._unwrapped_model = model

    @torch.no_grad()
    def predict_denoising(self, batch, per_image=False, disable_tqdm=True):
        out1, out2 = self.model(batch)
        return {"positions": out1.detach(), "positions_free": out2.detach()}
