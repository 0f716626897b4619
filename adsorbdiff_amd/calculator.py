"""Single-structure entry point — mirror of the reference's ``AdsorbDiffCalculator.run_diffusion``.

Reference: adsorbdiff/relaxation/calculator.py:180-210 (ASE ``Atoms`` -> ``AtomsToGraphs.convert`` with
``r_edges=False`` -> ``data_list_collater`` -> ``ml_diffuse`` -> ``batch_to_atoms``), plus
adsorbdiff/utils/atoms_to_graphs.py:130-199 and adsorbdiff/relaxation/ase_utils.py:19-48 for the two
conversions.  ASE is not a dependency: anything with the ASE ``Atoms`` getters works (duck typing); real
``ase.Atoms`` objects are returned when ``ase`` is importable, otherwise ``SimpleAtoms``.

The reference builds a whole trainer from a checkpoint's embedded config (calculator.py:85-128); that
control plane is out of scope — the calculator is given a ``PaiNN`` module (or a trainer-like object)
and the ``denoising_pos_params`` dict directly; ``load_checkpoint`` reads reference checkpoints.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from .data import Batch, Data, data_list_collater
from .ml_relaxation import ml_diffuse
from .scaling import ensure_fitted
from .trainer import DenoisingTrainer


class SimpleAtoms:
    """Minimal stand-in for ``ase.Atoms`` (getters used by the reference's converters)."""

    def __init__(self, numbers, positions, cell, tags=None, fixed=None, pbc=(True, True, True)):
        self.numbers = np.asarray(numbers, dtype=int)
        self.positions = np.asarray(positions, dtype=float)
        self.cell = np.asarray(cell, dtype=float).reshape(3, 3)
        self.tags = np.zeros(len(self.numbers), dtype=int) if tags is None else np.asarray(tags, dtype=int)
        self.fixed = np.zeros(len(self.numbers), dtype=bool) if fixed is None else np.asarray(fixed, dtype=bool)
        self.pbc = np.asarray(pbc, dtype=bool)

    def get_atomic_numbers(self):
        return self.numbers

    def get_positions(self):
        return self.positions

    def get_cell(self):
        return self.cell

    def get_tags(self):
        return self.tags

    def __len__(self):
        return len(self.numbers)


def atoms_to_data(atoms, sid=None) -> Data:
    """``AtomsToGraphs(r_edges=False, r_fixed=True, r_pbc=True).convert`` (atoms_to_graphs.py:130-199)."""
    pos = torch.tensor(np.asarray(atoms.get_positions()), dtype=torch.float32)
    n = pos.shape[0]
    data = Data(
        cell=torch.tensor(np.array(atoms.get_cell()), dtype=torch.float32).reshape(1, 3, 3),
        pos=pos,
        atomic_numbers=torch.tensor(np.asarray(atoms.get_atomic_numbers()), dtype=torch.float32),
        natoms=torch.tensor([n]),
        tags=torch.tensor(np.asarray(atoms.get_tags()), dtype=torch.long),
    )
    fixed = torch.zeros(n, dtype=torch.long)
    if hasattr(atoms, "constraints"):  # ase.Atoms: FixAtoms constraints carry .index
        for c in atoms.constraints:
            if c.__class__.__name__ == "FixAtoms":
                fixed[torch.as_tensor(np.asarray(c.index), dtype=torch.long)] = 1
    elif hasattr(atoms, "fixed"):
        fixed = torch.tensor(np.asarray(atoms.fixed), dtype=torch.long)
    data.fixed = fixed
    data.pbc = torch.tensor(np.asarray(getattr(atoms, "pbc", (True, True, True)), dtype=bool)).reshape(1, 3)
    data.sid = 0 if sid is None else sid
    return data


def batch_to_atoms(batch: Batch):
    """One atoms object per system (ase_utils.py:19-48)."""
    try:  # pragma: no cover - ase is not installed in the build image
        from ase import Atoms
        from ase.constraints import FixAtoms

        have_ase = True
    except Exception:
        have_ase = False
    out = []
    start = 0
    for b, n in enumerate(batch.natoms.tolist()):
        sl = slice(start, start + n)
        numbers = batch.atomic_numbers[sl].long().tolist()
        positions = batch.pos[sl].detach().cpu().numpy()
        tags = batch.tags[sl].long().tolist()
        cell = batch.cell[b].detach().cpu().numpy()
        fixed = batch.fixed[sl].long().tolist() if hasattr(batch, "fixed") else [0] * n
        if have_ase:  # pragma: no cover
            out.append(Atoms(numbers=numbers, positions=positions, tags=tags, cell=cell,
                             constraint=FixAtoms(mask=fixed), pbc=[True, True, True]))
        else:
            out.append(SimpleAtoms(numbers, positions, cell, tags, fixed))
        start += n
    return out


class AdsorbDiffCalculator:
    def __init__(self, model, denoising_pos_params: dict, device: str = "cuda:0", save_full_traj: bool = True,
                 seed: Optional[int] = 0, checkpoint_path: Optional[str] = None):
        self.trainer = model if hasattr(model, "predict_denoising") else DenoisingTrainer(model, device=device)
        self.config = {"optim": {"denoising_pos_params": dict(denoising_pos_params)},
                       "task": {"save_full_traj": save_full_traj}}
        self.device = device
        if checkpoint_path is not None:
            self.trainer.load_checkpoint(checkpoint_path)
        if seed is not None:
            torch.manual_seed(seed)

    def run_diffusion(self, atoms, trajectory: Optional[str] = None):
        """Sample an adsorbate placement for one structure (or a sequence of structures, batched).
        Returns the final structure(s)."""
        many = isinstance(atoms, Sequence) and not hasattr(atoms, "get_positions")
        structures = list(atoms) if many else [atoms]
        batch = data_list_collater([atoms_to_data(a, sid=i) for i, a in enumerate(structures)], otf_graph=True)
        ensure_fitted(self.trainer._unwrapped_model)
        relaxed = ml_diffuse(
            batch=batch, model=self.trainer,
            denoising_pos_params=self.config["optim"]["denoising_pos_params"], traj_dir=trajectory,
            save_full_traj=self.config["task"].get("save_full_traj", True), device=self.device, transform=None,
        )
        out = batch_to_atoms(relaxed)
        return out if many else out[0]
