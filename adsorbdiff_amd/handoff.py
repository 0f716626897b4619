"""Hand-off of the sampled sites to the next stage: what the reference does with the per-system
``.traj`` files in ``scripts/create_lmdbs/pred_traj_to_lmdb.py:52-105`` — take the final frame, lift the adsorbate when
it ended up less than 0.1 A above the surface, and store one record per system for the GemNet-OC relaxer.

Here the lift runs on the device on the whole batch (``adf_lift_adsorbates``) and the records are written once per
batch: a real LMDB (pickled attribute dicts under the reference's keys) when ``lmdb`` is importable, otherwise an
``.npz`` with the same fields.  The reference's ASE ``Trajectory`` files need the ``ase`` package (not installable
here): with it ``Denoiser`` writes genuine ``.traj`` files, without it the per-system sink is named ``<sid>.npz``.
"""
from __future__ import annotations

import ctypes as C
import pickle
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from . import lib as _lib


def lift_adsorbates(batch, min_gap: float = 0.1) -> torch.Tensor:
    """In place on ``batch.pos`` (device): per system, if min z(tag 2) - max z(tag 1) < min_gap, shift the adsorbate up
    by |diff| + min_gap.  Returns the applied shift per system [B]."""
    if not batch.pos.is_cuda:
        raise RuntimeError("lift_adsorbates runs on a ROCm device (no CPU fallback)")
    lib = _lib.load()
    dev = batch.pos.device
    natoms = batch.natoms.to(dev, torch.int64).reshape(-1)
    off = torch.zeros(natoms.shape[0] + 1, dtype=torch.int32, device=dev)
    off[1:] = torch.cumsum(natoms, 0).to(torch.int32)
    tags = batch.tags.to(dev, torch.int32).contiguous()
    pos = batch.pos if (batch.pos.dtype == torch.float32 and batch.pos.is_contiguous()) else batch.pos.float().contiguous()
    lifted = torch.empty(natoms.shape[0], dtype=torch.float32, device=dev)
    _lib.check(lib.adf_lift_adsorbates(pos.data_ptr(), tags.data_ptr(), off.data_ptr(), int(natoms.shape[0]),
                                       C.c_float(min_gap), lifted.data_ptr(),
                                       C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    batch.pos = pos
    return lifted


def write_final_frames(batch, path, fid: int = 0, start_index: int = 0, apply_lift: bool = True) -> Path:
    """One record per system: pos, cell, atomic_numbers, natoms, tags, fixed, sid, fid (the fields the reference's
    a2g.convert + converter attach, pred_traj_to_lmdb.py:76-92)."""
    if apply_lift:
        lift_adsorbates(batch)
    path = Path(path)
    data = batch.to("cpu").to_data_list()
    records = []
    for d in data:
        records.append({"pos": d.pos.numpy(), "cell": d.cell.reshape(1, 3, 3).numpy(), "atomic_numbers": d.atomic_numbers.numpy(),
                        "natoms": int(d.natoms), "tags": d.tags.numpy(), "fixed": d.fixed.numpy(), "sid": str(d.sid),
                        "fid": fid})
    try:
        import lmdb  # type: ignore
    except Exception:
        lmdb = None
    if lmdb is not None:  # pragma: no cover - lmdb is not installed in the build image
        env = lmdb.open(str(path), map_size=1 << 40, subdir=False, meminit=False, map_async=True)
        with env.begin(write=True) as txn:
            for i, r in enumerate(records):
                txn.put(f"{start_index + i}".encode("ascii"), pickle.dumps(r, protocol=-1))
            txn.put(b"length", pickle.dumps(start_index + len(records), protocol=-1))
        env.sync()
        env.close()
        return path
    out = path.with_suffix(".npz")
    entries = {}
    if start_index > 0:
        # per-batch calls append: keep the records already in the file (np.savez rewrites the whole archive)
        if not out.exists():
            raise FileNotFoundError(f"{out}: start_index={start_index} but the records of the earlier batches are missing")
        with np.load(out, allow_pickle=False) as old:
            have = int(old["length"])
            if have != start_index:
                raise ValueError(f"{out} holds {have} records, start_index={start_index} would leave a gap or overwrite")
            entries.update({k: old[k] for k in old.files if k != "length"})
    entries.update({f"{start_index + i}/{k}": v for i, r in enumerate(records) for k, v in r.items()})
    tmp = out.with_suffix(".tmp.npz")
    np.savez(tmp, **entries, length=start_index + len(records))
    tmp.replace(out)
    return out
