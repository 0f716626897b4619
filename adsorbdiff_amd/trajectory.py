"""Asynchronous, batched trajectory sink of the sampler.

Reference: ``Denoiser.write`` (relaxation/diffusers/denoising_torch.py:469-477) converts the whole batch to ASE ``Atoms``
on the host after EVERY reverse step (``batch_to_atoms``, relaxation/ase_utils.py:19-48: one ``.cpu()`` per tensor and
system) and appends to one open ``ase.io.Trajectory`` per system (``:66-82``: ``<sid>.traj_tmp``, renamed to ``<sid>.traj``
at the end); ``check_traj_files`` (utils/utils.py:968-973) later skips a batch iff every ``<sid>.traj`` exists.

Here the sampling loop stays one library call (``adf_sample_traj``): the library snapshots the positions after a step and
copies them into a pinned host ring on its own stream (csrc/frames.hip); ``TrajectoryWriter`` is the host thread at the
other end of the ring.  It streams the frames of the whole batch into ONE file (``<traj_dir>/batch_<first sid>.frames.npy``,
``[frames, N, 3]`` float32, written while the GPU computes the next steps) and, once the run has told it how many steps were
applied, writes the per-system files the resume rule reads — ``<sid>.npz`` (positions ``[frames, n, 3]``, numbers, tags,
fixed, cell; temporary name first, then renamed) — plus ``batch_<first sid>.json`` (sids, atom offsets, frame count).

FORMAT.  ``ase`` cannot be installed in this image and its binary ULM trajectory layout cannot be tested here, so the
format of this sink is **.npz, stated plainly**; ``npz_to_ase_traj`` converts a ``<sid>.npz`` into the reference's
``<sid>.traj`` wherever ``ase`` is importable (tests/test_host_logic.py runs it under ``pytest.importorskip("ase")``).
"""
from __future__ import annotations

import ctypes as C
import json
import threading
from pathlib import Path
from typing import Optional

import numpy as np

from . import lib as _lib


class FrameSink:
    """Owner of an ``adf_frames_t``: pinned host ring + device staging + copy stream (csrc/frames.hip)."""

    def __init__(self, device_index: int, num_atoms: int, slots: int = 8):
        self.lib = _lib.load()
        self.num_atoms = int(num_atoms)
        self.slots = int(slots)
        h = C.c_void_p()
        _lib.check(self.lib.adf_frames_create(int(device_index), 3 * self.num_atoms, self.slots, C.byref(h)))
        self.handle = h

    def wait(self, index: int, timeout_ms: int = 100) -> Optional[np.ndarray]:
        """Frame ``index`` as a [N,3] view of the pinned ring (valid until ``release``), or None on time-out."""
        ptr = C.POINTER(C.c_float)()
        _lib.check(self.lib.adf_frames_wait(self.handle, int(index), int(timeout_ms), C.byref(ptr)))
        if not ptr:
            return None
        return np.ctypeslib.as_array(ptr, shape=(self.num_atoms, 3))

    def release(self, index: int) -> None:
        _lib.check(self.lib.adf_frames_release(self.handle, int(index)))

    def pushed(self) -> int:
        return int(self.lib.adf_frames_pushed(self.handle))

    def abort(self) -> None:
        """Cancel the ring: a sampler blocked in (or later calling) ``adf_frames_push`` gets an error instead of waiting
        for a slot nobody will release."""
        if self.handle:
            self.lib.adf_frames_abort(self.handle)

    def close(self) -> None:
        if self.handle:
            self.lib.adf_frames_destroy(self.handle)
            self.handle = None


class TrajectoryWriter(threading.Thread):
    """Host thread that drains a frame source into the batch file and, at the end, the per-system files.

    ``source``: anything with ``wait(index, timeout_ms) -> ndarray[N,3] | None`` and ``release(index)`` (a ``FrameSink``,
    or a fake in the CPU tests).  ``meta``: numbers [N], tags [N], fixed [N], cell [B,3,3], natoms [B], names [B].
    ``finish(n_frames)`` tells the writer how many of the pushed frames count (the early stop of the sampler ends a run
    before ``max_frames``; frames pushed after the stop are dropped); ``join()`` then returns once every file exists."""

    def __init__(self, source, traj_dir, meta: dict, max_frames: int, keep_last_only: bool = False):
        super().__init__(daemon=True, name="adsorbdiff-trajectory-writer")
        self.source = source
        self.traj_dir = Path(traj_dir)
        self.meta = meta
        self.max_frames = int(max_frames)
        self.keep_last_only = bool(keep_last_only)
        self.error: Optional[BaseException] = None
        self._aborted = False
        self._n_final: Optional[int] = None
        self._lock = threading.Lock()
        self.frames_written = 0
        names = [str(n) for n in meta["names"]]
        self.batch_stem = self.traj_dir / f"batch_{names[0]}"

    def finish(self, n_frames: int) -> None:
        with self._lock:
            self._n_final = int(n_frames)

    def abort(self) -> None:
        """The run failed: drain nothing more, delete the temporary files and publish NOTHING (the reference renames
        ``.traj_tmp`` to ``.traj`` only after a completed run, denoising_torch.py:66-82, and ``check_traj_files`` treats
        every existing ``<sid>`` file as a finished system)."""
        with self._lock:
            self._aborted = True
            self._n_final = 0

    def _final(self) -> Optional[int]:
        with self._lock:
            return self._n_final

    def _is_aborted(self) -> bool:
        with self._lock:
            return self._aborted

    def _cleanup_tmp(self) -> None:
        # only THIS writer's temporary files: traj_dir is shared (one directory, files named by sid, as in the reference
        # layout), and another batch's writer - or another rank - may be publishing its own `<sid>.npz_tmp` right now
        own = [self.traj_dir / f"{name}.npz_tmp" for name in (str(n) for n in self.meta["names"])]
        for pth in own + [self.batch_stem.with_suffix(".frames.npy_tmp"), self.batch_stem.with_suffix(".json_tmp")]:
            try:
                pth.unlink()
            except OSError:
                pass

    def run(self) -> None:
        try:
            self._run()
        except BaseException as e:  # surfaced by join_checked()
            self.error = e
            # the sampler must not wait for ring slots this thread will never release
            if hasattr(self.source, "abort"):
                try:
                    self.source.abort()
                except Exception:
                    pass
            self._cleanup_tmp()

    def _run(self) -> None:
        self.traj_dir.mkdir(exist_ok=True, parents=True)
        N = int(np.asarray(self.meta["numbers"]).shape[0])
        tmp_frames = self.batch_stem.with_suffix(".frames.npy_tmp")
        mm = np.lib.format.open_memmap(tmp_frames, mode="w+", dtype=np.float32, shape=(self.max_frames, N, 3))
        got = 0
        while True:
            fin = self._final()
            if fin is not None and got >= min(fin, self.max_frames):
                break
            if got >= self.max_frames:
                if fin is not None:
                    break
                threading.Event().wait(0.002)
                continue
            frame = self.source.wait(got, 50)
            if frame is None:
                continue
            mm[got] = frame          # the only copy on the host: pinned ring -> page cache of the batch file
            self.source.release(got)
            got += 1
            self.frames_written = got
        n = min(self._final(), got)
        if self._is_aborted() or n <= 0:
            # nothing to publish: an aborted run, or a run that applied no step (never a <sid>.npz with zero frames: the
            # resume rule would count the system as done)
            del mm
            self._cleanup_tmp()
            if hasattr(self.source, "abort") and self._is_aborted():
                self.source.abort()
            return
        # frames pushed but not counted (after an early stop): give their slots back so that the ring never blocks
        k = got
        while True:
            extra = self.source.wait(k, 1) if hasattr(self.source, "pushed") and self.source.pushed() > k else None
            if extra is None:
                break
            self.source.release(k)
            k += 1
        mm.flush()
        frames = np.asarray(mm[:n]) if not self.keep_last_only else np.asarray(mm[max(n - 1, 0):n])
        natoms = [int(v) for v in np.asarray(self.meta["natoms"]).reshape(-1)]
        names = [str(v) for v in self.meta["names"]]
        numbers, tags, fixed = (np.asarray(self.meta[k]) for k in ("numbers", "tags", "fixed"))
        cell = np.asarray(self.meta["cell"]).reshape(-1, 3, 3)
        start = 0
        for b, (na, name) in enumerate(zip(natoms, names)):
            sl = slice(start, start + na)
            tmp = self.traj_dir / f"{name}.npz_tmp"
            with open(tmp, "wb") as fh:
                np.savez(fh, positions=frames[:, sl], numbers=numbers[sl], tags=tags[sl], fixed=fixed[sl], cell=cell[b])
            tmp.rename(self.traj_dir / f"{name}.npz")
            start += na
        # the batch file: trimmed to the frames that count, final name last (a reader never sees a half-written one)
        del mm
        final_frames = self.batch_stem.with_suffix(".frames.npy")
        if frames.shape[0] == self.max_frames and not self.keep_last_only:
            tmp_frames.rename(final_frames)
        else:
            np.save(final_frames, frames)
            tmp_frames.unlink()
        offs = np.concatenate([[0], np.cumsum(natoms)]).tolist()
        with open(self.batch_stem.with_suffix(".json_tmp"), "w") as fh:
            json.dump({"sids": names, "atom_offsets": offs, "frames": int(frames.shape[0]),
                       "frames_file": final_frames.name, "format": "npy float32 [frames, atoms, 3]"}, fh)
        self.batch_stem.with_suffix(".json_tmp").rename(self.batch_stem.with_suffix(".json"))

    def join_checked(self, timeout: Optional[float] = None) -> None:
        self.join(timeout)
        if self.is_alive():
            raise TimeoutError("trajectory writer still running")
        if self.error is not None:
            raise self.error


def npz_to_ase_traj(npz_path, traj_path=None):
    """``<sid>.npz`` of this sink -> the reference's ``<sid>.traj`` (one ASE ``Atoms`` per frame with tags, cell, pbc and a
    ``FixAtoms`` constraint: what relaxation/ase_utils.py:19-48 builds and denoising_torch.py:469-477 writes).  Needs
    ``ase``; returns the path written."""
    from ase import Atoms
    from ase.constraints import FixAtoms
    from ase.io import Trajectory

    npz_path = Path(npz_path)
    traj_path = Path(traj_path) if traj_path is not None else npz_path.with_suffix(".traj")
    with np.load(npz_path, allow_pickle=False) as z:
        pos, numbers, tags, fixed, cell = z["positions"], z["numbers"], z["tags"], z["fixed"], z["cell"]
    tmp = traj_path.with_suffix(".traj_tmp")
    with Trajectory(str(tmp), mode="w") as traj:
        for f in range(pos.shape[0]):
            traj.write(Atoms(numbers=numbers.astype(int), positions=pos[f], tags=tags.astype(int), cell=cell.reshape(3, 3),
                             constraint=FixAtoms(mask=fixed.astype(bool)), pbc=[True, True, True]))
    tmp.rename(traj_path)
    return traj_path
