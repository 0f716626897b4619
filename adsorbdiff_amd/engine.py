"""Device-side engine of the PaiNN denoiser: owns the C-ABI handle, binds the module's
parameters, converts a (PyG-like) batch into the ``adf_batch`` descriptor and enqueues
forward / stepper calls on torch's current HIP stream.

PyTorch is plumbing here (device memory, streams); all arithmetic is in
libadsorbdiff_hip.so.  Nothing in this file computes a model output on the host.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import lib as _lib


def cell_repeats(cell: torch.Tensor, radius: float, pbc: Sequence[bool] = (True, True, True)) -> List[int]:
    """Periodic images needed per lattice direction, max over the batch — host logic of
    the reference's radius_graph_pbc (adsorbdiff/utils/utils.py:634-662), evaluated with the same
    float32 torch ops on a CPU copy of ``cell``."""
    cell = cell.detach().to("cpu", torch.float32).reshape(-1, 3, 3)
    a1, a2, a3 = cell[:, 0], cell[:, 1], cell[:, 2]
    c23 = torch.cross(a2, a3, dim=-1)
    vol = torch.sum(a1 * c23, dim=-1, keepdim=True)
    crosses = (c23, torch.cross(a3, a1, dim=-1), torch.cross(a1, a2, dim=-1))
    reps = []
    for k in range(3):
        if pbc[k]:
            inv_min_dist = torch.norm(crosses[k] / vol, p=2, dim=-1)
            reps.append(int(torch.ceil(radius * inv_min_dist).max().item()))
        else:
            reps.append(0)
    return reps


def batch_pbc(data) -> List[bool]:
    """Reference: utils/utils.py:566-576 (default all-periodic; mixed PBC in one batch is an error)."""
    pbc = [True, True, True]
    if hasattr(data, "pbc") and getattr(data, "pbc") is not None:
        flags = torch.atleast_2d(data.pbc).to("cpu")
        for i in range(3):
            if not torch.any(flags[:, i]).item():
                pbc[i] = False
            elif torch.all(flags[:, i]).item():
                pbc[i] = True
            else:
                raise RuntimeError(
                    "Different structures in the batch have different PBC configurations. "
                    "This is not currently supported."
                )
    return pbc


@dataclass
class PreparedBatch:
    """Step-invariant device arrays of one batch (everything but positions)."""

    num_systems: int
    num_atoms: int
    cell: torch.Tensor          # [B,3,3] f32
    atomic_numbers: torch.Tensor  # [N] i32
    batch: torch.Tensor         # [N] i32
    atom_offset: torch.Tensor   # [B+1] i32
    reps: List[int]
    tags: Optional[torch.Tensor] = None   # [N] i32
    fixed: Optional[torch.Tensor] = None  # [N] i32

    def desc(self, pos: torch.Tensor) -> _lib.BatchDesc:
        d = _lib.BatchDesc()
        d.num_systems, d.num_atoms = self.num_systems, self.num_atoms
        d.pos = pos.data_ptr()
        d.cell = self.cell.data_ptr()
        d.atomic_numbers = self.atomic_numbers.data_ptr()
        d.batch = self.batch.data_ptr()
        d.atom_offset = self.atom_offset.data_ptr()
        d.reps[0], d.reps[1], d.reps[2] = self.reps
        return d


def _require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} is on {t.device}: the adsorbdiff_amd HIP path only runs on a ROCm device "
            "(there is no CPU fallback)"
        )


class PaiNNEngine:
    def __init__(self, model, device) -> None:
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"PaiNNEngine needs a ROCm device, got {self.device} (no CPU fallback)")
        self.model = model
        hp = _lib.Hparams(
            hidden_channels=model.hidden_channels, num_layers=model.num_layers, num_rbf=model.num_rbf,
            num_elements=model.num_elements, max_neighbors=model.max_neighbors,
            envelope_exponent=int(model.radial_basis.envelope.p), num_heads=2 if model.so3_denoising else 1,
            cutoff=float(model.cutoff),
        )
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_painn_create(C.byref(hp), C.byref(self.handle)))
        self._weights_keepalive: List[torch.Tensor] = []
        import os

        self.exact_f32 = os.environ.get("ADF_GEMM") == "f32"
        self._last_graph_N = 0
        self.bind_weights()

    # ------------------------------------------------------------------ weights
    def _weight_list(self) -> List[torch.Tensor]:
        m = self.model
        sd = dict(m.named_parameters())
        sd.update(dict(m.named_buffers()))
        names = ["atom_emb.embeddings.weight", "radial_basis.rbf.offset"]
        for i in range(m.num_layers):
            p, u = f"message_layers.{i}.", f"update_layers.{i}."
            names += [
                p + "x_layernorm.weight", p + "x_layernorm.bias", p + "x_proj.0.weight", p + "x_proj.0.bias",
                p + "x_proj.2.weight", p + "x_proj.2.bias", p + "rbf_proj.weight", p + "rbf_proj.bias",
                u + "vec_proj.weight", u + "xvec_proj.0.weight", u + "xvec_proj.0.bias",
                u + "xvec_proj.2.weight", u + "xvec_proj.2.bias",
            ]
        heads = ["out_forces"] + (["out_forces2"] if m.so3_denoising else [])
        for hname in heads:
            for b in range(2):
                q = f"{hname}.output_network.{b}."
                names += [q + "vec1_proj.weight", q + "vec2_proj.weight", q + "update_net.0.weight",
                          q + "update_net.0.bias", q + "update_net.2.weight", q + "update_net.2.bias"]
        out = []
        for n in names:
            t = sd[n].detach()
            _require_gpu(t, f"parameter {n}")
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.to(torch.float32).contiguous()
            out.append(t)
        return out

    def bind_weights(self) -> None:
        ws = self._weight_list()
        self._weights_keepalive = ws
        ptrs = (C.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
        scales = self.model.scale_factors()
        sf = (C.c_float * len(scales))(*scales)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_painn_set_weights(self.handle, len(ws), ptrs, sf, self._stream()))

    # ------------------------------------------------------------------ batches
    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def prepare(self, data) -> PreparedBatch:
        _require_gpu(data.pos, "data.pos")
        dev = self.device
        natoms = data.natoms.to(dev, torch.int64).reshape(-1)
        B = int(natoms.shape[0])
        N = int(data.pos.shape[0])
        off = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        off[1:] = torch.cumsum(natoms, 0).to(torch.int32)
        cell = data.cell.to(dev, torch.float32).reshape(B, 3, 3).contiguous()
        reps = cell_repeats(cell, float(self.model.cutoff), batch_pbc(data))
        prep = PreparedBatch(
            num_systems=B, num_atoms=N, cell=cell,
            atomic_numbers=data.atomic_numbers.to(dev).long().to(torch.int32).contiguous(),
            batch=data.batch.to(dev, torch.int32).contiguous(),
            atom_offset=off, reps=reps,
        )
        if hasattr(data, "tags") and data.tags is not None:
            prep.tags = data.tags.to(dev, torch.int32).contiguous()
        if hasattr(data, "fixed") and data.fixed is not None:
            prep.fixed = data.fixed.to(dev, torch.int32).contiguous()
        return prep

    def set_moving_atoms(self, prep: Optional[PreparedBatch], moving_mask: Optional[torch.Tensor]) -> None:
        """Declare which atoms may move between the next graph builds of ``prep`` (None switches the
        static-atom cache off).  The arrays are kept alive on the engine until the next call."""
        if moving_mask is None or prep is None:
            self._moving_keepalive = None
            _lib.check(self.lib.adf_graph_set_moving(self.handle, None, None, None))
            return
        mask = moving_mask.to(self.device, torch.int32).contiguous()
        idx = torch.nonzero(mask).reshape(-1).to(torch.int32).contiguous()
        per_sys = torch.bincount(prep.batch[idx.long()].long(), minlength=prep.num_systems)
        off = torch.zeros(prep.num_systems + 1, dtype=torch.int32, device=self.device)
        off[1:] = torch.cumsum(per_sys, 0).to(torch.int32)
        self._moving_keepalive = (mask, idx, off)
        _lib.check(self.lib.adf_graph_set_moving(self.handle, mask.data_ptr(), idx.data_ptr(), off.data_ptr()))

    # ------------------------------------------------------------------ calls
    def forward_prepared(self, prep: PreparedBatch, pos: torch.Tensor, f1: torch.Tensor, f2: Optional[torch.Tensor],
                         out_idx: Optional[torch.Tensor] = None) -> None:
        """Enqueue one forward; no host synchronisation.  ``out_idx`` (ascending int32 atom indices on the
        device): evaluate the outputs of these atoms only — their rows of f1 / f2 are bit-identical to the full
        forward's, the other rows are left untouched (``adf_painn_forward_subset``)."""
        desc = prep.desc(pos)
        with torch.cuda.device(self.device):
            if out_idx is None:
                _lib.check(self.lib.adf_painn_forward(
                    self.handle, C.byref(desc), f1.data_ptr(), f2.data_ptr() if f2 is not None else None, self._stream()))
            else:
                assert out_idx.dtype == torch.int32 and out_idx.is_contiguous() and out_idx.device == pos.device
                _lib.check(self.lib.adf_painn_forward_subset(
                    self.handle, C.byref(desc), out_idx.data_ptr(), int(out_idx.numel()), f1.data_ptr(),
                    f2.data_ptr() if f2 is not None else None, self._stream()))

    def check_flags(self) -> None:
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_check_flags(self.handle, self._stream()))

    def forward(self, data):
        prep = self.prepare(data)
        pos = data.pos.to(torch.float32).contiguous()
        f1 = torch.empty(prep.num_atoms, 3, dtype=torch.float32, device=self.device)
        f2 = torch.empty_like(f1) if self.model.so3_denoising else None
        self.forward_prepared(prep, pos, f1, f2)
        try:
            self.check_flags()  # ValueError on an image without neighbours, like the reference
        except _lib.NumericRangeError:
            if not self.use_exact_f32():  # already exact: the inputs / weights themselves are not finite
                raise
            self.forward_prepared(prep, pos, f1, f2)
            self.check_flags()
        return f1, f2

    def use_exact_f32(self) -> bool:
        """Switch the handle to exact-f32 arithmetic (after a NumericRangeError in the default f16x3 mode: an
        activation left the fp16 range).  Returns False if it already was exact."""
        if self.exact_f32:
            return False
        import logging

        logging.warning("adsorbdiff_amd: non-finite output in f16x3 arithmetic; re-running in exact f32 "
                        "(this engine stays in exact f32)")
        _lib.check(self.lib.adf_painn_set_arithmetic(self.handle, 1))
        self.exact_f32 = True
        return True

    def set_incremental(self, on: bool = True) -> None:
        """Incremental layers (adf_painn_set_incremental): keep per-layer node state across the forwards of a
        static-atom promise and recompute only rows whose inputs changed.  Bit-identical outputs; default on."""
        _lib.check(self.lib.adf_painn_set_incremental(self.handle, 1 if on else 0))

    def set_fused_mlp(self, mode: int = 2) -> None:
        """Form of the x_proj / xvec_proj pairs (adf_painn_set_fused_mlp): 0 two kernels per pair, 1 the fused two-layer
        kernel (csrc/mlp16.hip), 2 by size (default).  Bit-identical results."""
        _lib.check(self.lib.adf_painn_set_fused_mlp(self.handle, int(mode)))

    def build_graph(self, data, prep=None):
        """Graph only; returns the number of symmetrised edges.  ``prep``: an already prepared batch of ``data``."""
        if prep is None:
            prep = self.prepare(data)
        pos = data.pos.to(torch.float32).contiguous()
        desc = prep.desc(pos)
        n = C.c_int64(0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_graph_build(self.handle, C.byref(desc), self._stream(), C.byref(n)))
        self._last_graph_N = prep.num_atoms
        return int(n.value)

    def export_graph(self):
        """(nbr_count[N], nbr_src[N,K], nbr_shift[N,K,3], edge_src[E], edge_dst[E], dist[E], vec[E,3])."""
        N, K = self._last_graph_N, self.model.max_neighbors
        if N <= 0:
            raise RuntimeError("export_graph: call build_graph first")
        dev = self.device
        cnt = torch.empty(N, dtype=torch.int32, device=dev)
        src = torch.empty(N, K, dtype=torch.int32, device=dev)
        sh = torch.empty(N, K, 3, dtype=torch.int32, device=dev)
        cap = 2 * N * K
        es = torch.empty(cap, dtype=torch.int32, device=dev)
        ed = torch.empty(cap, dtype=torch.int32, device=dev)
        dist = torch.empty(cap, dtype=torch.float32, device=dev)
        vec = torch.empty(cap, 3, dtype=torch.float32, device=dev)
        n = C.c_int64(0)
        with torch.cuda.device(dev):
            _lib.check(self.lib.adf_graph_export(
                self.handle, cnt.data_ptr(), src.data_ptr(), sh.data_ptr(), cap, es.data_ptr(), ed.data_ptr(),
                dist.data_ptr(), vec.data_ptr(), C.byref(n), self._stream()))
        E = int(n.value)
        return cnt, src, sh, es[:E], ed[:E], dist[:E], vec[:E]

    def message_layer(self, layer: int, x: torch.Tensor, vec: torch.Tensor):
        x_out, vec_out = torch.empty_like(x), torch.empty_like(vec)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_painn_message_layer(
                self.handle, layer, x.shape[0], x.data_ptr(), vec.data_ptr(), x_out.data_ptr(), vec_out.data_ptr(),
                self._stream()))
        return x_out, vec_out

    def update_layer(self, layer: int, x: torch.Tensor, vec: torch.Tensor):
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_painn_update_layer(
                self.handle, layer, x.shape[0], x.data_ptr(), vec.data_ptr(), self._stream()))
        return x, vec

    def init_placement(self, prep: PreparedBatch, pos: torch.Tensor, noise: torch.Tensor) -> None:
        desc = prep.desc(pos)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_sde_init_placement(
                self.handle, C.byref(desc), pos.data_ptr(), prep.tags.data_ptr(), noise.data_ptr(), self._stream()))

    def sde_step(self, prep: PreparedBatch, pos, f1, f2, coef: _lib.StepCoef, state, z_tr=None, z_rot=None,
                 early_stop_count: int = 10, dcom=None, drot=None) -> None:
        desc = prep.desc(pos)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_sde_step(
                self.handle, C.byref(desc), pos.data_ptr(), prep.tags.data_ptr(),
                prep.fixed.data_ptr() if prep.fixed is not None else None, f1.data_ptr(), f2.data_ptr(),
                C.byref(coef), z_tr.data_ptr() if z_tr is not None else None,
                z_rot.data_ptr() if z_rot is not None else None, early_stop_count, state.data_ptr(),
                dcom.data_ptr() if dcom is not None else None, drot.data_ptr() if drot is not None else None,
                self._stream()))

    def sde_step_scheduled(self, prep: PreparedBatch, pos, f1, f2, coefs_dev: torch.Tensor, num_steps: int, state,
                           z_tr=None, z_rot=None, early_stop_count: int = 10) -> None:
        """Step whose schedule scalars come from a device table indexed by state[4] (graph-capturable)."""
        desc = prep.desc(pos)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_sde_step_scheduled(
                self.handle, C.byref(desc), pos.data_ptr(), prep.tags.data_ptr(),
                prep.fixed.data_ptr() if prep.fixed is not None else None, f1.data_ptr(), f2.data_ptr(),
                coefs_dev.data_ptr(), num_steps, z_tr.data_ptr() if z_tr is not None else None,
                z_rot.data_ptr() if z_rot is not None else None, early_stop_count, state.data_ptr(), None, None,
                self._stream()))

    def sample(self, prep: PreparedBatch, pos, f1, f2, coefs_dev: torch.Tensor, num_steps: int, state,
               z_tr_all=None, z_rot_all=None, early_stop_count: int = 10, poll_every: int = 0,
               out_idx: Optional[torch.Tensor] = None, sink=None, frame_every: int = 1) -> None:
        """The whole reverse loop in one library call (``adf_sample``; with ``sink`` — a ``trajectory.FrameSink`` —
        ``adf_sample_traj``: a frame of the positions leaves the device after every ``frame_every``-th step)."""
        desc = prep.desc(pos)
        args = [self.handle, C.byref(desc), pos.data_ptr(), prep.tags.data_ptr(),
                prep.fixed.data_ptr() if prep.fixed is not None else None, coefs_dev.data_ptr(), num_steps,
                z_tr_all.data_ptr() if z_tr_all is not None else None,
                z_rot_all.data_ptr() if z_rot_all is not None else None, early_stop_count, poll_every,
                state.data_ptr(), out_idx.data_ptr() if out_idx is not None else None,
                int(out_idx.numel()) if out_idx is not None else 0, f1.data_ptr(), f2.data_ptr()]
        with torch.cuda.device(self.device):
            if sink is None:
                _lib.check(self.lib.adf_sample(*args, self._stream()))
            else:
                _lib.check(self.lib.adf_sample_traj(*args, sink.handle, int(frame_every), self._stream()))

    def counters(self) -> _lib.Counters:
        c = _lib.Counters()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_get_counters(self.handle, C.byref(c), self._stream()))
        return c

    PROFILE_CATEGORIES = ("graph", "message", "node_dense", "heads", "stepper")

    def profile_enable(self, on: bool = True) -> None:
        _lib.check(self.lib.adf_profile_enable(self.handle, 1 if on else 0))

    def profile_read(self):
        """{category: (total_ms, groups)} since the last read (HIP events on the launch stream)."""
        ms = (C.c_float * 5)()
        cnt = (C.c_int64 * 5)()
        ksteps = C.c_int64(0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_profile_read(self.handle, ms, cnt, C.byref(ksteps), self._stream()))
        out = {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(self.PROFILE_CATEGORIES)}
        out["message_ksteps"] = int(ksteps.value)
        return out

    def measure_peaks(self):
        """On-box peaks for the roofline fractions: HBM stream copy (GB/s), f16 and f32 MFMA (TFLOP/s)."""
        out = (C.c_float * 3)()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_measure_peaks(out, self._stream()))
        return {"hbm_copy_gbps": float(out[0]), "mfma_f16_tflops": float(out[1]), "mfma_f32_tflops": float(out[2])}

    def close(self) -> None:
        if getattr(self, "handle", None) is not None and self.handle:
            with torch.cuda.device(self.device):
                torch.cuda.synchronize(self.device)
                self.lib.adf_painn_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
