"""Device-side engine of the EquiformerV2 denoiser: owns the ``adf_eqv2`` handle, hands over the constant SO(3) tables
(so3_math.py) and the module's parameters, and enqueues forward / stepper calls on torch's current HIP stream.  Same
interface as ``PaiNNEngine`` towards ``Denoiser`` (denoising_torch.py).  PyTorch is plumbing here; nothing in this file
computes a model output on the host.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np
import torch

from . import lib as _lib
from . import so3_math
from .engine import PreparedBatch, _require_gpu, batch_pbc, cell_repeats


def attn_weight_names(prefix: str, mmax: int) -> List[str]:
    rad = [f"net.{i}.{k}" for i in (0, 1, 3, 4, 6) for k in ("weight", "bias")]
    names = [prefix + "alpha_dot", prefix + "source_embedding.weight", prefix + "target_embedding.weight",
             prefix + "so2_conv_1.fc_m0.weight", prefix + "so2_conv_1.fc_m0.bias"]
    names += [prefix + f"so2_conv_1.so2_m_conv.{m}.fc.weight" for m in range(mmax)]
    names += [prefix + "so2_conv_1.rad_func." + r for r in rad]
    names += [prefix + "alpha_norm.weight", prefix + "alpha_norm.bias", prefix + "so2_conv_2.fc_m0.weight",
              prefix + "so2_conv_2.fc_m0.bias"]
    names += [prefix + f"so2_conv_2.so2_m_conv.{m}.fc.weight" for m in range(mmax)]
    names += [prefix + "proj.weight", prefix + "proj.bias"]
    return names


def weight_names(num_layers: int, mmax: int) -> List[str]:
    """The table order of ``adf_eqv2_set_weights`` (include/adsorbdiff_hip.h) in reference state_dict names."""
    rad = [f"net.{i}.{k}" for i in (0, 1, 3, 4, 6) for k in ("weight", "bias")]
    norm = ["affine_weight", "norm_l0.weight", "norm_l0.bias"]
    names = ["atom_radii", "sphere_embedding.weight", "edge_degree_embedding.source_embedding.weight",
             "edge_degree_embedding.target_embedding.weight"]
    names += ["edge_degree_embedding.rad_func." + r for r in rad]
    for i in range(num_layers):
        p = f"blocks.{i}."
        names += [p + "norm_1." + n for n in norm]
        names += attn_weight_names(p + "ga.", mmax)
        names += [p + "norm_2." + n for n in norm]
        names += [p + "ffn." + n for n in ("so3_linear_1.weight", "so3_linear_1.bias", "scalar_mlp.0.weight",
                                            "scalar_mlp.0.bias", "grid_mlp.0.weight", "grid_mlp.2.weight",
                                            "grid_mlp.4.weight", "so3_linear_2.weight", "so3_linear_2.bias")]
    names += ["norm." + n for n in norm]
    names += attn_weight_names("force_block.", mmax)
    names += attn_weight_names("force_block2.", mmax)
    return names


class EqV2Engine:
    PROFILE_CATEGORIES = ("graph", "radial", "rotate", "so2_conv", "s2_act", "attn_weights", "node", "ffn_grid", "stepper")

    def __init__(self, model, device) -> None:
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"EqV2Engine needs a ROCm device, got {self.device} (no CPU fallback)")
        self.model = model
        self.lmax, self.mmax = int(model.lmax_list[0]), int(model.mmax_list[0])
        hp = _lib.EqV2Hparams(
            lmax=self.lmax, mmax=self.mmax, num_layers=model.num_layers, sphere_channels=model.sphere_channels,
            attn_hidden_channels=model.attn_hidden_channels, num_heads=model.num_heads,
            attn_alpha_channels=model.attn_alpha_channels, attn_value_channels=model.attn_value_channels,
            ffn_hidden_channels=model.ffn_hidden_channels, grid_resolution=int(model.grid_resolution),
            edge_channels=model.edge_channels, num_distance_basis=model.NUM_GAUSSIANS,
            max_num_elements=model.max_num_elements, max_neighbors=model.max_neighbors,
            max_radius=float(model.max_radius), avg_degree=float(model.avg_degree),
        )
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_eqv2_create(C.byref(hp), C.byref(self.handle)))
            t = so3_math.device_tables(self.lmax, self.mmax, int(model.grid_resolution))
            ptr = lambda a: a.ctypes.data_as(C.c_void_p)
            _lib.check(self.lib.adf_eqv2_set_constants(self.handle, ptr(t["jd"]), ptr(t["to_red"]), ptr(t["from_red"]),
                                                       ptr(t["to_full"]), ptr(t["from_full"])))
        self._weights_keepalive: List[torch.Tensor] = []
        self._moving_keepalive = None
        self._edges_keepalive = None
        import os

        self.exact_f32 = os.environ.get("ADF_GEMM") == "f32"
        self.bind_weights()

    # ------------------------------------------------------------------ weights
    def bind_weights(self) -> None:
        sd = dict(self.model.named_parameters())
        out = []
        for n in weight_names(self.model.num_layers, self.mmax):
            t = sd[n].detach()
            _require_gpu(t, f"parameter {n}")
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.to(torch.float32).contiguous()
            out.append(t)
        self._weights_keepalive = out
        ptrs = (C.c_void_p * len(out))(*[w.data_ptr() for w in out])
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_eqv2_set_weights(self.handle, len(out), ptrs, self._stream()))

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
            batch=data.batch.to(dev, torch.int32).contiguous(), atom_offset=off, reps=reps,
        )
        if hasattr(data, "tags") and data.tags is not None:
            prep.tags = data.tags.to(dev, torch.int32).contiguous()
        if hasattr(data, "fixed") and data.fixed is not None:
            prep.fixed = data.fixed.to(dev, torch.int32).contiguous()
        return prep

    def set_moving_atoms(self, prep: Optional[PreparedBatch], moving_mask: Optional[torch.Tensor]) -> None:
        if moving_mask is None or prep is None:
            self._moving_keepalive = None
            _lib.check(self.lib.adf_eqv2_set_moving(self.handle, None, None, None))
            return
        mask = moving_mask.to(self.device, torch.int32).contiguous()
        idx = torch.nonzero(mask).reshape(-1).to(torch.int32).contiguous()
        per_sys = torch.bincount(prep.batch[idx.long()].long(), minlength=prep.num_systems)
        off = torch.zeros(prep.num_systems + 1, dtype=torch.int32, device=self.device)
        off[1:] = torch.cumsum(per_sys, 0).to(torch.int32)
        self._moving_keepalive = (mask, idx, off)
        _lib.check(self.lib.adf_eqv2_set_moving(self.handle, mask.data_ptr(), idx.data_ptr(), off.data_ptr()))

    def set_edges(self, edge_index: Optional[torch.Tensor], edge_vec: Optional[torch.Tensor]) -> None:
        """Run the next forwards on this edge list ([2,E] (source, target) sorted by target, vectors [E,3]) instead of
        building one; ``None`` switches back.  For parity runs against reference outputs whose choice among exactly tied
        K-th neighbours is implementation-defined."""
        if edge_index is None:
            _lib.check(self.lib.adf_eqv2_set_edges(self.handle, 0, None, None, None, 1, self._stream()))
            return
        dst = edge_index[1].to(self.device, torch.int32).contiguous()
        src = edge_index[0].to(self.device, torch.int32).contiguous()
        if dst.numel() > 1 and bool((dst[1:] < dst[:-1]).any().item()):
            raise ValueError("set_edges: edges must be sorted by target")
        vec = edge_vec.to(self.device, torch.float32).contiguous()
        maxdeg = int(torch.bincount(dst.long()).max().item())
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_eqv2_set_edges(self.handle, int(dst.numel()), src.data_ptr(), dst.data_ptr(),
                                                   vec.data_ptr(), maxdeg, self._stream()))

    # ------------------------------------------------------------------ calls
    def forward_prepared(self, prep: PreparedBatch, pos: torch.Tensor, f1: torch.Tensor, f2: Optional[torch.Tensor],
                         out_idx=None, x_blocks: Optional[torch.Tensor] = None) -> None:
        """Enqueue one forward; no host synchronisation.  ``out_idx`` (ascending int32 atom indices on the device): only
        those rows of f1 / f2 are evaluated - bit-identical to the full forward's - and written
        (``adf_eqv2_forward_subset``)."""
        desc = prep.desc(pos)
        with torch.cuda.device(self.device):
            if out_idx is not None:
                if x_blocks is not None:
                    raise ValueError("x_blocks are recorded by the full forward only")
                assert out_idx.dtype == torch.int32 and out_idx.is_contiguous() and out_idx.device == pos.device
                _lib.check(self.lib.adf_eqv2_forward_subset(
                    self.handle, C.byref(desc), out_idx.data_ptr(), int(out_idx.numel()), f1.data_ptr(),
                    f2.data_ptr() if f2 is not None else None, self._stream()))
                return
            _lib.check(self.lib.adf_eqv2_forward(
                self.handle, C.byref(desc), f1.data_ptr(), f2.data_ptr() if f2 is not None else None,
                x_blocks.data_ptr() if x_blocks is not None else None, self._stream()))

    def check_flags(self) -> None:
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_eqv2_check_flags(self.handle, self._stream()))

    def forward(self, data, return_blocks: bool = False):
        prep = self.prepare(data)
        pos = data.pos.to(torch.float32).contiguous()
        f1 = torch.empty(prep.num_atoms, 3, dtype=torch.float32, device=self.device)
        f2 = torch.empty_like(f1)
        xb = None
        if return_blocks:
            S = (self.lmax + 1) ** 2
            xb = torch.empty(self.model.num_layers + 1, prep.num_atoms, S, self.model.sphere_channels,
                             dtype=torch.float32, device=self.device)
        self.forward_prepared(prep, pos, f1, f2, x_blocks=xb)
        self.check_flags()
        return (f1, f2, xb) if return_blocks else (f1, f2)

    def use_exact_f32(self) -> bool:
        if self.exact_f32:
            return False
        _lib.check(self.lib.adf_eqv2_set_arithmetic(self.handle, 1))
        self.exact_f32 = True
        return True

    def set_arithmetic(self, exact_f32: bool) -> None:
        _lib.check(self.lib.adf_eqv2_set_arithmetic(self.handle, 1 if exact_f32 else 0))
        self.exact_f32 = bool(exact_f32)

    def set_incremental(self, on: bool = True) -> None:
        """Incremental blocks (adf_eqv2_set_incremental): keep every block's output across the forwards of a
        static-atom run and recompute only the rows whose inputs changed (bit-identical results)."""
        _lib.check(self.lib.adf_eqv2_set_incremental(self.handle, 1 if on else 0))

    def init_placement(self, prep: PreparedBatch, pos: torch.Tensor, noise: torch.Tensor) -> None:
        desc = prep.desc(pos)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_eqv2_init_placement(
                self.handle, C.byref(desc), pos.data_ptr(), prep.tags.data_ptr(), noise.data_ptr(), self._stream()))

    def sde_step(self, prep: PreparedBatch, pos, f1, f2, coef: _lib.StepCoef, state, z_tr=None, z_rot=None,
                 early_stop_count: int = 10, dcom=None, drot=None) -> None:
        desc = prep.desc(pos)
        opt = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_eqv2_sde_step(
                self.handle, C.byref(desc), pos.data_ptr(), prep.tags.data_ptr(), opt(prep.fixed), f1.data_ptr(),
                f2.data_ptr(), C.byref(coef), None, 0, opt(z_tr), opt(z_rot), early_stop_count, state.data_ptr(),
                opt(dcom), opt(drot), self._stream()))

    def sde_step_scheduled(self, prep: PreparedBatch, pos, f1, f2, coefs_dev: torch.Tensor, num_steps: int, state,
                           z_tr=None, z_rot=None, early_stop_count: int = 10) -> None:
        desc = prep.desc(pos)
        opt = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_eqv2_sde_step(
                self.handle, C.byref(desc), pos.data_ptr(), prep.tags.data_ptr(), opt(prep.fixed), f1.data_ptr(),
                f2.data_ptr(), None, coefs_dev.data_ptr(), num_steps, opt(z_tr), opt(z_rot), early_stop_count,
                state.data_ptr(), None, None, self._stream()))

    def sample(self, prep: PreparedBatch, pos, f1, f2, coefs_dev: torch.Tensor, num_steps: int, state,
               z_tr_all=None, z_rot_all=None, early_stop_count: int = 10, poll_every: int = 0, out_idx=None, sink=None,
               frame_every: int = 1) -> None:
        desc = prep.desc(pos)
        opt = lambda t: t.data_ptr() if t is not None else None
        args = [self.handle, C.byref(desc), pos.data_ptr(), prep.tags.data_ptr(), opt(prep.fixed), coefs_dev.data_ptr(),
                num_steps, opt(z_tr_all), opt(z_rot_all), early_stop_count, poll_every, state.data_ptr(), opt(out_idx),
                int(out_idx.numel()) if out_idx is not None else 0, f1.data_ptr(), f2.data_ptr()]
        with torch.cuda.device(self.device):
            if sink is None:
                _lib.check(self.lib.adf_eqv2_sample(*args, self._stream()))
            else:
                _lib.check(self.lib.adf_eqv2_sample_traj(*args, sink.handle, int(frame_every), self._stream()))

    def counters(self) -> _lib.EqV2Counters:
        c = _lib.EqV2Counters()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_eqv2_get_counters(self.handle, C.byref(c), self._stream()))
        return c

    def profile_enable(self, on: bool = True) -> None:
        _lib.check(self.lib.adf_eqv2_profile_enable(self.handle, 1 if on else 0))

    def profile_read(self):
        n = len(self.PROFILE_CATEGORIES)
        ms = (C.c_float * n)()
        cnt = (C.c_int64 * n)()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.adf_eqv2_profile_read(self.handle, ms, cnt, self._stream()))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(self.PROFILE_CATEGORIES)}

    def close(self) -> None:
        if getattr(self, "handle", None) is not None and self.handle:
            with torch.cuda.device(self.device):
                torch.cuda.synchronize(self.device)
                self.lib.adf_eqv2_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
