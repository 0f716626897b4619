"""Trainer-like object for sampling — the part of the reference's ``DenoisingTrainer`` that the
stepper touches (reference: adsorbdiff/trainers/sde_denoising_trainer.py:539-652, 750-813, and
denoising_torch.py:38,491-500): ``predict_denoising(batch, per_image=False)``,
``_unwrapped_model``, ``ema``, ``scaler``, ``device``, ``config["model_attributes"]`` and
``run_relaxations`` (the ``run-relaxations`` task entry, tasks/task.py:90-100).

``train_step`` is the per-batch body of the reference's training loop (sde_denoising_trainer.py:410-441 +
base_trainer.py:787-820): noising, forward, loss, backward, gradient all-reduce, clipping, AdamW, EMA — all device
arithmetic in HIP kernels (adsorbdiff_amd/train_step.py).  Datasets, LR schedules, logging, evaluation and checkpoint
*writing* stay out of scope; ``load_checkpoint`` reads the reference's checkpoint layout
(base_trainer.py:456-533) so trained weights can be sampled with.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Iterable, Optional

import torch

from .ml_relaxation import ml_diffuse
from .painn_denoising import PaiNN
from .scaling import ensure_fitted


def check_traj_files(batch, traj_dir) -> bool:
    """Resume rule of the sampler: a batch is skipped iff every <traj_dir>/<sid>.traj exists
    (reference: utils/utils.py:968-973); the ase-less sink <sid>.npz counts as well."""
    if traj_dir is None:
        return False
    traj_dir = Path(traj_dir)
    return all((traj_dir / f"{sid}.traj").exists() or (traj_dir / f"{sid}.npz").exists() for sid in batch.sid)


class DenoisingTrainer:
    def __init__(self, model: PaiNN, device="cuda:0", config: Optional[dict] = None, ema=None, relax_loader=None):
        self.device = torch.device(device)
        self.model = model.to(self.device)
        self.ema = ema
        self.scaler = None  # fp32 path; the reference's --amp autocast is not offered
        self.relax_loader = relax_loader
        self.config = config or {}
        self.config.setdefault("model_attributes", {})
        self.config["model_attributes"].setdefault("so3_denoising", bool(model.so3_denoising))
        self.config.setdefault("task", {})
        self.config.setdefault("optim", {})

    @property
    def _unwrapped_model(self):
        module = self.model
        while hasattr(module, "module"):  # DDP / OCPDataParallel wrappers
            module = module.module
        return module

    # ---------------------------------------------------------------- inference
    def _forward_denoising(self, batch):
        """Reference: sde_denoising_trainer.py:539-553."""
        if not self.config["model_attributes"].get("so3_denoising", False):
            return {"positions": self.model(batch.to(self.device))}
        out1, out2 = self.model(batch.to(self.device))
        return {"positions": out1, "positions_free": out2}

    @torch.no_grad()
    def predict_denoising(self, data_loader, per_image: bool = True, results_file=None, disable_tqdm: bool = False):
        """Only the ``per_image=False`` form the stepper uses (reference :555-652)."""
        if per_image:
            raise NotImplementedError("per_image=True (result-file writer) is outside the sampling path")
        ensure_fitted(self._unwrapped_model, warn=True)
        self.model.eval()
        if self.ema:
            self.ema.store()
            self.ema.copy_to()
        try:
            out = self._forward_denoising(data_loader)
            predictions = {"positions": out["positions"].detach()}
            if "positions_free" in out:
                predictions["positions_free"] = out["positions_free"].detach()
        finally:
            if self.ema:
                self.ema.restore()
        return predictions

    # ---------------------------------------------------------------- training
    def setup_training(self, denoising_pos_params: dict, lr: float = 1e-3, weight_decay: float = 0.001,
                       clip_grad_norm: float = 100.0, ema_decay: float = 0.999, tables=None) -> None:
        """Optimizer / EMA / noising parameters; defaults = configs/denoising/painn_so3.yml:56-83 (AdamW, weight decay
        1e-3 except no_weight_decay() names, clip 100, EMA 0.999)."""
        from .exponential_moving_average import ExponentialMovingAverage
        from .train_step import FusedAdamW, PaiNNTrainStep

        self.denoising_pos_params = dict(denoising_pos_params)
        self.train_engine = PaiNNTrainStep(self._unwrapped_model, self.device, igso3=tables)
        if ema_decay:
            self.ema = ExponentialMovingAverage(self._unwrapped_model.parameters(), ema_decay)
        self.optimizer = FusedAdamW(self._unwrapped_model, lr=lr, weight_decay=weight_decay, max_grad_norm=clip_grad_norm,
                                    ema=self.ema)
        self.step = 0

    def train_step(self, batch, noised: bool = False) -> dict:
        """One optimisation step on ``batch`` (clean positions unless ``noised``).  Multi-GPU: one process per GPU, each
        with its own batch; gradients are averaged with a bucketed all-reduce (RCCL over xGMI under backend nccl)."""
        import torch.distributed as dist

        from .noising import tr_so3_schedule
        from .train_step import GradientReducer

        self.model.train()
        batch = batch.to(self.device)
        if hasattr(batch, "pos_relaxed"):
            batch.pos = batch.pos_relaxed
        if not noised:
            batch = tr_so3_schedule(batch, self.denoising_pos_params, self.train_engine.igso3)
        targets = {k: getattr(batch, k) for k in ("tr_sigma", "rot_sigma", "tr_score", "rot_score")}
        self.train_engine.zero_grad()
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # the buckets' all-reduces are issued from inside the backward (heads, then layer by layer) and overlap with it
        reducer = GradientReducer(self._unwrapped_model, world)
        loss = self.train_engine.loss_and_grad(batch, targets, grads_ready=reducer.ready if world > 1 else None)
        # NaN policy of the reference loop (sde_denoising_trainer.py:428-440), mirrored: a step whose loss is NaN is
        # skipped (`continue`: warning, no update; the reference's `nan_count > 10` test sits behind the reset and can never
        # fire, so there is no such stop here either); a loss above 1e6 - an Inf loss included, isnan() is False for it -
        # logs a warning and STOPS the epoch loop (`break`): returned as "stop": True, never raised.  All ranks must take
        # the same branch, so ONE two-element flag (NaN, too high) is MAX-reduced over the ranks and read back once; the
        # fused optimizer additionally turns an update with a non-finite gradient norm into a no-op on the device
        # (csrc/train.hip: tr_adamw_kernel).
        l0 = loss.detach().reshape(-1)
        flag = torch.stack([torch.isnan(l0).any(), l0[0] > 1e6]).to(torch.int32)
        if world > 1:
            flag = flag if dist.get_backend() != "gloo" else flag.cpu()
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        is_nan, too_high = (bool(v) for v in flag.tolist())    # the step's single device-to-host read
        if is_nan:
            logging.warning("NaN loss detected, skipping step")
            self.nan_count = getattr(self, "nan_count", 0) + 1
            reducer.finish()   # every rank takes this branch: drain the buckets already in flight
            self.train_engine.zero_grad()
            return {"loss": loss, "grad_norm": None, "skipped": True, "stop": False}
        self.nan_count = 0
        if too_high:
            logging.warning("Loss too high: %s", float(l0[0]))
            reducer.finish()
            self.train_engine.zero_grad()
            return {"loss": loss, "grad_norm": None, "skipped": True, "stop": True}
        if world > 1 and loss.is_cuda:   # what the backward did not hide: time spent waiting for the last buckets
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reducer.finish()
            e1.record()
            self.allreduce_wait_events = getattr(self, "allreduce_wait_events", [])[-63:] + [(e0, e1)]
        else:
            reducer.finish()
        grad_norm = self.optimizer.step()
        self.step += 1
        return {"loss": loss, "grad_norm": grad_norm, "skipped": False, "stop": False}

    # ---------------------------------------------------------------- checkpoint ingest
    def load_checkpoint(self, checkpoint_path: str) -> None:
        """Reads ``state_dict`` (with 0-2 ``module.`` prefixes) from a reference checkpoint
        (base_trainer.py:456-533).  ``ema`` shadow parameters, if present, are applied directly."""
        ckpt = torch.load(checkpoint_path, map_location="cpu")
        sd = ckpt.get("state_dict", ckpt)
        clean = {}
        for k, v in sd.items():
            while k.startswith("module."):
                k = k[len("module."):]
            clean[k] = v
        missing, unexpected = self._unwrapped_model.load_state_dict(clean, strict=False)
        for k in missing:
            logging.warning(f"checkpoint is missing key {k}")
        for k in unexpected:
            logging.warning(f"checkpoint has unexpected key {k}")
        ema = ckpt.get("ema")
        if ema and "shadow_params" in ema:
            params = [p for p in self._unwrapped_model.parameters() if p.requires_grad]
            with torch.no_grad():
                for p, s in zip(params, ema["shadow_params"]):
                    p.copy_(s.to(p.device))

    # ---------------------------------------------------------------- sampling entry
    def run_relaxations(self, batches: Optional[Iterable] = None):
        """``--mode run-relaxations`` (reference :750-813): for every batch of the relax loader not
        already finished on disk, run the diffusion sampler.  Returns the list of sampled batches."""
        self.model.eval()
        task = self.config["task"]
        params = self.config["optim"].get("denoising_pos_params", {})
        traj_dir = task.get("relax_opt", {}).get("traj_dir", None)
        out = []
        for batch in (batches if batches is not None else self.relax_loader):
            if check_traj_files(batch, traj_dir):
                logging.info(f"Skipping batch: {batch.sid}")
                continue
            out.append(
                ml_diffuse(
                    batch=batch, model=self, denoising_pos_params=params, traj_dir=traj_dir,
                    save_full_traj=task.get("save_full_traj", True), device=str(self.device),
                    transform=None,
                )
            )
        return out
