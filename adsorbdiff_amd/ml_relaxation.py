"""``ml_diffuse`` — driver of the diffusion sampler.

Drop-in for ``adsorbdiff.relaxation.ml_relaxation.ml_diffuse`` (reference:
adsorbdiff/relaxation/ml_relaxation.py:98-168): a deque of batches; a ``RuntimeError`` from a
batch (the HIP library reports device OOM as RuntimeError) splits it into two halves that are
retried, a single-system failure is re-raised; the relaxed batches are re-collated at the end.
"""
from __future__ import annotations

import logging
from collections import deque
from pathlib import Path
from typing import Optional

import torch

from .data import Batch, data_list_collater
from .denoising_torch import Denoiser, DiffTorchCalc


def ml_diffuse(
    batch,
    model,
    denoising_pos_params: dict,
    traj_dir,
    save_full_traj,
    device: str = "cuda:0",
    transform=None,
    early_stop_batch: bool = False,
    logger=None,
):
    batches = deque([batch])
    relaxed_batches = []
    while batches:
        batch = batches.popleft()
        oom = False
        ids = batch.sid
        calc = DiffTorchCalc(model, transform)
        optimizer = Denoiser(
            batch,
            calc,
            device=device,
            save_full_traj=save_full_traj,
            traj_dir=Path(traj_dir) if traj_dir is not None else None,
            traj_names=ids,
            early_stop_batch=early_stop_batch,
            denoising_pos_params=denoising_pos_params,
            logger=logger,
        )
        e: Optional[RuntimeError] = None
        try:
            relaxed_batch = optimizer.run()
            relaxed_batches.append(relaxed_batch)
        except RuntimeError as err:
            e = err
            oom = True
            if torch.cuda.is_available():
                torch.cuda.empty_cache()
        if oom:
            data_list = batch.to_data_list()
            if len(data_list) == 1:
                assert isinstance(e, RuntimeError)
                raise e
            logging.info(f"Failed to relax batch with size: {len(data_list)}, splitting into two...")
            mid = len(data_list) // 2
            batches.appendleft(data_list_collater(data_list[:mid]))
            batches.appendleft(data_list_collater(data_list[mid:]))
    return Batch.from_data_list(relaxed_batches)
