"""``ml_diffuse`` — driver of the diffusion sampler.

Drop-in for ``adsorbdiff.relaxation.ml_relaxation.ml_diffuse`` (reference:
adsorbdiff/relaxation/ml_relaxation.py:98-168).  Written from its contract, not its text:

* same signature and return type (one re-collated ``Batch``);
* a ``RuntimeError`` while sampling a batch of more than one system (the HIP library reports device OOM
  and 32-bit-offset overflow as ``RuntimeError``) makes the batch be sampled as two halves instead;
* a ``RuntimeError`` on a single system propagates to the caller;
* order of the returned systems: the reference pushes both halves on the *left* of its work deque, first
  half first, so the second half is sampled (and collated) before the first one.  ``_sample_or_split``
  reproduces that order by recursing into the upper half first.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Iterator

import torch

from .data import Batch, data_list_collater
from .denoising_torch import Denoiser, DiffTorchCalc


def _sample_or_split(batch, make_denoiser) -> Iterator:
    """Yield sampled (sub-)batches of ``batch``; halve and retry on RuntimeError."""
    try:
        yield make_denoiser(batch).run()
        return
    except RuntimeError:
        systems = batch.to_data_list()
        if len(systems) == 1:
            raise
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    logging.info(f"Failed to relax batch with size: {len(systems)}, splitting into two...")
    half = len(systems) // 2
    for part in (systems[half:], systems[:half]):
        yield from _sample_or_split(data_list_collater(part), make_denoiser)


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
    sink = Path(traj_dir) if traj_dir is not None else None

    def make_denoiser(b):
        return Denoiser(b, DiffTorchCalc(model, transform), denoising_pos_params=denoising_pos_params,
                        device=device, save_full_traj=save_full_traj, traj_dir=sink, traj_names=b.sid,
                        early_stop_batch=early_stop_batch, logger=logger)

    return Batch.from_data_list(list(_sample_or_split(batch, make_denoiser)))
