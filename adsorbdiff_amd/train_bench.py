"""Throughput of the score-matching training step (BASELINE.json config 5 shape: PaiNN H=512, 6 layers, OC20-shaped
~200-atom graphs): `python -m adsorbdiff_amd.train_bench [--systems 64] [--steps 5]` prints one JSON line with
graphs/s of noising + forward + loss + backward + clip + AdamW + EMA on one GPU (secondary measurement; bench.py's
headline is the sampling loop).  Multi-GPU: launch with torch.distributed.run, one process per GPU; the per-rank batch
is fixed (weak scaling, as DDP training is) and gradients go through the bucketed all-reduce."""
from __future__ import annotations

import argparse
import json
import os
import time

import torch


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--systems", type=int, default=64, help="graphs per rank and step")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    import torch.distributed as dist

    from .painn_denoising import PaiNN
    from .scaling import PAINN_NB6_SCALE_FACTORS
    from .so3_tables import Igso3Tables
    from .synthetic import make_batch
    from .trainer import DenoisingTrainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(0)
    model = PaiNN(None, 50, 1, hidden_channels=512, num_layers=6, num_rbf=128, cutoff=10.0, max_neighbors=50,
                  scale_file=PAINN_NB6_SCALE_FACTORS, so3_denoising=True)
    tr = DenoisingTrainer(model, device=dev)
    params = dict(ads_std_low=0.1, ads_std_high=10, rot_std_low=0.01, rot_std_high=1.55)
    tr.setup_training(params, lr=1e-4, tables=Igso3Tables.shared())
    batch = make_batch(args.systems, seed=2000 + rank).to(dev)

    def step():
        b = batch.clone()
        return tr.train_step(b)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({
            "metric": "score-matching training step throughput (PaiNN H=512 x 6, ~200-atom graphs)",
            "value": args.systems * world * args.steps / dt, "unit": "graphs/s", "n_gpus": world,
            "graphs_per_rank_and_step": args.systems, "ms_per_step": dt / args.steps * 1e3,
            "loss": float(out["loss"][0]), "grad_norm": float(out["grad_norm"]), "dtype": "f32",
            "data": "synthetic", "scaling": "weak"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
