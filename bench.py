#!/usr/bin/env python
"""Benchmark of the hot path: diffusion-sampled adsorbate sites per second.

One bench "step" = one complete pass of the sampler over the rank's batch of synthetic
OC20-Dense-shaped systems: initial placement, then `--num-steps` (50) reverse steps of
(periodic graph build -> PaiNN forward -> ODE update), i.e. BASELINE.json configs[1]
("PaiNN denoiser, 50-step sampling on 1000 synthetic OC20-Dense systems (~200 atoms, 10 A
cutoff), 1xMI355X").  Inputs are resident in HBM before the timed region starts.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: systems are independent, so each rank samples its own shard (no data-path
collective); the only exchange is one RCCL all_gather of the sampled adsorbate sites at the end
of every pass (inside the timed region).  Default `--scaling weak`: every rank gets `--systems`
systems; `--scaling strong` splits `--systems` over the ranks.

Rank 0 prints ONE JSON line (contract in the task statement) including
  "roofline":     dominant kernel (fused message kernel, f32 MFMA bound), timed live with HIP
                  events on the launch stream inside the library (adf_profile_*),
  "cpu_baseline": the CPU oracle (oracle/painn_oracle.py, a PyTorch restatement of the
                  reference path) timed on the host cores on a bounded sample (N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

METRIC = "sampled adsorbate sites/sec (50 denoise steps, ~200-atom slabs)"
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: Peak FP32 (matrix), dense
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: Peak BF16/FP16 MFMA, dense
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--systems", type=int, default=1000, help="systems per rank (weak) or in total (strong)")
    ap.add_argument("--num-steps", type=int, default=50, help="reverse-diffusion steps per sample")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for "
                                                      "testing the N>1 path with several ranks on one GPU)")
    ap.add_argument("--cpu-systems", type=int, default=4)
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args()


def cpu_baseline(model_sd, scale_factors, n_sys, n_steps, params):
    """Time the CPU oracle on `n_sys` systems x `n_steps` reverse steps of the same workload."""
    from adsorbdiff_amd.synthetic import make_batch
    from oracle import painn_oracle as O

    b = make_batch(n_sys, seed=1000)
    torch.manual_seed(0)
    noise = torch.rand(n_sys, 3)
    pos = O.initial_placement(b.pos.clone(), b.cell, b.tags, b.batch, noise)
    t0 = time.perf_counter()
    for t in range(n_steps):
        f1, f2 = O.painn_forward(model_sd, pos, b.atomic_numbers, b.cell, b.natoms, cutoff=10.0, max_neighbors=50,
                                 scale_factors=scale_factors)
        pos, _, _, _ = O.reverse_step(pos, b.cell, b.tags, b.batch, f1, f2, b.fixed, t, params)
    dt = time.perf_counter() - t0
    sys_steps_per_s = n_sys * n_steps / dt
    return {
        "value": sys_steps_per_s / params["num_steps"],
        "unit": "sites/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{n_sys} systems x {n_steps} of {params['num_steps']} reverse steps in {dt:.1f} s "
                  f"({sys_steps_per_s:.3f} system-steps/s), linearly extrapolated to {params['num_steps']} steps",
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"LOCAL_RANK={local_rank} but only {ndev} GPU(s) visible")
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from adsorbdiff_amd.denoising_torch import Denoiser, DiffTorchCalc
    from adsorbdiff_amd.painn_denoising import PaiNN
    from adsorbdiff_amd.sampler import gather_sites
    from adsorbdiff_amd.scaling import PAINN_NB6_SCALE_FACTORS
    from adsorbdiff_amd.synthetic import make_batch
    from adsorbdiff_amd.trainer import DenoisingTrainer

    # random-init weights of the reference architecture under seed 0 (+ shipped scale factors)
    torch.manual_seed(0)
    model = PaiNN(None, 50, 1, hidden_channels=512, num_layers=6, num_rbf=128, cutoff=10.0, max_neighbors=50,
                  scale_file=PAINN_NB6_SCALE_FACTORS, so3_denoising=True).eval()
    cpu_sd = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
    scale_factors = model.scale_factors()
    trainer = DenoisingTrainer(model, device=dev)

    if args.scaling == "weak":
        n_local = args.systems
        first = rank * args.systems
    else:
        base, rem = divmod(args.systems, world)
        n_local = base + (1 if rank < rem else 0)
        first = rank * base + min(rank, rem)
    total_systems = args.systems * world if args.scaling == "weak" else args.systems
    batch0 = make_batch(n_local, seed=1000 + rank, sid_offset=first).to(dev)
    params = dict(num_steps=args.num_steps, ads_std_low=0.1, ads_std_high=10, rot_std_low=0.01, rot_std_high=1.55,
                  ode=True, early_stop=False)
    eng = model.engine(dev)

    def one_pass(extra=None):
        b = batch0.clone()
        torch.manual_seed(0)
        den = Denoiser(b, DiffTorchCalc(trainer), dict(params, **(extra or {})), device=str(dev))
        out = den.run()
        assert den.steps_applied == args.num_steps, den.steps_applied
        return gather_sites(out, world)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        one_pass()
    eng.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sites = one_pass()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = eng.profile_read()
    eng.profile_enable(False)
    counters = eng.counters()

    # Secondary measurement, N = 1 only, never `value`: the same pass with the model outputs evaluated on the
    # adsorbate atoms only (the stepper reads nothing else; sampled positions are bit-identical, checked here).
    ads_only = None
    if world == 1:
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        sites_ads = one_pass({"scores_on_adsorbate_only": True})
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t1
        ads_only = {"value": total_systems / dt, "unit": "sites/s", "ms_per_step": dt * 1e3,
                    "identical_sites": bool(torch.equal(sites_ads, sites)),
                    "note": "opt-in denoising_pos_params['scores_on_adsorbate_only']: last layer's message targets, its "
                            "update and the heads evaluated for tag-2 atoms only (adf_painn_forward_subset); one pass, "
                            "not part of `value`"}

    if rank == 0:
        assert sites.shape[0] == total_systems, (sites.shape, total_systems)
        assert bool(torch.isfinite(sites).all())
        H, R = 512, 128
        E = counters.num_edges
        N_atoms = counters.num_atoms
        msg_ms, msg_launches = prof["message"]
        f16 = os.environ.get("ADF_MSG", os.environ.get("ADF_GEMM", "f16")) != "f32"
        products = 3 if f16 else 1  # f16x3 split issues three MFMA products per contraction step
        peak = PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS
        dense_flops_per_launch = 2.0 * R * 3 * H * E
        issued_flops_per_launch = prof["message_ksteps"] * 32 * 192 * 2.0 * products / max(msg_launches, 1)
        avg_s = msg_ms * 1e-3 / max(msg_launches, 1)
        issued = issued_flops_per_launch / avg_s / 1e12 if avg_s > 0 else 0.0
        dense_equiv = dense_flops_per_launch / avg_s / 1e12 if avg_s > 0 else 0.0
        gathered_bytes = E * 5 * H * 4.0  # per edge: the source's gather record (xa, xc, P0, P1, P2 per channel), from L2
        hbm_alg_bytes = counters.message_bytes_per_layer - E * 3 * H * 4.0 + N_atoms * 4 * H * 4.0  # fused: no rbfh
        traffic = None
        pmc = ROOT / "profiles" / "message_kernel_pmc.json"
        if pmc.exists():
            try:
                traffic = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        gpu_ms = {k: round(v[0] / args.steps, 2) for k, v in prof.items() if isinstance(v, tuple)}
        measured = eng.measure_peaks()  # stream copy + register-resident MFMA loops, on this box, after the timed region
        peak_meas = measured["mfma_f16_tflops"] if f16 else measured["mfma_f32_tflops"]
        out = {
            "metric": METRIC,
            "value": total_systems * args.steps / elapsed,
            "unit": "sites/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32" if not f16 else "f32 (f16x3-split MFMA, fp32 accumulate; ADF_GEMM=f32 selects exact f32)",
            "data": "synthetic",
            "config": {
                "workload": "PaiNN denoiser (H=512, 6 layers, R=128, K=50), %d-step ODE sampling on %d synthetic "
                            "OC20-Dense systems (200 atoms, 10 A cutoff) per GPU" % (args.num_steps, n_local),
                "systems_total": total_systems,
                "systems_per_gpu": n_local,
                "num_reverse_steps": args.num_steps,
                "atoms_per_system": 200,
                "edges_per_system": round(E / max(n_local, 1), 1),
                "weights": "reference initialisers, seed 0, shipped scale factors",
                "parallelism": "systems sharded over %d GPU(s), one all_gather of sites per pass" % world,
                "loop_invariant_reuse": "slab-slab top-K candidates and layer-0 gather records (functions of the static "
                                        "slab / atomic numbers only) are computed at the first of the 50 steps of each "
                                        "pass and reused; every pass starts cold; bit-identical to recomputing "
                                        "(denoising_pos_params['static_atom_cache']=False)",
            },
            "system_steps_per_s": total_systems * args.steps * args.num_steps / elapsed,
            "gpu_ms_per_pass": gpu_ms,
            "roofline": {
                "kernel": "adf_message_kernel (fused rbf-projection MFMA + gather + segmented sum), %s mode"
                          % ("f16x3-split" if f16 else "exact-f32"),
                "bound": "mfma",
                "achieved": issued,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": issued / peak,
                "measured_peak": peak_meas,
                "frac_of_measured_peak": issued / peak_meas if peak_meas > 0 else None,
                "traffic": traffic,
                "avg_launch_ms": avg_s * 1e3,
                "launches": msg_launches,
                "flops_per_launch": issued_flops_per_launch,
                "dense_equivalent_tflops": dense_equiv,
                "algorithmic_f16x3_tflops": dense_equiv * products,
                "l2_gather_tbps": gathered_bytes / avg_s / 1e12 if avg_s > 0 else 0.0,
                "hbm_algorithmic_gbps": hbm_alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0,
                "hbm_algorithmic_frac": (hbm_alg_bytes / avg_s / 1e9) / PEAK_HBM_GBS if avg_s > 0 else 0.0,
                "note": "achieved = matrix-core flops the kernel issues (k-window x 32 x 192 x 2 per 32-edge row "
                        "block, padded rows and the 3 split products included) / launch time from HIP events on the "
                        "launch stream. This is the conservative count: the algorithm's dense contraction (SURVEY 8d: "
                        "2*R*3H*E per layer, x3 products in this arithmetic = algorithmic_f16x3_tflops) is ~1.9x larger, "
                        "the k-window skips Gaussian terms below 1.5e-8 of the leading one. Timed on the "
                        "launch stream; measured_peak = the same MFMA instruction in a register-resident loop on this "
                        "box (non-zero operands). rbfh is never materialised, so neither SURVEY 8d roofline binds alone: "
                        "per 32-edge block the kernel issues ~600 VALU instructions (8 FMA per gathered channel-row) "
                        "beside ~64 MFMAs and 10 KB/edge of L2-served record gathers (l2_gather_tbps; the same access "
                        "pattern alone sustains ~28 TB/s) - see DESIGN.md 4.",
            },
            "measured_peaks": measured,
            "scores_on_adsorbate_only": ads_only,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cpu_sd, scale_factors, args.cpu_systems, args.cpu_steps, params)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
