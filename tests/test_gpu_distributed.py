"""N>1 path with the HIP sampler: two ranks share cuda:0 (gloo backend — RCCL refuses two ranks on one device), each
samples its shard of the same batch, and the gathered sites must equal the single-process run bit for bit.
Reference being replaced: sde_denoising_trainer.py:862-909 (per-rank npz + merge), datasets/data_parallel.py:32-48."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


WORKER = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from adsorbdiff_amd.data import Batch
from adsorbdiff_amd.denoising_torch import Denoiser, DiffTorchCalc
from adsorbdiff_amd.painn_denoising import PaiNN
from adsorbdiff_amd.sampler import gather_sites, shard_batch
from adsorbdiff_amd.synthetic import make_batch
from adsorbdiff_amd.trainer import DenoisingTrainer

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo")
torch.manual_seed(0)
model = PaiNN(None, 50, 1, hidden_channels=128, num_layers=2, cutoff=6.0, max_neighbors=20, so3_denoising=True).eval()
trainer = DenoisingTrainer(model, device="cuda:0")
full = Batch.from_data_list(make_batch(4, n_slab=36, n_ads=3, seed=5).to_data_list()
                            + make_batch(3, n_slab=64, n_ads=4, seed=6).to_data_list())
B = len(full.natoms)
if world > 1:
    mine, ids = shard_batch(full, rank, world)
else:
    mine, ids = full, list(range(B))
torch.manual_seed(0)
noise = torch.rand(B, 3)[torch.tensor(ids)]
params = dict(num_steps=4, ads_std_low=0.1, ads_std_high=10, rot_std_low=0.01, rot_std_high=1.55, ode=True,
              early_stop=False, placement_noise=noise)
out = Denoiser(mine.to("cuda:0"), DiffTorchCalc(trainer), params, device="cuda:0").run()
sites = gather_sites(out, world, system_ids=ids)
if rank == 0:
    torch.save({"sites": sites.cpu(), "ids": ids}, sys.argv[2])
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
"""


def _run(world, out, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), str(ROOT), str(out)], env=env))
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0] * world, codes
    return torch.load(out)


def test_two_ranks_on_one_gpu_equal_single_rank(tmp_path):
    one = _run(1, tmp_path / "w1.pt", tmp_path)
    two = _run(2, tmp_path / "w2.pt", tmp_path)
    assert one["sites"].shape == (7, 4, 3) and sorted(two["ids"]) != list(range(7))
    assert torch.equal(torch.nan_to_num(one["sites"]), torch.nan_to_num(two["sites"]))


@pytest.mark.parametrize("model", ["painn", "eqv2"])
def test_bench_launches_its_own_ranks(tmp_path, model):
    """`python bench.py --gpus 2 --backend gloo` on one GPU: n_gpus 2 in the JSON line, same sites as --gpus 1 - with
    either score model (BASELINE config 3 sharding; config 4's model through the same launcher)."""
    def bench(n):
        cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(n), "--backend", "gloo", "--model", model,
               "--systems", "6" if model == "painn" else "4", "--num-steps", "3" if model == "painn" else "2",
               "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary"]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        assert res.returncode == 0, res.stderr[-2000:]
        return json.loads(res.stdout.strip().splitlines()[-1])

    a, b = bench(1), bench(2)
    total = 6 if model == "painn" else 4
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["scaling"] == "strong"
    assert b["config"]["systems_total"] == total and b["config"]["systems_per_gpu"] == total // 2
    assert a["sites_sha256_16"] == b["sites_sha256_16"]


def test_eight_way_split_of_the_benchmark_batch_samples_the_single_run_sites():
    """BASELINE config 3's decomposition on the REAL 1000-system batch: the eight shards of `shard_batch` (125 systems each, the
    reference's balanced partition) are sampled one after the other in THIS process with the placement noise keyed by global
    system id, each packs its `[B_max, 1 + 3 A_max]` message exactly as `gather_sites` does, the eight messages are merged
    the way the all-gather's result is - and the merged sites equal the single 1000-system run's bit for bit.  (Eight
    PROCESSES of this size on one device were tried first: they take turns with whole-chip context switches - 4 ranks x 250
    systems x 3 steps took 6 min, 8 x 125 took 17 min against 5 s of GPU work - so the process-level path is covered on
    small batches below and by the 2-rank tests; the partition, the noise keying and the merge are what this test pins.)"""
    from adsorbdiff_amd.denoising_torch import Denoiser, DiffTorchCalc
    from adsorbdiff_amd.painn_denoising import PaiNN
    from adsorbdiff_amd.sampler import adsorbate_sites, merge_packed_sites, pack_sites, shard_batch, shard_bounds
    from adsorbdiff_amd.scaling import PAINN_NB6_SCALE_FACTORS
    from adsorbdiff_amd.synthetic import make_batch
    from adsorbdiff_amd.trainer import DenoisingTrainer

    torch.manual_seed(0)
    model = PaiNN(None, 50, 1, hidden_channels=512, num_layers=6, num_rbf=128, cutoff=10.0, max_neighbors=50,
                  scale_file=PAINN_NB6_SCALE_FACTORS, so3_denoising=True).eval()
    trainer = DenoisingTrainer(model, device="cuda:0")
    full = make_batch(1000, seed=1000)
    torch.manual_seed(0)
    placement = torch.rand(1000, 3)
    params = dict(num_steps=3, ads_std_low=0.1, ads_std_high=10, rot_std_low=0.01, rot_std_high=1.55, ode=True, early_stop=False)

    def sample(batch, ids):
        out = Denoiser(batch.to("cuda:0"), DiffTorchCalc(trainer),
                       dict(params, placement_noise=placement[torch.tensor(ids, dtype=torch.long)]), device="cuda:0").run()
        return adsorbate_sites(out).cpu()

    one = sample(full.clone(), list(range(1000)))
    bounds = shard_bounds(full, 8)
    assert bounds == (125, 4)
    messages, seen = [], []
    for r in range(8):
        mine, ids = shard_batch(full, r, 8)
        assert len(ids) == 125
        seen += ids
        messages.append(pack_sites(sample(mine, ids), ids, bounds))
    assert sorted(seen) == list(range(1000)) and seen != list(range(1000))   # a partition, and not the trivial one
    merged = merge_packed_sites(torch.stack(messages), bounds[1], ordered=True)
    assert merged.shape == one.shape == (1000, 4, 3)
    assert torch.equal(merged, one)


def test_bench_with_eight_ranks_reports_per_rank_times(tmp_path):
    """`bench.py --gpus 8 --backend gloo` (eight processes on cuda:0) on a SIX-system batch: two ranks are dealt nothing and only
    join the exchange; same site digest as --gpus 1, and the line carries every rank's wall and GPU-busy time, its share of
    the batch and the max / mean imbalance - what the first real 8-GPU run will need to explain its curve."""
    def bench(n):
        cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(n), "--backend", "gloo", "--systems", "6", "--steps", "1",
               "--warmup", "0", "--num-steps", "3", "--no-cpu-baseline", "--no-secondary"]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        assert res.returncode == 0, res.stderr[-2000:]
        return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])

    one, eight = bench(1), bench(8)
    assert eight["n_gpus"] == 8 and eight["config"]["systems_total"] == 6 and eight["scaling"] == "strong"
    assert one["sites_sha256_16"] == eight["sites_sha256_16"]
    pr = eight["per_rank"]
    assert len(pr["ms_per_step"]) == 8 and sorted(pr["systems"]) == [0, 0, 1, 1, 1, 1, 1, 1] and sum(pr["atoms"]) == 1200
    assert pr["imbalance_max_over_mean"] >= 1.0 and abs(max(pr["ms_per_step"]) - eight["ms_per_step"]) < 1.0
    assert one["per_rank"] is None


def test_bench_under_torch_distributed_run(tmp_path):
    """The driver's launch form: `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` (gloo here: two
    ranks share the one GPU of this box).  Rank 0 prints the JSON line; same sites as the self-launched run."""
    args = ["--gpus", "2", "--backend", "gloo", "--systems", "6", "--num-steps", "3", "--steps", "1", "--warmup", "0",
            "--no-cpu-baseline", "--no-secondary"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py")] + args
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    a = json.loads(lines[0])
    res2 = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=900, env=env)
    assert res2.returncode == 0, res2.stderr[-2000:]
    b = json.loads(res2.stdout.strip().splitlines()[-1])
    assert a["n_gpus"] == 2 and a["sites_sha256_16"] == b["sites_sha256_16"]


def test_rccl_allgather_c_abi_single_rank():
    """adf_comm_* / adf_allgather_sites through RCCL with a 1-rank communicator (all this box can host)."""
    import ctypes as C

    from adsorbdiff_amd import lib as L

    lib = L.load()
    buf = (C.c_uint8 * 128)()
    L.check(lib.adf_comm_unique_id(buf))
    comm = C.c_void_p()
    L.check(lib.adf_comm_create(buf, 0, 1, C.byref(comm)))
    x = torch.randn(5, 4, 3, device="cuda:0")
    y = torch.empty(1, 5, 4, 3, device="cuda:0")
    L.check(lib.adf_allgather_sites(comm, x.data_ptr(), x.numel() * 4, y.data_ptr(),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert torch.equal(y[0], x)
    L.check(lib.adf_comm_destroy(comm))


def _nccl_train_worker(rank, world, port, out_dir):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from adsorbdiff_amd.painn_denoising import PaiNN
    from adsorbdiff_amd.so3_tables import Igso3Tables
    from adsorbdiff_amd.synthetic import make_batch
    from adsorbdiff_amd.trainer import DenoisingTrainer

    torch.manual_seed(0)
    m = PaiNN(None, 50, 1, hidden_channels=128, num_layers=2, num_rbf=128, cutoff=6.0, max_neighbors=20, so3_denoising=True,
              scale_file={"upd_out_scalar_scale_0": 1.0, "upd_out_scalar_scale_1": 1.0})
    tr = DenoisingTrainer(m, device=f"cuda:{rank}")
    tr.setup_training(dict(ads_std_low=0.1, ads_std_high=10, rot_std_low=0.01, rot_std_high=1.55), lr=0.0,
                      tables=Igso3Tables.shared())
    out = tr.train_step(make_batch(2, n_slab=36, n_ads=4, seed=100 + rank).to(f"cuda:{rank}"))
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None]).cpu()
    torch.save({"g": g, "loss": out["loss"].cpu()}, os.path.join(out_dir, f"n{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL all-reduce between devices)")
def test_two_gpu_nccl_training_step_averages_gradients(tmp_path):
    """Backend nccl (= RCCL) over two devices: the overlapped bucketed all-reduce of a training step leaves the SAME averaged
    gradients on both ranks (each rank had its own batch).  Runs whenever the box shows two GPUs; the 1-GPU pool skips it."""
    import torch.multiprocessing as mp

    mp.spawn(_nccl_train_worker, args=(2, 29671, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "n0.pt"), torch.load(tmp_path / "n1.pt")
    assert torch.equal(a["g"], b["g"]) and bool(torch.isfinite(a["g"]).all()) and float(a["g"].abs().max()) > 0


_EMPTY_RANK_SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, {root!r})
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29683", HSA_ENABLE_IPC_MODE_LEGACY="0")
from adsorbdiff_amd.sampler import adsorbate_sites, gather_sites
assert not adsorbate_sites(None).is_cuda                      # no process group: host
assert gather_sites(None, 1, via="rccl").is_cuda              # the library's RCCL entry wants device memory
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
e = adsorbate_sites(None)
assert e.is_cuda and tuple(e.shape) == (0, 1, 3), (e.device, e.shape)
g = gather_sites(None, 1)
assert g.is_cuda
# the message an empty rank sends: all padding, on the device the collective runs on
from adsorbdiff_amd.sampler import pack_sites
p = pack_sites(e, [], (3, 2))
assert p.is_cuda and bool((p[:, 0] == -1).all())
outs = [torch.empty_like(p)]
dist.all_gather(outs, p)                                      # RCCL accepts it (a CPU tensor raises here)
assert torch.equal(outs[0], p)
dist.destroy_process_group()
print("EMPTY_RANK_OK")
"""


def test_rank_without_systems_sends_its_message_from_the_device_under_nccl():
    """`gather_sites(None, ...)` without `local=` built its empty message on the host; under backend nccl (and
    via="rccl") that raises on the empty rank while the others wait in the all-gather.  One-rank nccl group in a child."""
    res = subprocess.run([sys.executable, "-c", _EMPTY_RANK_SCRIPT.format(root=str(ROOT))], capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0 and "EMPTY_RANK_OK" in res.stdout, res.stderr[-2000:]
