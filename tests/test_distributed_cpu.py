"""N>1 path on CPU: two gloo ranks shard a batch by atom count and all_gather their sites."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from adsorbdiff_amd.data import Batch
from adsorbdiff_amd.sampler import adsorbate_sites, gather_sites, shard_batch, shard_bounds
from adsorbdiff_amd.synthetic import make_batch


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = Batch.from_data_list(
        make_batch(3, n_slab=16, n_ads=2, seed=7).to_data_list() + make_batch(2, n_slab=36, n_ads=4, seed=8).to_data_list()
    )
    mine, ids = shard_batch(full, rank, world)
    mine.pos = mine.pos + float(rank + 1)  # stand-in for "sampled" positions
    calls = {"n": 0}
    real = dist.all_gather

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    dist.all_gather = counting
    sites = gather_sites(mine, world)                                      # shapes agreed on first: 2 collectives
    n_plain = calls["n"]
    ordered = gather_sites(mine, world, system_ids=ids, bounds=shard_bounds(full, world))  # ONE collective
    dist.all_gather = real
    torch.save({"sites": sites, "ids": ids, "ordered": ordered, "collectives": (n_plain, calls["n"] - n_plain)},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    # every rank holds the same gathered tensor: all 5 systems, rank-major, NaN-padded to 4 atoms
    assert r0["sites"].shape == (5, 4, 3)
    assert r0["collectives"] == (2, 1) and r1["collectives"] == (2, 1)  # with locally derived bounds: one all_gather
    # global system order when ids are passed: row g = system g
    assert r0["ordered"].shape == (5, 4, 3) and torch.equal(torch.nan_to_num(r0["ordered"]), torch.nan_to_num(r1["ordered"]))
    gid = torch.tensor(r0["ids"] + r1["ids"])
    assert torch.equal(torch.nan_to_num(r0["ordered"]), torch.nan_to_num(r0["sites"][torch.argsort(gid)]))
    assert torch.equal(torch.nan_to_num(r0["sites"]), torch.nan_to_num(r1["sites"]))
    assert sorted(r0["ids"] + r1["ids"]) == [0, 1, 2, 3, 4]
    # atom-count balanced: the two 40-atom systems land on different ranks
    assert (3 in r0["ids"]) != (4 in r0["ids"])
    # content check against a single-process evaluation
    full = Batch.from_data_list(
        make_batch(3, n_slab=16, n_ads=2, seed=7).to_data_list() + make_batch(2, n_slab=36, n_ads=4, seed=8).to_data_list()
    )
    data = full.to_data_list()
    want = []
    for rank, ids in enumerate((r0["ids"], r1["ids"])):
        sub = Batch.from_data_list([data[i] for i in ids])
        sub.pos = sub.pos + float(rank + 1)
        s = adsorbate_sites(sub)
        pad = torch.full((s.shape[0], 4, 3), float("nan"))
        pad[:, : s.shape[1]] = s
        want.append(pad)
    want = torch.cat(want)
    assert torch.equal(torch.nan_to_num(r0["sites"]), torch.nan_to_num(want))


def _empty_rank_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = make_batch(2, n_slab=16, n_ads=3, seed=9)      # two systems, three ranks: one rank is dealt nothing
    mine, ids = shard_batch(full, rank, world)
    bounds = shard_bounds(full, world)
    if ids:
        mine.pos = mine.pos + float(rank + 1)
        got = gather_sites(mine, world, system_ids=ids, bounds=bounds)
    else:
        got = gather_sites(None, world, system_ids=ids, bounds=bounds)
    torch.save({"sites": got, "ids": ids}, os.path.join(out_dir, f"e{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_without_systems_joins_the_exchange(tmp_path):
    """More ranks than systems (a user sampling 2 structures on a 3-GPU launch): the surplus rank sends an all-padding message
    and every rank still receives both systems' sites in global order."""
    world = 3
    mp.spawn(_empty_rank_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"e{r}.pt") for r in range(world)]
    assert sorted(len(r["ids"]) for r in res) == [0, 1, 1]
    for r in res:
        assert r["sites"].shape == (2, 3, 3) and bool(torch.isfinite(r["sites"]).all())
        assert torch.equal(r["sites"], res[0]["sites"])


def _grad_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adsorbdiff_amd.train_step import allreduce_gradients

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4), torch.nn.Linear(4, 2))
    for p in net[2].parameters():
        p.grad = None                      # a parameter no rank has a gradient for (the reference's unused heads)
    for i, p in enumerate(list(net[0].parameters()) + list(net[1].parameters())):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    allreduce_gradients(net, world, bucket_mb=1e-4)  # tiny buckets: several all-reduces
    torch.save([None if p.grad is None else p.grad.clone() for p in net.parameters()], os.path.join(out_dir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce(tmp_path):
    """DDP step of the training path: bucketed average over ranks, parameters without a gradient are skipped."""
    world = 2
    mp.spawn(_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    for i, (a, b) in enumerate(zip(g0, g1)):
        if i >= 4:
            assert a is None and b is None
        else:
            assert torch.equal(a, b) and torch.allclose(a, torch.full_like(a, 1.5 * (i + 1)))


def _reducer_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adsorbdiff_amd.train_step import GradientReducer

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.atom_emb = torch.nn.Linear(4, 4)
            self.message_layers = torch.nn.ModuleList([torch.nn.Linear(4, 4) for _ in range(2)])
            self.update_layers = torch.nn.ModuleList([torch.nn.Linear(4, 4) for _ in range(2)])
            self.out_forces = torch.nn.Linear(4, 2)
            self.out_energy = torch.nn.Linear(4, 1)   # never gets a gradient (find_unused_parameters semantics)

    torch.manual_seed(0)
    net = Net()
    for i, (k, p) in enumerate(net.named_parameters()):
        p.grad = None if k.startswith("out_energy") else torch.full_like(p, float(rank + 1) * (i + 1))
    red = GradientReducer(net, world, bucket_mb=1e-5)   # a bucket per tensor: many collectives in flight at once
    calls = []
    orig = dist.all_reduce

    def counting(t, *a, **k):
        calls.append(t.numel())
        return orig(t, *a, **k)

    dist.all_reduce = counting
    # the order the backward announces its groups in (train_step.loss_and_grad): heads, layers from the last, embedding
    red.ready(["out_forces.", "out_forces2."])
    n_heads = len(calls)
    for l in (1, 0):
        red.ready([f"message_layers.{l}.", f"update_layers.{l}."])
    n_layers = len(calls)
    red.finish()                                         # picks up what was never announced (atom_emb)
    dist.all_reduce = orig
    torch.save({"grads": [None if p.grad is None else p.grad.clone() for p in net.parameters()],
                "calls": (n_heads, n_layers, len(calls))}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_reducer_issues_buckets_from_the_backward_hooks(tmp_path):
    """train_step.GradientReducer (the all-reduce overlapped with the backward): the buckets of a group are
    started when the group is announced, what is never announced goes out at finish(), every gradient ends up as the
    average over the ranks and gradient-less parameters are skipped on every rank."""
    world = 2
    mp.spawn(_reducer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["calls"] == r1["calls"] == (2, 10, 12)     # heads: 2 tensors; + 4 layers x 2; + embedding 2 at finish
    n = 0
    for i, (a, b) in enumerate(zip(r0["grads"], r1["grads"])):
        if a is None:
            assert b is None
            n += 1
        else:
            assert torch.equal(a, b) and torch.allclose(a, torch.full_like(a, 1.5 * (i + 1)))
    assert n == 2
