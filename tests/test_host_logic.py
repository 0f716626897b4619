"""CPU tests of the host-side mirror: containers, schedule scalars, repeat counts, weights table,
C-ABI surface.  No GPU compute is called."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from adsorbdiff_amd import lib as L
from adsorbdiff_amd.data import Batch, Data
from adsorbdiff_amd.denoising_torch import schedule_coefs
from adsorbdiff_amd.engine import batch_pbc, cell_repeats
from adsorbdiff_amd.painn_denoising import PaiNN
from adsorbdiff_amd.sampler import adsorbate_sites, balanced_partition
from adsorbdiff_amd.scaling import PAINN_NB6_SCALE_FACTORS, ScaleFactor, ensure_fitted
from adsorbdiff_amd.synthetic import make_batch
from oracle import painn_oracle as O

ROOT = Path(__file__).resolve().parent.parent


def test_c_abi_exports_match_header():
    hdr = (ROOT / "include" / "adsorbdiff_hip.h").read_text()
    declared = set(re.findall(r"\b(adf_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"adf_painn"}  # struct tag
    lib = L.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert set(L.EXPORTS) == declared
    assert b"gfx950" in lib.adf_version()


def test_no_cpu_fallback():
    m = PaiNN(None, 50, 1, hidden_channels=128, num_layers=1, so3_denoising=True)
    b = make_batch(1, n_slab=16, n_ads=2, seed=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(b)


def test_invalid_hparams_rejected():
    lib = L.load()
    hp = L.Hparams(hidden_channels=100, num_layers=1, num_rbf=128, num_elements=83, max_neighbors=50,
                   envelope_exponent=5, num_heads=2, cutoff=6.0)
    h = ctypes.c_void_p()
    assert lib.adf_painn_create(ctypes.byref(hp), ctypes.byref(h)) == L.ADF_EINVAL
    assert b"hidden_channels" in lib.adf_last_error()
    with pytest.raises(ValueError):
        L.check(L.ADF_ENONEIGHBOR)
    with pytest.raises(RuntimeError):
        L.check(L.ADF_EOOM)


def test_state_dict_layout_matches_reference_keys():
    m = PaiNN(None, 50, 1, so3_denoising=True, scale_file=PAINN_NB6_SCALE_FACTORS)
    sd = m.state_dict()
    assert sd["atom_emb.embeddings.weight"].shape == (83, 512)
    assert sd["message_layers.5.x_proj.2.weight"].shape == (1536, 512)
    assert sd["message_layers.0.rbf_proj.weight"].shape == (1536, 128)
    assert sd["update_layers.3.vec_proj.weight"].shape == (1024, 512)
    assert sd["update_layers.3.xvec_proj.0.weight"].shape == (512, 1024)
    assert sd["out_forces2.output_network.1.update_net.2.weight"].shape == (2, 256)
    assert sd["radial_basis.rbf.offset"].shape == (128,)
    assert float(sd["upd_out_scalar_scale_0.scale_factor"]) == pytest.approx(1.0364354848861694)
    assert m.num_params == 21451888  # SURVEY.md §6 [probe]
    assert m.scale_factors()[5] == pytest.approx(0.8822302222251892)
    ensure_fitted(m)
    cond = PaiNN(None, 50, 1, hidden_channels=128, num_layers=1, so3_denoising=True, energy_encoding="scalar")
    assert "energy_embedding.weight" in cond.state_dict() and "concat_lin.0.weight" in cond.state_dict()
    with pytest.raises(ValueError):
        ensure_fitted(cond)
    sf = ScaleFactor()
    x = torch.ones(3)
    assert torch.equal(sf(x), x)  # unfitted -> identity (scale_factor.py:166-167)
    sf.set_(2.0)
    assert torch.equal(sf(x), 2 * x)


def test_schedule_coefs_match_oracle_scalars():
    for ode in (True, False):
        params = dict(num_steps=50, ads_std_low=0.1, ads_std_high=10, rot_std_low=0.01, rot_std_high=1.55, ode=ode)
        cs = schedule_coefs(params)
        assert len(cs) == 50
        for t in (0, 1, 25, 49):
            tr_g, rot_g, dt = O.schedule_scalars(t, 50, 0.1, 10, 0.01, 1.55)
            s = torch.tensor([[0.3, -0.7, 0.11]])
            if ode:
                want_tr = 0.5 * tr_g**2 * dt * s
                want_rot = 0.5 * s * dt * rot_g**2
            else:
                want_tr = tr_g**2 * dt * s
                want_rot = s * dt * rot_g**2
            got_tr = torch.tensor(cs[t].coef_tr, dtype=torch.float32) * s
            got_rot = ((torch.tensor(cs[t].rot_pre) * s) * torch.tensor(cs[t].rot_dt)) * torch.tensor(cs[t].rot_g2)
            assert torch.equal(got_tr, want_tr)
            assert torch.allclose(got_rot, want_rot.float(), rtol=2e-7, atol=0)
    assert cs[0].noise_tr > 0 and schedule_coefs(dict(params, ode=True))[0].noise_tr == 0


def test_cell_repeats_and_pbc():
    b = make_batch(3, n_slab=36, n_ads=4, seed=2)
    assert cell_repeats(b.cell, 6.0) == O.cell_repeats(b.cell, 6.0) == [2, 2, 1]
    big = make_batch(2, seed=3)
    assert cell_repeats(big.cell, 10.0) == [1, 1, 1]
    assert cell_repeats(big.cell, 10.0, (True, True, False)) == [1, 1, 0]
    assert batch_pbc(b) == [True, True, True]
    b.pbc = torch.tensor([[True, True, False]] * 3)
    assert batch_pbc(b) == [True, True, False]
    b.pbc = torch.tensor([[True, True, False], [True, True, True], [True, True, True]])
    with pytest.raises(RuntimeError):
        batch_pbc(b)


def test_batch_collate_roundtrip_and_sites():
    b = make_batch(3, n_slab=16, n_ads=3, seed=4)
    parts = b.to_data_list()
    assert [int(p.natoms) for p in parts] == [19, 19, 19]
    b2 = Batch.from_data_list(parts)
    assert torch.equal(b2.pos, b.pos) and torch.equal(b2.batch, b.batch) and b2.sid == b.sid
    nested = Batch.from_data_list([Batch.from_data_list(parts[:1]), Batch.from_data_list(parts[1:])])
    assert torch.equal(nested.cell, b.cell)
    sites = adsorbate_sites(b)
    assert sites.shape == (3, 3, 3)
    assert torch.equal(sites[1], b.pos[(b.batch == 1) & (b.tags == 2)])
    assert balanced_partition([5, 9, 3, 7, 1], 2) == [[1, 2, 4], [0, 3]]


def test_synthetic_batch_shape():
    b = make_batch(2, seed=1000)
    assert b.pos.shape == (400, 3) and b.natoms.tolist() == [200, 200]
    assert int((b.tags == 2).sum()) == 8 and int(b.atomic_numbers.max()) <= 83
    assert torch.equal(b.fixed, (b.tags == 0).long())
    assert torch.equal(make_batch(2, seed=1000).pos, b.pos)  # deterministic under the seed


def test_ml_diffuse_splits_on_runtime_error(monkeypatch):
    """RuntimeError (what the HIP library reports for device OOM) -> batch split in halves and retried;
    a single-system failure is re-raised (reference ml_relaxation.py:146-165)."""
    import adsorbdiff_amd.ml_relaxation as R

    seen = []

    class FakeDenoiser:
        def __init__(self, batch, calc, **kw):
            self.batch = batch

        def run(self):
            n = len(self.batch.natoms)
            seen.append(n)
            if n > 2:
                raise RuntimeError("HIP out of memory")
            self.batch.pos = self.batch.pos + 1.0
            return self.batch

    monkeypatch.setattr(R, "Denoiser", FakeDenoiser)
    b = make_batch(5, n_slab=16, n_ads=2, seed=9)
    out = R.ml_diffuse(b, model=object(), denoising_pos_params={}, traj_dir=None, save_full_traj=False, device="cpu")
    assert seen[0] == 5 and max(seen[1:]) <= 3 and sorted(out.sid) == sorted(b.sid)
    # the reference pushes both halves on the left of its deque, first half first: the upper half is sampled first
    assert seen == [5, 3, 2, 1, 2] and list(out.sid) == [b.sid[i] for i in (3, 4, 2, 0, 1)]
    assert torch.allclose(out.pos.sum(), b.pos.sum() + 3.0 * b.pos.shape[0])

    class AlwaysFail(FakeDenoiser):
        def run(self):
            raise RuntimeError("boom")

    monkeypatch.setattr(R, "Denoiser", AlwaysFail)
    with pytest.raises(RuntimeError, match="boom"):
        R.ml_diffuse(make_batch(2, n_slab=16, n_ads=2, seed=9), object(), {}, None, False, device="cpu")


def test_checkpoint_ingest(tmp_path):
    """Reference checkpoint layout: state_dict with module. prefixes + ema shadow params (base_trainer.py:456-533)."""
    from adsorbdiff_amd.trainer import DenoisingTrainer

    torch.manual_seed(5)
    src = PaiNN(None, 50, 1, hidden_channels=128, num_layers=1, so3_denoising=True)
    sd = {"module.module." + k: v for k, v in src.state_dict().items()}
    shadow = [p.detach() * 0 + 0.25 for p in src.parameters() if p.requires_grad]
    torch.save({"state_dict": sd, "ema": {"shadow_params": shadow}, "config": {}}, tmp_path / "ckpt.pt")
    dst = PaiNN(None, 50, 1, hidden_channels=128, num_layers=1, so3_denoising=True)
    tr = DenoisingTrainer.__new__(DenoisingTrainer)
    tr.model = dst
    tr.load_checkpoint(str(tmp_path / "ckpt.pt"))
    for (k, p) in dst.named_parameters():
        if p.requires_grad:
            assert torch.all(p == 0.25), k
    assert torch.equal(dst.state_dict()["radial_basis.rbf.offset"], src.state_dict()["radial_basis.rbf.offset"])


def test_atoms_conversions_roundtrip():
    from adsorbdiff_amd.calculator import SimpleAtoms, atoms_to_data, batch_to_atoms

    b = make_batch(1, n_slab=16, n_ads=2, seed=21)
    a = SimpleAtoms(b.atomic_numbers.long().numpy(), b.pos.numpy(), b.cell[0].numpy(), b.tags.numpy(), b.fixed.numpy())
    d = atoms_to_data(a, sid="s")
    assert torch.allclose(d.pos, b.pos) and torch.equal(d.tags, b.tags) and torch.equal(d.fixed, b.fixed)
    assert d.cell.shape == (1, 3, 3) and d.pbc.tolist() == [[True, True, True]] and int(d.natoms) == 18
    back = batch_to_atoms(Batch.from_data_list([d]))[0]
    assert np.allclose(back.get_positions(), b.pos.numpy()) and list(back.get_tags()) == b.tags.tolist()


def test_balanced_batch_sampler_is_collective_free_and_balanced():
    """Every rank derives all ranks' index streams locally (DistributedSampler is a pure function of seed / epoch /
    rank): the per-step batches are disjoint, cover the step's global batch and are balanced by atom count."""
    from torch.utils.data import DistributedSampler

    from adsorbdiff_amd.data_parallel import BalancedBatchSampler, distributed_indices

    rng = np.random.default_rng(0)
    sizes = rng.integers(20, 220, size=203)
    # the index stream is torch's DistributedSampler's
    for rank in range(4):
        ds = DistributedSampler(list(range(203)), num_replicas=4, rank=rank, shuffle=True, seed=3)
        ds.set_epoch(2)
        assert list(ds) == distributed_indices(203, 4, rank, True, 3, 2)
    samplers = [BalancedBatchSampler(sizes, batch_size=8, num_replicas=4, rank=r, shuffle=True, seed=3) for r in range(4)]
    for s in samplers:
        s.set_epoch(2)
    streams = [list(s) for s in samplers]
    assert len({len(st) for st in streams}) == 1 and len(streams[0]) == len(samplers[0])
    plain = [[distributed_indices(203, 4, r, True, 3, 2)[i : i + 8] for i in range(0, 51, 8)] for r in range(4)]
    for step in range(len(streams[0])):
        got = sorted(i for r in range(4) for i in streams[r][step])
        assert got == sorted(i for r in range(4) for i in plain[r][step])          # same global batch, re-dealt
        loads = [int(sizes[streams[r][step]].sum()) for r in range(4)]
        naive = [int(sizes[plain[r][step]].sum()) for r in range(4)]
        assert max(loads) - min(loads) <= max(sizes) and max(loads) <= max(naive)
    single = BalancedBatchSampler(sizes, 8, 1, 0, shuffle=False)
    assert list(single)[0] == list(range(8))


def test_record_dataset_roundtrip(tmp_path):
    from adsorbdiff_amd.data_parallel import OCPCollater, RecordDataset

    b = make_batch(3, n_slab=16, n_ads=2, seed=4)
    recs = {}
    for i, d in enumerate(b.to_data_list()):
        for k, v in (("pos", d.pos.numpy()), ("cell", d.cell.numpy()), ("atomic_numbers", d.atomic_numbers.numpy()),
                     ("natoms", int(d.natoms)), ("tags", d.tags.numpy()), ("fixed", d.fixed.numpy()), ("sid", str(d.sid)), ("fid", 0)):
            recs[f"{i}/{k}"] = v
    np.savez(tmp_path / "r.npz", length=3, **recs)
    ds = RecordDataset(tmp_path / "r.npz")
    assert len(ds) == 3 and ds.natoms.tolist() == [18, 18, 18]
    back = OCPCollater()([ds[i] for i in range(3)])
    assert torch.equal(back.pos, b.pos) and torch.equal(back.tags, b.tags) and back.sid == b.sid


def test_balanced_partition_and_sampler_equal_the_reference_fixture():
    """datasets/data_parallel.py:32-48 (balanced_partition, output order included) and :165-200 (the per-step balancing of
    BalancedBatchSampler, every rank) on the seeded cases recorded from the reference (oracle/make_golden.py section 8)."""
    import numpy as np

    from adsorbdiff_amd.data_parallel import BalancedBatchSampler, balanced_partition_ref
    from tests.helpers import load_npz

    fx = load_npz("balanced_partition.npz")
    for name in ("rand8", "ties4", "two", "more_parts"):
        parts = int(fx[f"{name}/parts"])
        got = balanced_partition_ref(fx[f"{name}/sizes"], parts)
        assert got == [fx[f"{name}/part{r}"].tolist() for r in range(parts)], name
    world, bs, seed, epoch, steps = [int(v) for v in fx["sampler/meta"]]
    for rank in range(world):
        s = BalancedBatchSampler(fx["sampler/sizes"], bs, world, rank, mode="atoms", shuffle=True, seed=seed)
        s.set_epoch(epoch)
        got = [list(b) for b in s]
        assert len(got) == steps
        assert got == [fx[f"sampler/rank{rank}/step{i}"].tolist() for i in range(steps)]


def test_final_frame_records_append_per_batch(tmp_path):
    """write_final_frames is called once per batch with a running start_index: the npz sink keeps the earlier records."""
    import numpy as np

    from adsorbdiff_amd.data_parallel import RecordDataset
    from adsorbdiff_amd.handoff import write_final_frames
    from adsorbdiff_amd.synthetic import make_batch

    a, b = make_batch(2, n_slab=16, n_ads=2, seed=1), make_batch(3, n_slab=16, n_ads=2, seed=2, sid_offset=2)
    out = write_final_frames(a, tmp_path / "frames.lmdb", apply_lift=False)
    out2 = write_final_frames(b, tmp_path / "frames.lmdb", start_index=2, apply_lift=False)
    assert out == out2
    ds = RecordDataset(out)
    assert len(ds) == 5
    assert [ds[i].sid for i in range(5)] == ["0", "1", "2", "3", "4"]
    assert np.allclose(ds[0].pos.numpy(), a.pos[:18].numpy()) and np.allclose(ds[4].pos.numpy(), b.pos[36:].numpy())
    import pytest

    with pytest.raises(ValueError):
        write_final_frames(b, tmp_path / "frames.lmdb", start_index=3, apply_lift=False)


def test_igso3_table_cache_is_atomic_and_validated(tmp_path, monkeypatch):
    """so3_tables: the cache file is written under a temporary name and renamed; an unreadable or mis-shaped file is
    ignored (several ranks may race on first use)."""
    import numpy as np

    from adsorbdiff_amd import so3_tables as S

    cache = tmp_path / "c" / "igso3.npz"
    monkeypatch.setattr(S, "_CACHE", cache)
    assert S.Igso3Tables._load_cache() is None
    t = {"omegas": np.linspace(0, np.pi, S.X_N), "cdf": np.zeros((S.N_EPS, S.X_N), np.float32),
         "score": np.ones((S.N_EPS, S.X_N), np.float32), "exp_score_norm": np.arange(S.N_EPS, dtype=np.float64)}
    S.Igso3Tables._store_cache(t)
    assert cache.exists() and not list(cache.parent.glob("*.tmp.npz"))
    back = S.Igso3Tables._load_cache()
    assert back is not None and np.array_equal(back["exp_score_norm"], t["exp_score_norm"])
    cache.write_bytes(cache.read_bytes()[:1000])  # a partially written file
    assert S.Igso3Tables._load_cache() is None
    np.savez(cache, omegas=t["omegas"][:10], cdf=t["cdf"], score=t["score"], exp_score_norm=t["exp_score_norm"])
    assert S.Igso3Tables._load_cache() is None  # wrong shape


class _FakeFrameSource:
    """Stand-in for trajectory.FrameSink: frames appear from another thread, slots must be released."""

    def __init__(self, frames, slots=3):
        import threading

        self.frames, self.slots = frames, slots
        self.avail = 0
        self.released = set()
        self.cv = threading.Condition()

    def push(self):
        with self.cv:
            # like adf_frames_push: blocks while the slot of frame (avail - slots) is unreleased
            self.cv.wait_for(lambda: self.avail < self.slots or (self.avail - self.slots) in self.released, timeout=5)
            assert self.avail < self.slots or (self.avail - self.slots) in self.released, "ring overrun"
            self.avail += 1
            self.cv.notify_all()

    def pushed(self):
        with self.cv:
            return self.avail

    def wait(self, index, timeout_ms):
        with self.cv:
            if not self.cv.wait_for(lambda: self.avail > index, timeout=timeout_ms / 1000.0):
                return None
        return self.frames[index]

    def release(self, index):
        with self.cv:
            self.released.add(index)
            self.cv.notify_all()


def test_trajectory_writer_thread_streams_trims_and_renames(tmp_path):
    """trajectory.TrajectoryWriter (asynchronous, batched sink) against a fake frame source with a 3-slot
    ring: frames are taken while they are still being produced, the run's early stop trims them (pushed 7, applied 5),
    the slots of the dropped frames are released, and the per-system files the resume rule reads (<sid>.npz) plus the batch
    file and its index exist under their final names only."""
    import json
    import threading

    from adsorbdiff_amd.trainer import check_traj_files
    from adsorbdiff_amd.trajectory import TrajectoryWriter

    rng = np.random.default_rng(0)
    natoms = [5, 3, 4]
    N, T = sum(natoms), 8
    frames = rng.normal(size=(7, N, 3)).astype(np.float32)
    meta = dict(numbers=rng.integers(1, 80, N), tags=rng.integers(0, 3, N), fixed=rng.integers(0, 2, N),
                cell=rng.normal(size=(3, 3, 3)).astype(np.float32), natoms=np.array(natoms), names=["a_1", "b_2", "c_3"])
    src = _FakeFrameSource(frames)
    w = TrajectoryWriter(src, tmp_path, meta, max_frames=T)
    w.start()

    def produce():
        for _ in range(7):
            src.push()

    t = threading.Thread(target=produce)
    t.start()
    t.join(10)
    assert not t.is_alive(), "producer blocked: the writer does not release ring slots"
    w.finish(5)
    w.join_checked(10)
    assert src.released >= set(range(7))
    start = 0
    for b, (n, name) in enumerate(zip(natoms, meta["names"])):
        z = np.load(tmp_path / f"{name}.npz")
        assert np.array_equal(z["positions"], frames[:5, start:start + n])
        assert np.array_equal(z["numbers"], meta["numbers"][start:start + n]) and np.array_equal(z["cell"], meta["cell"][b])
        start += n
    idx = json.loads((tmp_path / "batch_a_1.json").read_text())
    assert idx["frames"] == 5 and idx["sids"] == meta["names"] and idx["atom_offsets"] == [0, 5, 8, 12]
    assert np.array_equal(np.load(tmp_path / idx["frames_file"]), frames[:5])
    assert not list(tmp_path.glob("*_tmp"))

    class B:
        sid = meta["names"]

    assert check_traj_files(B, tmp_path)
    # final frame only (save_full_traj=False): one frame kept
    w2 = TrajectoryWriter(_FakeFrameSource(frames[:1]), tmp_path / "last", meta, max_frames=1)
    w2.source.push()
    w2.start()
    w2.finish(1)
    w2.join_checked(10)
    assert np.load(tmp_path / "last" / "b_2.npz")["positions"].shape == (1, 3, 3)


def test_trajectory_writer_abort_publishes_nothing_and_writer_death_wakes_the_producer(tmp_path):
    """(1) a failed sampling attempt aborts the writer - no <sid>.npz, no batch file, no temporary file remains, so
    the resume rule (check_traj_files) does not count the batch as done; (2) a writer that dies (here: its directory cannot
    be created) aborts its frame source, so the producer's push returns instead of waiting for ring slots forever."""
    import threading

    from adsorbdiff_amd.trainer import check_traj_files
    from adsorbdiff_amd.trajectory import TrajectoryWriter

    rng = np.random.default_rng(2)
    natoms = [4, 2]
    N = sum(natoms)
    frames = rng.normal(size=(6, N, 3)).astype(np.float32)
    meta = dict(numbers=rng.integers(1, 80, N), tags=rng.integers(0, 3, N), fixed=rng.integers(0, 2, N),
                cell=rng.normal(size=(2, 3, 3)).astype(np.float32), natoms=np.array(natoms), names=["x_1", "y_2"])

    class AbortableSource(_FakeFrameSource):
        aborted = False

        def abort(self):
            with self.cv:
                self.aborted = True
                self.cv.notify_all()

        def push(self):
            with self.cv:
                self.cv.wait_for(lambda: self.aborted or self.avail < self.slots or (self.avail - self.slots) in self.released,
                                 timeout=10)
                if self.aborted:
                    raise RuntimeError("ring aborted")
                self.avail += 1
                self.cv.notify_all()

    # (1) abort after three frames were streamed
    src = AbortableSource(frames)
    w = TrajectoryWriter(src, tmp_path / "a", meta, max_frames=6)
    w.start()
    for _ in range(3):
        src.push()
    w.abort()
    w.join_checked(10)
    assert not list((tmp_path / "a").glob("*")), list((tmp_path / "a").glob("*"))

    class B:
        sid = meta["names"]

    assert not check_traj_files(B, tmp_path / "a")
    assert src.aborted
    # a finished run with zero counted frames publishes nothing either
    src0 = AbortableSource(frames)
    w0 = TrajectoryWriter(src0, tmp_path / "z", meta, max_frames=6)
    w0.start()
    w0.finish(0)
    w0.join_checked(10)
    assert not list((tmp_path / "z").glob("*.npz"))

    # (2) the writer dies at once (its "directory" is a file): the producer must not hang on the 3-slot ring
    blocker = tmp_path / "file"
    blocker.write_text("not a directory")
    src2 = AbortableSource(frames)
    w2 = TrajectoryWriter(src2, blocker / "sub", meta, max_frames=6)
    w2.start()
    err = []

    def produce():
        try:
            for _ in range(6):
                src2.push()
        except RuntimeError as e:
            err.append(e)

    t = threading.Thread(target=produce)
    t.start()
    t.join(10)
    assert not t.is_alive(), "producer still blocked after the writer died"
    assert err and src2.aborted
    with pytest.raises(OSError):
        w2.join_checked(10)


def test_site_messages_pack_and_merge_for_eight_ragged_ranks():
    """sampler.pack_sites / merge_packed_sites (what gather_sites sends and what it makes of the all-gather's result), without
    any process group: eight ranks with different system counts and adsorbate sizes, ids dealt out of order, NaN and
    denormal-looking bit patterns in the sites - the merged array is the global-order array, bit for bit; padding rows and
    padding atoms never leak; a shard larger than the bounds is refused."""
    from adsorbdiff_amd.sampler import merge_packed_sites, pack_sites

    g = torch.Generator().manual_seed(5)
    B, Amax = 37, 5
    natoms = torch.randint(1, Amax + 1, (B,), generator=g)
    full = torch.full((B, Amax, 3), float("nan"))
    for i in range(B):
        full[i, : natoms[i]] = torch.randn(int(natoms[i]), 3, generator=g) * 10.0 ** float(torch.randint(-30, 20, (1,), generator=g))
    full[3, 0, 0] = torch.tensor([1e-42])[0]          # a denormal survives the int32 transport
    perm = torch.randperm(B, generator=g).tolist()
    cuts = [0, 4, 4, 11, 15, 23, 24, 31, 37]           # rank 1 is empty, the others ragged
    Bmax = max(b - a for a, b in zip(cuts[:-1], cuts[1:]))
    msgs = []
    for r in range(8):
        ids = perm[cuts[r]:cuts[r + 1]]
        local = full[ids][:, : max([int(natoms[i]) for i in ids] + [1])] if ids else torch.empty(0, 1, 3)
        msgs.append(pack_sites(local, ids, (Bmax, Amax)))
        assert msgs[-1].dtype == torch.int32 and msgs[-1].shape == (Bmax, 1 + 3 * Amax)
    merged = merge_packed_sites(torch.stack(msgs), Amax, ordered=True)
    assert merged.shape == full.shape
    assert torch.equal(merged.view(torch.int32)[~torch.isnan(full)], full.view(torch.int32)[~torch.isnan(full)])
    assert torch.equal(torch.isnan(merged), torch.isnan(full))
    rank_major = merge_packed_sites(torch.stack(msgs), Amax, ordered=False)
    assert torch.equal(torch.nan_to_num(rank_major), torch.nan_to_num(full[perm]))
    with pytest.raises(ValueError):
        pack_sites(full[:Bmax + 1], list(range(Bmax + 1)), (Bmax, Amax))


def test_npz_to_ase_traj_converter(tmp_path):
    """The sink's format is .npz (ase is not installable in the build image); where ase IS importable the converter must
    reproduce the reference's <sid>.traj content (relaxation/ase_utils.py:19-48)."""
    ase_io = pytest.importorskip("ase.io")
    from adsorbdiff_amd.trajectory import npz_to_ase_traj

    rng = np.random.default_rng(1)
    pos = rng.normal(size=(3, 6, 3)).astype(np.float32)
    np.savez(tmp_path / "s_0.npz", positions=pos, numbers=np.array([78, 78, 78, 6, 8, 1]), tags=np.array([0, 1, 1, 2, 2, 2]),
             fixed=np.array([1, 0, 0, 0, 0, 0]), cell=np.diag([10.0, 11.0, 30.0]).astype(np.float32))
    out = npz_to_ase_traj(tmp_path / "s_0.npz")
    frames = ase_io.read(str(out), index=":")
    assert len(frames) == 3
    np.testing.assert_allclose(frames[-1].get_positions(), pos[-1], atol=1e-6)
    assert list(frames[0].get_tags()) == [0, 1, 1, 2, 2, 2] and list(frames[0].numbers) == [78, 78, 78, 6, 8, 1]
    assert list(frames[0].constraints[0].get_indices()) == [0]


def test_igso3_batched_sampling_consumes_the_numpy_stream_like_the_per_system_loop():
    """Igso3Tables.sample_and_score_vecs (what tr_so3_schedule calls) against the reference's per-system calls
    (so3_utils.py sample_vec / score_vec, trainers/sde_denoising_trainer.py tr_so3_schedule): same draws from the global
    numpy stream in the same order, same vectors bit for bit, same scores to 1e-12, same generator state afterwards."""
    import numpy as np

    from adsorbdiff_amd.so3_tables import Igso3Tables

    t = Igso3Tables.shared()
    rng = np.random.RandomState(7)
    eps = np.concatenate([10 ** rng.uniform(np.log10(0.01), np.log10(1.55), size=61), [0.01, 1.55, 1e-3, 3.0]])
    np.random.seed(11)
    want_v, want_s = np.empty((eps.size, 3)), np.empty((eps.size, 3))
    for b, e in enumerate(eps):
        want_v[b] = t.sample_vec(eps=float(e))
        want_s[b] = t.score_vec(vec=want_v[b], eps=float(e))
    after_loop = np.random.rand()
    np.random.seed(11)
    got_v, got_s = t.sample_and_score_vecs(eps)
    after_batch = np.random.rand()
    assert np.array_equal(got_v, want_v)
    np.testing.assert_allclose(got_s, want_s, rtol=1e-12, atol=0)
    assert after_loop == after_batch
