"""Training step on the GPU against the REFERENCE's autograd: tests/golden/train_small.npz holds a
batch noised by the reference's tr_so3_schedule, the reference model's loss (_compute_loss) and the gradients
torch.autograd gave for its parameters (oracle/make_golden.py section 6).  Tolerance 1e-4 relative (north_star)."""
import numpy as np
import pytest
import torch

from adsorbdiff_amd.painn_denoising import PaiNN
from adsorbdiff_amd.so3_tables import Igso3Tables
from adsorbdiff_amd.train_step import FusedAdamW, PaiNNTrainStep
from tests.helpers import batch_from_fixture, load_npz, rel_err, state_dict_from_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup():
    fx = load_npz("train_small.npz")
    tb = load_npz("igso3_tables.npz")
    m = PaiNN(None, 50, 1, hidden_channels=128, num_layers=2, num_rbf=128, cutoff=6.0, max_neighbors=20,
              scale_file={"upd_out_scalar_scale_0": 1.05, "upd_out_scalar_scale_1": 0.9}, so3_denoising=True)
    missing, unexpected = m.load_state_dict(state_dict_from_fixture(fx), strict=False)
    assert set(missing) <= {"atom_radii"} and not unexpected
    m = m.to(DEV)
    b = batch_from_fixture(fx, pos_key="pos_noised", device=DEV)
    targets = {k: torch.from_numpy(fx[k]) for k in ("tr_sigma", "rot_sigma", "tr_score", "rot_score")}
    tables = Igso3Tables(tb["omegas"], None, None, tb["exp_score_norm"])
    return fx, m, b, targets, tables


def test_loss_and_gradients_vs_reference_autograd():
    fx, m, b, targets, tables = _setup()
    step = PaiNNTrainStep(m, DEV, igso3=tables)
    step.zero_grad()
    loss = step.loss_and_grad(b, targets).cpu()
    # forward outputs and the loss
    assert rel_err(step.last_outputs[0].cpu(), fx["out1"]) < 1e-5 and rel_err(step.last_outputs[1].cpu(), fx["out2"]) < 1e-5
    assert abs(float(loss[0]) - float(fx["loss"])) < 1e-5 * abs(float(fx["loss"]))
    np.testing.assert_allclose(loss[1:].numpy(), fx["loss_terms"], rtol=1e-5)
    # every parameter's gradient norm, and the full gradient of the stored ones
    P = dict(m.named_parameters())
    for name, gn in zip(fx["grad_names"], fx["grad_norms"]):
        name = str(name)
        p = P[name]
        if not p.requires_grad:
            continue
        if gn == 0.0:  # out_energy.*: no path to the loss; .grad stays None as under the reference's autograd
            assert p.grad is None or float(p.grad.norm()) == 0.0, name
            continue
        got = float(p.grad.norm())
        assert abs(got - gn) < 1e-4 * gn, (name, got, gn)
    for key in fx:
        if key.startswith("grad::"):
            name = key[6:]
            assert rel_err(P[name].grad.cpu(), fx[key]) < 1e-4, (name, rel_err(P[name].grad.cpu(), fx[key]))


@pytest.mark.parametrize("fixture", ["train_full.npz", "train_full_trained_like.npz"])
def test_config5_width_loss_and_gradients_vs_reference_autograd(fixture):
    """H = 512, 6 layers, 12 A / 50 neighbours (the shipped PaiNN config) on 2 x 200-atom systems: tests/golden/
    train_full.npz (oracle/make_golden.py section 10) holds the reference model's loss and, for each of its 114
    parameters, the norm and a strided 256-element sample of torch.autograd's gradient.  The weights are rebuilt from the
    seeds (the generator asserts that this mirror reproduces the reference's bit for bit).  train_full.npz sits at the bare
    initialisers (loss 1.3e6, gradient norms 1e3 ... 3e8: a point the reference's loop would abort on);
    train_full_trained_like.npz (round 5) is the same step with the weights rescaled by a rule of their names to trained-like
    magnitudes (LayerNorm gain x 0.05, radial-direction rows x 1e-2, heads x 2e-4): loss 2.19, gradient norms 1e-12 ... 5e-5."""
    from adsorbdiff_amd.scaling import PAINN_NB6_SCALE_FACTORS

    fx = load_npz(fixture)
    tb = load_npz("igso3_tables.npz")
    torch.manual_seed(int(fx["weight_seed"]))
    m = PaiNN(None, 50, 1, cutoff=float(fx["cutoff"]), max_neighbors=int(fx["max_neighbors"]),
              scale_file=PAINN_NB6_SCALE_FACTORS, so3_denoising=True)
    g = torch.Generator().manual_seed(int(fx["bias_seed"]))
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if n_.endswith("bias") or "layernorm" in n_:
                p_.add_(0.1 * torch.randn(p_.shape, generator=g))
        if int(fx.get("trained_like", 0)):   # oracle/make_golden.py::trained_like_rescale_ (a rule of parameter names)
            H = 512
            for n_, p_ in m.named_parameters():
                if "x_layernorm" in n_:
                    p_.mul_(0.05)
                if n_.endswith("x_proj.2.weight") or n_.endswith("x_proj.2.bias"):
                    p_[2 * H:].mul_(1e-2)
                if n_.endswith("xvec_proj.2.weight") or n_.endswith("xvec_proj.2.bias"):
                    p_.mul_(0.3)
                if n_.endswith("output_network.1.vec2_proj.weight"):
                    p_.mul_(2e-4)
    m = m.to(DEV)
    b = batch_from_fixture(fx, pos_key="pos_noised", device=DEV)
    targets = {k: torch.from_numpy(fx[k]) for k in ("tr_sigma", "rot_sigma", "tr_score", "rot_score")}
    step = PaiNNTrainStep(m, DEV, igso3=Igso3Tables(tb["omegas"], None, None, tb["exp_score_norm"]))
    step.zero_grad()
    loss = step.loss_and_grad(b, targets).cpu()
    assert rel_err(step.last_outputs[0].cpu(), fx["out1"]) < 1e-5 and rel_err(step.last_outputs[1].cpu(), fx["out2"]) < 1e-5
    assert abs(float(loss[0]) - float(fx["loss"])) < 1e-5 * abs(float(fx["loss"]))
    P = dict(m.named_parameters())
    assert list(P) == [str(n) for n in fx["grad_names"]]
    worst, worst_name, worst_rbf = 0.0, "", 0.0
    for name, gn in zip(fx["grad_names"], fx["grad_norms"]):
        name = str(name)
        got = P[name].grad
        if gn == 0.0:
            assert got is None or float(got.norm()) == 0.0, name
            continue
        assert abs(float(got.double().norm()) - gn) < 1e-4 * gn, (name, float(got.norm()), gn)
        idx = torch.from_numpy(fx["gidx::" + name])
        ref = torch.from_numpy(fx["gval::" + name]).double()
        e = float((got.reshape(-1).cpu()[idx].double() - ref).norm() / ref.norm())
        if e > worst:
            worst, worst_name = e, name
        if "rbf_proj.weight" in name:
            worst_rbf = max(worst_rbf, e)
        assert e < 1e-4, (name, e)
    print(f"config-5 width ({fixture}): worst sampled gradient error {worst:.2e} ({worst_name}); rbf_proj weights {worst_rbf:.2e}")


def test_igso3_tables_computed_on_the_device_are_finite_and_pinned():
    """compute_tables on the ROCm device (what a fresh checkout does on first use): every row of the grid finite
    (in the tail of a narrow distribution the series is cancellation noise and a device reduction order can land on
    exactly 0), the expected score norms equal to the reference's table at 1e-7 on all 1000 rows, CDF and score equal
    on the golden rows / columns where an angle carries probability mass."""
    from adsorbdiff_amd.so3_tables import compute_tables

    z = load_npz("igso3_tables.npz")
    got = compute_tables(torch.device(DEV))
    for k in ("cdf", "score", "exp_score_norm"):
        assert np.isfinite(got[k]).all(), k
    np.testing.assert_allclose(got["exp_score_norm"], z["exp_score_norm"], rtol=1e-7)
    rows, cols = z["eps_rows"], z["om_cols"]
    np.testing.assert_allclose(got["cdf"][rows][:, cols], z["cdf"], rtol=1e-9, atol=1e-12)
    pdf = np.diff(np.concatenate([np.zeros((len(rows), 1)), got["cdf"][rows]], 1), axis=1)[:, cols]
    live = pdf > 1e-9 * pdf.max(axis=1, keepdims=True)
    np.testing.assert_allclose(got["score"][rows][:, cols][live], z["score"][live], rtol=1e-6)


def test_gradients_are_reproducible_and_accumulate():
    fx, m, b, targets, tables = _setup()
    step = PaiNNTrainStep(m, DEV, igso3=tables)
    step.zero_grad()
    step.loss_and_grad(b, targets)
    g1 = {k: p.grad.clone() for k, p in m.named_parameters() if p.requires_grad and p.grad is not None}
    step.zero_grad()
    step.loss_and_grad(b, targets)
    for k, p in m.named_parameters():
        if k in g1 and k != "atom_emb.embeddings.weight":  # the embedding gradient uses float atomics
            assert torch.equal(p.grad, g1[k]), k
    step.loss_and_grad(b, targets)  # no zero_grad: gradients add up
    k = "message_layers.1.rbf_proj.weight"
    assert rel_err(dict(m.named_parameters())[k].grad, 2 * g1[k]) < 1e-6


def test_fused_message_backward_agrees_with_the_rbfh_reading_backward(monkeypatch):
    """message_bwd.hip regenerates rbfh on the matrix cores (f16x3) and writes drbfh in its lane order; the step permutes
    the rbf_proj gradient back.  Every gradient agrees with the backward that reads a materialised rbfh (exact f32)."""
    fx, m, b, targets, tables = _setup()
    grads = {}
    for mode in ("plain", "fused"):
        monkeypatch.setenv("ADF_TRAIN_MSG_BWD", mode)
        step = PaiNNTrainStep(m, DEV, igso3=tables)
        assert step.fused_message_backward == (mode == "fused")
        step.zero_grad()
        step.loss_and_grad(b, targets)
        grads[mode] = {k: p.grad.clone() for k, p in m.named_parameters() if p.requires_grad and p.grad is not None}
    assert set(grads["plain"]) == set(grads["fused"])
    for k, g in grads["plain"].items():
        if float(g.norm()) == 0.0:
            assert float(grads["fused"][k].norm()) == 0.0, k
            continue
        assert rel_err(grads["fused"][k], g) < 2e-5, (k, rel_err(grads["fused"][k], g))


def test_rbf_proj_gradient_without_drbfh_in_memory_equals_the_materialised_path(monkeypatch):
    """csrc/rbf_wgrad.hip forms d(rbfh) again while it stages the weight-gradient product (the backward then stores nothing per
    edge); ADF_TRAIN_RBF_WGRAD=materialised is the round-4 path (d(rbfh) [E, 3H] written and read back).  The staged values
    are computed by the same expressions, the product is the same three-term bf16 split: rbf_proj's weight gradient agrees to
    fp32 summation order (the edge ranges of the partial sums differ), its bias gradient likewise; the other gradients come from
    the same arithmetic (bit-identical except where the step sums with atomics, e.g. the embedding rows)."""
    fx, m, b, targets, tables = _setup()
    grads = {}
    for mode in ("materialised", "fused"):
        monkeypatch.setenv("ADF_TRAIN_RBF_WGRAD", mode)
        step = PaiNNTrainStep(m, DEV, igso3=tables)
        assert step.fused_message_backward and step.rbf_wgrad_fused == (mode == "fused")
        step.zero_grad()
        step.loss_and_grad(b, targets)
        grads[mode] = {k: p.grad.clone() for k, p in m.named_parameters() if p.requires_grad and p.grad is not None}
    assert set(grads["materialised"]) == set(grads["fused"])
    worst, same = 0.0, 0
    for k, g in grads["materialised"].items():
        if "rbf_proj" in k:
            assert float(g.norm()) > 0.0, k
            e = rel_err(grads["fused"][k], g)
            worst = max(worst, e)
            assert e < 2e-6, (k, e)
        else:
            same += int(torch.equal(grads["fused"][k], g))
            assert rel_err(grads["fused"][k], g) < 1e-6, (k, rel_err(grads["fused"][k], g))
    print(f"rbf_proj gradients, fused vs materialised: worst relative difference {worst:.2e}; "
          f"{same} of {len(grads['fused']) - 4} other gradients bit-identical")


def test_fused_adamw_matches_torch_adamw_clip_ema():
    """AdamW + clip_grad_norm_ + EMA in one kernel per tensor == torch.optim.AdamW, torch clip and the EMA mirror."""
    from adsorbdiff_amd.exponential_moving_average import ExponentialMovingAverage

    fx, m, b, targets, tables = _setup()
    step = PaiNNTrainStep(m, DEV, igso3=tables)
    ref = PaiNN(None, 50, 1, hidden_channels=128, num_layers=2, num_rbf=128, cutoff=6.0, max_neighbors=20,
                scale_file={"upd_out_scalar_scale_0": 1.05, "upd_out_scalar_scale_1": 0.9}, so3_denoising=True).to(DEV)
    ref.load_state_dict(m.state_dict())
    no_decay = set(ref.no_weight_decay())
    groups = [{"params": [p for n, p in ref.named_parameters() if p.requires_grad and n in no_decay], "weight_decay": 0.0},
              {"params": [p for n, p in ref.named_parameters() if p.requires_grad and n not in no_decay], "weight_decay": 0.01}]
    topt = torch.optim.AdamW(groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    ema_ref = ExponentialMovingAverage(ref.parameters(), 0.99)
    ema = ExponentialMovingAverage(m.parameters(), 0.99)
    opt = FusedAdamW(m, lr=1e-3, weight_decay=0.01, max_grad_norm=0.05, ema=ema)
    for it in range(3):
        step.zero_grad()
        step.loss_and_grad(b, targets)
        for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
            q.grad = p.grad.clone() if (p.requires_grad and p.grad is not None) else None
        gn_ref = torch.nn.utils.clip_grad_norm_([q for q in ref.parameters() if q.grad is not None], max_norm=0.05)
        topt.step()
        ema_ref.update()
        gn = opt.step()
        assert abs(float(gn) - float(gn_ref)) < 1e-5 * float(gn_ref)
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert rel_err(p.detach(), q.detach()) < 2e-6, n
    for s1, s2 in zip(ema.shadow_params, ema_ref.shadow_params):
        assert rel_err(s1, s2) < 2e-6


WORKER = r"""
import os, sys, torch, numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from adsorbdiff_amd.data import Batch
from adsorbdiff_amd.painn_denoising import PaiNN
from adsorbdiff_amd.so3_tables import Igso3Tables
from adsorbdiff_amd.trainer import DenoisingTrainer
from tests.helpers import batch_from_fixture, load_npz, state_dict_from_fixture

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
fx, tb = load_npz("train_small.npz"), load_npz("igso3_tables.npz")
m = PaiNN(None, 50, 1, hidden_channels=128, num_layers=2, num_rbf=128, cutoff=6.0, max_neighbors=20,
          scale_file={"upd_out_scalar_scale_0": 1.05, "upd_out_scalar_scale_1": 0.9}, so3_denoising=True)
m.load_state_dict(state_dict_from_fixture(fx), strict=False)
tr = DenoisingTrainer(m, device="cuda:0")
tr.setup_training(dict(ads_std_low=0.1, ads_std_high=10, rot_std_low=0.01, rot_std_high=1.55), lr=0.0, weight_decay=0.0,
                  clip_grad_norm=0.0, ema_decay=0.0, tables=Igso3Tables(tb["omegas"], None, None, tb["exp_score_norm"]))
full = batch_from_fixture(fx, pos_key="pos_noised")
data = full.to_data_list()
mine = Batch.from_data_list(data[2 * rank : 2 * rank + 2])          # two of the four systems per rank
for k in ("tr_sigma", "rot_sigma", "tr_score", "rot_score"):
    setattr(mine, k, torch.from_numpy(fx[k])[2 * rank : 2 * rank + 2])
out = tr.train_step(mine, noised=True)                              # lr = 0: only the averaged gradients matter
if rank == 0:
    torch.save({k: p.grad.cpu() for k, p in m.named_parameters() if p.requires_grad and p.grad is not None}, sys.argv[2])
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rank_training_step_equals_full_batch(tmp_path):
    """DDP semantics: two ranks (sharing cuda:0, gloo) with half of the fixture batch each; after the bucketed
    all-reduce every rank holds the gradient of the full-batch loss = the reference's autograd gradients."""
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), str(root), str(tmp_path / "g.pt")], env=env))
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    got = torch.load(tmp_path / "g.pt")
    fx = load_npz("train_small.npz")
    for key in fx:
        if key.startswith("grad::"):
            assert rel_err(got[key[6:]], fx[key]) < 1e-4, key


_WGRAD_F32_SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r})
from adsorbdiff_amd.train_step import _Ops
ops = _Ops("cuda:0")
torch.manual_seed(11)
worst = 0.0
for (M, N, K) in ((4096 + 17, 128, 128), (3000, 256, 128), (33, 128, 256)):
    A = torch.randn(M, K, device="cuda:0"); W = torch.randn(N, K, device="cuda:0") * 0.05
    dC = torch.randn(M, N, device="cuda:0")
    dW = torch.zeros(N, K, device="cuda:0"); db = torch.zeros(N, device="cuda:0")
    dA = ops.linear_bwd(A, W, dC, M, N, K, dW, db)
    torch.cuda.synchronize()
    rW = dC.double().t() @ A.double(); rb = dC.double().sum(0); rA = dC.double() @ W.double()
    for got, ref in ((dW, rW), (db, rb), (dA, rA)):
        worst = max(worst, float((got.double() - ref).norm() / ref.norm()))
print("WORST", worst)
"""


@pytest.mark.parametrize("mode", ["f32", "default"])
def test_linear_backward_weight_gradient_kernels_vs_torch(mode):
    """the exact-f32 weight-gradient kernel (ADF_WGRAD=f32, tr_wgrad128_kernel) staged only the first
    32-row chunk of every split.  ADF_WGRAD is read once per process, hence the child process; M >> 32 and ragged."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = str(Path(__file__).resolve().parent.parent)
    env = dict(os.environ)
    env.pop("ADF_WGRAD", None)
    if mode == "f32":
        env["ADF_WGRAD"] = "f32"
    res = subprocess.run([sys.executable, "-c", _WGRAD_F32_SCRIPT.format(root=root)], env=env, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    worst = float(res.stdout.strip().split("WORST")[-1])
    assert worst < 1e-5, worst
