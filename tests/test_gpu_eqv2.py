"""GPU parity tests of the HIP EquiformerV2 denoiser (BASELINE config 4) through the C ABI.

PARITY STATUS: the fixtures eqv2_l*m*.npz are outputs of the REAL reference model run on CPU, but under the e3nn stand-in
of oracle/refshim (e3nn 0.4.4 is not installable here), so the S2-grid normalisation is unpinned.
Tolerance: 1e-4 relative (BASELINE.json north_star), written next to each assertion."""
import numpy as np
import pytest
import torch

from adsorbdiff_amd.equiformer_v2_denoising import EquiformerV2S_OC20_DenoisingPos
from tests.helpers import batch_from_fixture, load_npz, rel_err, state_dict_from_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REL_TOL = 1e-4


def parse_hp(fx):
    out = {}
    for tok in str(fx["hp"]).split():
        k, v = tok.split("=")
        if v in ("True", "False"):
            out[k] = v == "True"
        else:
            try:
                out[k] = int(v)
            except ValueError:
                try:
                    out[k] = float(v)
                except ValueError:
                    out[k] = v
    return out


def model_from_fixture(fx):
    hp = parse_hp(fx)
    m = EquiformerV2S_OC20_DenoisingPos(
        None, None, None, lmax_list=[int(fx["lmax"])], mmax_list=[int(fx["mmax"])], use_s2_act_attn=False,
        use_attn_renorm=True, use_gate_act=False, alpha_drop=0.0, drop_path_rate=0.0, proj_drop=0.0,
        weight_init="uniform", **hp)
    m.load_state_dict(state_dict_from_fixture(fx))
    return m.to(DEV).eval()


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("name", ["eqv2_l4m2.npz", "eqv2_l6m2.npz", "eqv2_l6m2_w32.npz"])
def test_eqv2_forward_vs_reference_fixture(name, exact):
    """(f1, f2) and the node embeddings after the edge-degree embedding and after every block, against the reference
    model's own recordings (its edge list: exact K-th-neighbour ties in this small cell), 1e-4.  `w32`: every
    contraction length is a multiple of 32, the shapes the f16x3 matrix-core kernels take; `exact`: f32 arithmetic."""
    fx = load_npz(name)
    m = model_from_fixture(fx)
    b = batch_from_fixture(fx, device=DEV)
    eng = m.engine()
    eng.set_arithmetic(exact)
    eng.set_edges(torch.from_numpy(fx["edge_index"]), torch.from_numpy(fx["edge_vec"]))
    f1, f2, xb = eng.forward(b, return_blocks=True)
    e1, e2 = rel_err(f1.cpu(), fx["f1"]), rel_err(f2.cpu(), fx["f2"])
    print(name, "exact" if exact else "f16x3", "rel err", e1, e2)
    assert e1 < REL_TOL and e2 < REL_TOL
    L = int(fx["lmax"])
    for k in range(xb.shape[0]):
        for l in range(L + 1):
            got, ref = xb[k, :, l * l:(l + 1) ** 2].cpu(), torch.from_numpy(fx["x_blocks"][k, :, l * l:(l + 1) ** 2])
            assert rel_err(got, ref) < REL_TOL, (k, l, rel_err(got, ref))
    # per-atom bound: every row within 1e-4 of the largest row norm
    for got, ref in ((f1.cpu().numpy(), fx["f1"]), (f2.cpu().numpy(), fx["f2"])):
        scale = np.linalg.norm(ref, axis=1).max()
        assert np.abs(got - ref).max() < REL_TOL * scale


def make_model(lmax, mmax, C, hidden, heads, alpha, value, ffn, ec, layers, cutoff, K=20, seed=0):
    torch.manual_seed(seed)
    m = EquiformerV2S_OC20_DenoisingPos(
        None, None, None, max_neighbors=K, max_radius=cutoff, max_num_elements=90, num_layers=layers, sphere_channels=C,
        attn_hidden_channels=hidden, num_heads=heads, attn_alpha_channels=alpha, attn_value_channels=value,
        ffn_hidden_channels=ffn, norm_type="layer_norm_sh", lmax_list=[lmax], mmax_list=[mmax], grid_resolution=18,
        edge_channels=ec, attn_activation="silu", ffn_activation="silu", use_grid_mlp=True, use_sep_s2_act=True,
        alpha_drop=0.0, drop_path_rate=0.0, weight_init="uniform", FOR_denoising=True)
    # trained-like magnitudes for the edge embeddings (the reference initialises them at 1e-3) so that the radial
    # path carries signal in the comparison
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("source_embedding.weight") or n.endswith("target_embedding.weight"):
                p.mul_(300.0)
    return m.eval()


def oracle_hp(m):
    return dict(lmax=m.lmax_list[0], mmax=m.mmax_list[0], num_layers=m.num_layers, sphere_channels=m.sphere_channels,
                attn_hidden_channels=m.attn_hidden_channels, num_heads=m.num_heads,
                attn_alpha_channels=m.attn_alpha_channels, attn_value_channels=m.attn_value_channels,
                ffn_hidden_channels=m.ffn_hidden_channels, grid_resolution=m.grid_resolution, max_radius=m.max_radius,
                max_neighbors=m.max_neighbors)


def safe_batch(n_sys, n_slab, seed):
    from adsorbdiff_amd.synthetic import make_batch

    b = make_batch(n_sys, n_slab=n_slab, n_ads=4, seed=seed)
    z = b.atomic_numbers.clone()
    z[(z == 36) | (z == 54)] = 47.0  # elements without a tabulated radius give NaN in the reference
    b.atomic_numbers = z
    return b


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("lmax,hidden", [(4, 32), (6, 32), (6, 64)])
def test_eqv2_matrix_core_path_vs_oracle(lmax, hidden, exact):
    """Shapes the f16x3 matrix-core products take (every K a multiple of 32), against the CPU oracle on the oracle's
    own edge list; also the exact-f32 arithmetic on the same shapes.  1e-4 on the outputs and on every block.  64 hidden
    channels (the benchmark's value) = two channel blocks, i.e. two waves of the S2 activation, per edge."""
    from oracle import eqv2_oracle as Q

    m = make_model(lmax, 2, C=32, hidden=hidden, heads=2, alpha=16, value=16, ffn=32, ec=32, layers=2, cutoff=12.0)
    b = safe_batch(2, 36, seed=7)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ei, sh, nb = Q.radius_graph_pbc(b.pos, b.cell, b.natoms, 12.0, 20)
    ei, d, v, _ = Q.pbc_distances(b.pos, ei, b.cell, sh, nb)
    with torch.no_grad():
        r1, r2 = Q.eqv2_forward(sd, oracle_hp(m), b.pos, b.atomic_numbers, b.cell, b.natoms, graph=(ei, v))
    m = m.to(DEV)
    eng = m.engine()
    eng.set_arithmetic(exact)
    eng.set_edges(ei, v)
    f1, f2 = m(b.to(DEV))
    e1, e2 = rel_err(f1.cpu(), r1), rel_err(f2.cpu(), r2)
    print(f"L={lmax} hidden={hidden} exact={exact}: rel err {e1:.2e} {e2:.2e}")
    assert e1 < REL_TOL and e2 < REL_TOL
    for got, ref in ((f1.cpu(), r1), (f2.cpu(), r2)):
        assert float((got - ref).abs().max()) < REL_TOL * float(ref.norm(dim=1).max())


def test_eqv2_own_graph_vs_oracle():
    """The device-built graph (strict top-K, not symmetrised; models/base.py:33-123) on 200-atom systems whose cells
    are larger than the cutoff in-plane (no exactly tied self-images): same edges as the oracle's builder, and the
    forward on it equals the oracle's forward."""
    from oracle import eqv2_oracle as Q

    m = make_model(4, 2, C=8, hidden=8, heads=2, alpha=4, value=4, ffn=16, ec=8, layers=1, cutoff=12.0)
    b = safe_batch(2, 196, seed=11)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        r1, r2 = Q.eqv2_forward(sd, oracle_hp(m), b.pos, b.atomic_numbers, b.cell, b.natoms)
    m = m.to(DEV)
    f1, f2 = m(b.to(DEV))
    e1, e2 = rel_err(f1.cpu(), r1), rel_err(f2.cpu(), r2)
    print(f"own graph: rel err {e1:.2e} {e2:.2e}")
    assert e1 < REL_TOL and e2 < REL_TOL


def test_eqv2_sampling_vs_oracle_stepper():
    """Denoiser.run with the EquiformerV2 score model (HIP forward + the shared HIP stepper) against the oracle loop
    (oracle EquiformerV2 forward + oracle reverse step) on 2 x 200-atom systems, 3 ODE steps: positions at 1e-4 A."""
    from adsorbdiff_amd.denoising_torch import Denoiser, DiffTorchCalc
    from adsorbdiff_amd.trainer import DenoisingTrainer
    from oracle import eqv2_oracle as Q
    from oracle import painn_oracle as O

    m = make_model(4, 2, C=8, hidden=8, heads=2, alpha=4, value=4, ffn=16, ec=8, layers=1, cutoff=12.0)
    m.so3_denoising = True
    b = safe_batch(2, 196, seed=21)
    params = dict(num_steps=3, ads_std_low=0.1, ads_std_high=10, rot_std_low=0.01, rot_std_high=1.55, ode=True,
                  early_stop=False)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.manual_seed(5)
    noise = torch.rand(2, 3)
    pos = O.initial_placement(b.pos.clone(), b.cell, b.tags, b.batch, noise)
    with torch.no_grad():
        for t in range(3):
            f1, f2 = Q.eqv2_forward(sd, oracle_hp(m), pos, b.atomic_numbers, b.cell, b.natoms)
            pos, _, _, _ = O.reverse_step(pos, b.cell, b.tags, b.batch, f1, f2, b.fixed, t, params)
    trainer = DenoisingTrainer(m.to(DEV), device=DEV)
    den = Denoiser(b.clone().to(DEV), DiffTorchCalc(trainer), dict(params, placement_noise=noise), device=DEV)
    out = den.run()
    assert den.steps_applied == 3
    ads = (b.tags == 2)
    diff = float((out.pos.cpu()[ads] - pos[ads]).abs().max())
    moved = float((pos[ads] - b.pos[ads]).abs().max())
    print(f"eqv2 sampling: max |dpos| {diff:.2e} A (adsorbate moved up to {moved:.2f} A)")
    assert diff < 1e-4
    assert torch.equal(out.pos.cpu()[~ads], b.pos[~ads])


def test_eqv2_static_radial_tables_equal_per_edge_evaluation(monkeypatch):
    """With the reference's radii the distance basis vanishes on every edge and the radial MLPs are tabulated per element
    pair at weight binding; ADF_EQV2_RADIAL=edge evaluates them per edge instead: same outputs."""
    m = make_model(4, 2, C=32, hidden=32, heads=2, alpha=16, value=16, ffn=32, ec=32, layers=1, cutoff=12.0).to(DEV)
    b = safe_batch(2, 36, seed=9).to(DEV)
    f1, f2 = m(b)
    monkeypatch.setenv("ADF_EQV2_RADIAL", "edge")
    m._engine.close()
    m._engine = None
    g1, g2 = m(b)
    assert rel_err(f1.cpu(), g1.cpu()) < 2e-6 and rel_err(f2.cpu(), g2.cpu()) < 2e-6


def test_eqv2_folded_feed_forward_and_compact_force_blocks_equal_the_plain_evaluation(monkeypatch):
    """Two algebraic shortcuts of the default path: the grid MLP's first / last map folded into the SO(3) linears around
    it (one product on the grid rows instead of three) and the force blocks' second convolution restricted to the columns
    that reach l = 1.  ADF_EQV2_FOLD=0 / ADF_EQV2_COMPACT=0 evaluate everything as the reference does: same outputs and
    same node embeddings after every block."""
    m = make_model(6, 2, C=32, hidden=64, heads=2, alpha=16, value=16, ffn=32, ec=32, layers=2, cutoff=12.0).to(DEV)
    b = safe_batch(2, 36, seed=9).to(DEV)
    f1, f2, xb = m.engine().forward(b, return_blocks=True)
    monkeypatch.setenv("ADF_EQV2_FOLD", "0")
    monkeypatch.setenv("ADF_EQV2_COMPACT", "0")
    m._engine.close()
    m._engine = None
    g1, g2, yb = m.engine().forward(b, return_blocks=True)
    assert rel_err(f1.cpu(), g1.cpu()) < 2e-6 and rel_err(f2.cpu(), g2.cpu()) < 2e-6
    for k in range(xb.shape[0]):
        assert rel_err(xb[k].cpu(), yb[k].cpu()) < 2e-6, k


def test_eqv2_second_convolution_through_the_streamed_fragment_kernel_is_bit_identical(monkeypatch):
    """Edge-level plain products on whole 256-column tiles (the second SO(2) convolution's orders m >= 1 at config 4's head
    widths: N = 1536 / 1280, K = 384 / 320; so2_ops.py:158-238) run gemm16.hip's streamed-fragment kernel with the row
    magnitudes of this path.  ADF_EQV2_CONV2_WR=0 keeps them on `eq_gemm16_256_kernel`: both outputs and the node embeddings
    after every block equal bit for bit.  The batch has more than 32 768 edges (the route takes launches of >= 65 536 rows)."""
    m = make_model(6, 2, C=32, hidden=64, heads=8, alpha=16, value=16, ffn=32, ec=32, layers=2, cutoff=12.0).to(DEV)
    b = safe_batch(9, 196, seed=17).to(DEV)
    eng = m.engine()
    f1, f2, xb = eng.forward(b, return_blocks=True)
    assert int(eng.counters().num_edges) > 32768
    monkeypatch.setenv("ADF_EQV2_CONV2_WR", "0")
    m._engine.close()
    m._engine = None
    g1, g2, yb = m.engine().forward(b, return_blocks=True)
    assert torch.equal(f1, g1) and torch.equal(f2, g2) and torch.equal(xb, yb)


def test_eqv2_attention_logits_four_heads_per_wave_equal_the_one_head_kernel(monkeypatch):
    """transformer_block.py:312-345 (LayerNorm over the 64 alpha channels of a head, smooth leaky ReLU, dot with alpha_dot):
    the default kernel takes four heads per wave and sums inside 16-lane rows (`eq_alpha_logit64_kernel`); the generic kernel
    (ADF_EQV2_ALPHA_GENERIC=1, any channel count) one head at a time over 64 lanes.  Different summation orders: outputs and
    every block's node embeddings within 2e-6."""
    m = make_model(6, 2, C=32, hidden=64, heads=8, alpha=64, value=16, ffn=32, ec=32, layers=2, cutoff=12.0).to(DEV)
    b = safe_batch(2, 64, seed=23).to(DEV)
    f1, f2, xb = m.engine().forward(b, return_blocks=True)
    monkeypatch.setenv("ADF_EQV2_ALPHA_GENERIC", "1")
    m._engine.close()
    m._engine = None
    g1, g2, yb = m.engine().forward(b, return_blocks=True)
    assert rel_err(f1.cpu(), g1.cpu()) < 2e-6 and rel_err(f2.cpu(), g2.cpu()) < 2e-6
    for k in range(xb.shape[0]):
        assert rel_err(xb[k].cpu(), yb[k].cpu()) < 2e-6, k


def test_eqv2_subset_forward_rows_are_bit_identical():
    """adf_eqv2_forward_subset: the force blocks on the listed targets' incoming edges only (what the sampler needs: the
    adsorbate rows).  Listed rows equal the full forward's bit for bit, the other rows are not written."""
    m = make_model(6, 2, C=32, hidden=64, heads=2, alpha=16, value=16, ffn=32, ec=32, layers=2, cutoff=12.0).to(DEV)
    b = safe_batch(3, 36, seed=13).to(DEV)
    eng = m.engine()
    f1, f2 = eng.forward(b)
    prep = eng.prepare(b)
    idx = torch.nonzero(prep.tags == 2).reshape(-1).to(torch.int32).contiguous()
    assert 0 < idx.numel() < prep.num_atoms
    g1 = torch.full_like(f1, 7.0)
    g2 = torch.full_like(f2, 7.0)
    eng.forward_prepared(prep, b.pos.float().contiguous(), g1, g2, out_idx=idx)
    eng.check_flags()
    li = idx.long()
    assert torch.equal(g1[li], f1[li]) and torch.equal(g2[li], f2[li])
    rest = torch.ones(prep.num_atoms, dtype=torch.bool, device=DEV)
    rest[li] = False
    assert bool((g1[rest] == 7.0).all()) and bool((g2[rest] == 7.0).all())


def test_eqv2_incremental_blocks_are_bit_identical():
    """adf_eqv2_set_incremental: with the static-atom promise in force a forward recomputes, in block i, only the targets
    within i + 1 hops of a changed in-edge list.  Every block's output and both force outputs equal, bit for bit, a full
    forward at the same positions (an engine that keeps nothing); a forward at unchanged positions recomputes no row."""
    m = make_model(4, 2, C=32, hidden=32, heads=2, alpha=16, value=16, ffn=32, ec=32, layers=4, cutoff=12.0).to(DEV)
    b = safe_batch(2, 196, seed=31).to(DEV)
    N = int(b.pos.shape[0])
    S, C_ = 25, 32
    ads = (b.tags == 2)

    def drop_engine():
        if m._engine is not None:
            m._engine.close()
        m._engine = None

    def full_forward(pos):
        drop_engine()  # a fresh handle: nothing kept
        eng = m.engine()
        prep = eng.prepare(b)
        f1, f2 = torch.empty(N, 3, device=DEV), torch.empty(N, 3, device=DEV)
        xb = torch.empty(5, N, S, C_, device=DEV)
        eng.forward_prepared(prep, pos, f1, f2, x_blocks=xb)
        eng.check_flags()
        return f1, f2, xb

    pos0 = b.pos.float().contiguous()
    gen = torch.Generator().manual_seed(5)
    moves = [torch.zeros(N, 3, device=DEV) for _ in range(3)]
    moves[0][ads] = (0.3 * torch.randn(int(ads.sum()), 3, generator=gen)).to(DEV)
    moves[1][ads] = (0.05 * torch.randn(int(ads.sum()), 3, generator=gen)).to(DEV)
    positions = [pos0, pos0 + moves[0], pos0 + moves[0] + moves[1], pos0 + moves[0] + moves[1]]
    want = [full_forward(p) for p in positions]

    drop_engine()
    eng = m.engine()
    prep = eng.prepare(b)
    eng.set_moving_atoms(prep, ads)
    eng.set_incremental(True)
    rows = []
    for p, (w1, w2, wx) in zip(positions, want):
        f1, f2 = torch.empty(N, 3, device=DEV), torch.empty(N, 3, device=DEV)
        xb = torch.empty(5, N, S, C_, device=DEV)
        eng.forward_prepared(prep, p, f1, f2, x_blocks=xb)
        eng.check_flags()
        for i in range(5):
            assert torch.equal(xb[i], wx[i]), f"block {i}"
        assert torch.equal(f1, w1) and torch.equal(f2, w2)
        c = eng.counters()
        rows.append((int(c.inc_rows), int(c.inc_rows_full)))
    assert rows[0] == (4 * N, 4 * N)                      # nothing kept yet
    assert rows[1][1] == 8 * N and rows[1][0] < 8 * N - N // 4   # block 0 (and 1) recompute a part of the slab only
    assert rows[3][0] == rows[2][0]                       # unchanged positions: no row recomputed
    # the adsorbate-only subset forward uses the same kept state
    idx = torch.nonzero(ads).reshape(-1).to(torch.int32).contiguous()
    g1, g2 = torch.zeros(N, 3, device=DEV), torch.zeros(N, 3, device=DEV)
    p = positions[1]
    eng.forward_prepared(prep, p, g1, g2, out_idx=idx)
    eng.check_flags()
    assert torch.equal(g1[idx.long()], want[1][0][idx.long()]) and torch.equal(g2[idx.long()], want[1][1][idx.long()])
    # dropping the promise drops the kept state: the next forward is a full one
    eng.set_moving_atoms(None, None)
    eng.forward_prepared(prep, positions[0], g1, g2)
    eng.check_flags()
    assert torch.equal(g1, want[0][0])


def test_eqv2_sampling_with_and_without_incremental_blocks_gives_identical_sites():
    from adsorbdiff_amd.denoising_torch import Denoiser, DiffTorchCalc
    from adsorbdiff_amd.trainer import DenoisingTrainer

    m = make_model(4, 2, C=32, hidden=32, heads=2, alpha=16, value=16, ffn=32, ec=32, layers=3, cutoff=12.0)
    m.so3_denoising = True
    m = m.to(DEV)
    params = dict(num_steps=4, ads_std_low=0.1, ads_std_high=10, rot_std_low=0.01, rot_std_high=1.55, ode=True,
                  early_stop=False)
    placement = torch.rand(2, 3, generator=torch.Generator().manual_seed(4))
    outs = []
    for inc in (False, True):
        b = safe_batch(2, 196, seed=23)
        trainer = DenoisingTrainer(m, device=DEV)
        den = Denoiser(b.clone().to(DEV), DiffTorchCalc(trainer),
                       dict(params, placement_noise=placement, incremental_layers=inc), device=DEV)
        out = den.run()
        assert den.steps_applied == 4
        outs.append(out.pos.cpu())
        c = m.engine().counters()
        if inc:
            assert 0 < int(c.inc_rows) < int(c.inc_rows_full)
        else:
            assert int(c.inc_rows_full) == 0
    assert torch.equal(outs[0], outs[1])


def test_eqv2_sampling_on_adsorbate_scores_only_gives_identical_sites():
    from adsorbdiff_amd.denoising_torch import Denoiser, DiffTorchCalc
    from adsorbdiff_amd.trainer import DenoisingTrainer

    m = make_model(4, 2, C=32, hidden=32, heads=2, alpha=16, value=16, ffn=32, ec=32, layers=1, cutoff=12.0)
    m.so3_denoising = True
    m = m.to(DEV)
    params = dict(num_steps=3, ads_std_low=0.1, ads_std_high=10, rot_std_low=0.01, rot_std_high=1.55, ode=True,
                  early_stop=False)
    placement = torch.rand(2, 3, generator=torch.Generator().manual_seed(3))
    outs = []
    for ads_only in (False, True):
        b = safe_batch(2, 36, seed=21)
        trainer = DenoisingTrainer(m, device=DEV)
        den = Denoiser(b.clone().to(DEV), DiffTorchCalc(trainer),
                       dict(params, placement_noise=placement, scores_on_adsorbate_only=ads_only), device=DEV)
        out = den.run()
        assert den.steps_applied == 3
        outs.append(out.pos.cpu())
    assert torch.equal(outs[0], outs[1])
    assert float((outs[0] - safe_batch(2, 36, seed=21).pos).abs().max()) > 0.1  # the adsorbates did move


def test_eqv2_forward_is_run_to_run_and_batch_independent():
    """No atomics on values (row magnitudes are order-independent maxima), sums in CSR order, lifts per row / per node:
    the same forward twice gives the same bits, and a system evaluated alone gives the bits it has inside a batch."""
    m = make_model(6, 2, C=32, hidden=64, heads=2, alpha=16, value=16, ffn=32, ec=32, layers=2, cutoff=12.0).to(DEV)
    b3 = safe_batch(3, 36, seed=17)
    f1, f2 = m(b3.clone().to(DEV))
    g1, g2 = m(b3.clone().to(DEV))
    assert torch.equal(f1, g1) and torch.equal(f2, g2)
    from adsorbdiff_amd.data import Batch

    first = Batch.from_data_list(b3.to_data_list()[:1])
    n0 = int(first.natoms[0])
    h1, h2 = m(first.to(DEV))
    assert torch.equal(h1, f1[:n0]) and torch.equal(h2, f2[:n0])


def test_eqv2_error_contract():
    """The reference's exceptions on this path: an image without neighbours raises ValueError (models/base.py via
    generate_graph; message "An image has no neighbors"), an atomic number outside the embedding / radius tables raises
    ValueError instead of reading out of bounds; the sticky device flags are cleared by the failed check."""
    from adsorbdiff_amd.synthetic import make_batch

    m = make_model(4, 2, C=8, hidden=8, heads=2, alpha=4, value=4, ffn=16, ec=8, layers=1, cutoff=2.0).to(DEV)
    far = make_batch(1, n_slab=16, n_ads=1, seed=3)
    far.pos = far.pos * 0 + torch.arange(far.pos.shape[0]).float()[:, None] * 40.0  # everything far apart
    far.cell = far.cell * 50
    with pytest.raises(ValueError, match="no neighbors"):
        m(far.to(DEV))
    m2 = make_model(4, 2, C=8, hidden=8, heads=2, alpha=4, value=4, ffn=16, ec=8, layers=1, cutoff=12.0).to(DEV)
    b = safe_batch(1, 36, seed=5)
    bad = b.clone()
    bad.atomic_numbers = bad.atomic_numbers.clone()
    bad.atomic_numbers[0] = 200
    with pytest.raises(ValueError, match="atomic number"):
        m2(bad.to(DEV))
    f1, _ = m2(b.to(DEV))  # the flags were cleared
    assert bool(torch.isfinite(f1).all())


def test_eqv2_distance_basis_path_vs_oracle():
    """Radii small enough for the Gaussian distance basis to be non-zero (the table divided by 100: what the
    reference's discarded `/ 100` would have produced, equiformer_v2_denoising.py:168-169): the per-edge radial path
    with its basis window, against the oracle given the same radii."""
    from oracle import eqv2_oracle as Q

    m = make_model(4, 2, C=8, hidden=8, heads=2, alpha=4, value=4, ffn=16, ec=8, layers=1, cutoff=12.0)
    with torch.no_grad():
        m.atom_radii.div_(100.0)
        for n, p in m.named_parameters():  # let the basis part of the first radial layer matter
            if n.endswith("rad_func.net.0.weight"):
                p[:, :600].mul_(3.0)
    b = safe_batch(2, 36, seed=13)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ei, sh, nb = Q.radius_graph_pbc(b.pos, b.cell, b.natoms, 12.0, 20)
    ei, d, v, _ = Q.pbc_distances(b.pos, ei, b.cell, sh, nb)
    with torch.no_grad():
        r1, r2 = Q.eqv2_forward(sd, oracle_hp(m), b.pos, b.atomic_numbers, b.cell, b.natoms, graph=(ei, v),
                                atom_radii=m.atom_radii.detach())
        z1, _ = Q.eqv2_forward(sd, oracle_hp(m), b.pos, b.atomic_numbers, b.cell, b.natoms, graph=(ei, v))
    assert rel_err(z1, r1) > 1e-3, "the distance basis does not reach the outputs of this test model"
    m = m.to(DEV)
    eng = m.engine()
    eng.set_edges(ei, v)
    f1, f2 = m(b.to(DEV))
    e1, e2 = rel_err(f1.cpu(), r1), rel_err(f2.cpu(), r2)
    print(f"distance basis: rel err {e1:.2e} {e2:.2e}")
    assert e1 < REL_TOL and e2 < REL_TOL


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("scale", [1e-4, 1e-2, 1.0, 1e3])
def test_eqv2_linear_f16x3_row_lifts_cover_the_activation_range(mode, scale):
    """The f16x3 dense products of this path lift every A row by its own power of two before the fp16 hi / lo split, so
    the result does not depend on the overall magnitude of the activations (the unscaled split of gemm16.hip loses the
    a_lo term below ~0.1: 4e-4 at 1e-2).  Rows of very different magnitude inside one launch included.  <= 5e-6."""
    import ctypes as C

    from adsorbdiff_amd import lib as L

    lib = L.load()
    torch.manual_seed(0)
    M, N, K = 9000, 640, 448   # spans the 256-row tile kernels (M >= 8192) and a partially filled column tile
    A = torch.randn(M, K, device=DEV) * scale
    A[::7] *= 1e-3             # rows far below their neighbours: the lift is per row
    A[5] = 0.0
    W = torch.randn(N, K, device=DEV) * 0.05
    b = torch.randn(N, device=DEV) * scale
    out = torch.empty(M, N, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.adf_eqv2_linear_forward(A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 0, mode, 1, st))
    ref = A.double() @ W.double().T + b.double()
    err = ((out.double() - ref).norm(dim=1) / ref.norm(dim=1).clamp(min=1e-30))
    assert float(err.max()) < 5e-6, float(err.max())   # every row against its own norm
    # SiLU epilogue, small shape (128-row tile kernel)
    L.check(lib.adf_eqv2_linear_forward(A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), 300, N, K, 2, mode, 1, st))
    ref2 = torch.nn.functional.silu(ref[:300])
    assert rel_err(out[:300].cpu(), ref2.cpu()) < 5e-6


@pytest.mark.parametrize("shape", [(9000, 640, 512), (1000, 1024, 1792), (777, 768, 1536), (5000, 128, 128), (300, 160, 64)])
def test_eqv2_first_convolution_with_streamed_weight_fragments_is_bit_identical(shape):
    """The sampler's first SO(2) convolution (so2_ops.py:158-238) runs `eq_gemm16pw_kernel`: pre-split operand rows through
    LDS, the weights streamed from their MFMA-fragment image into registers.  Same products in the same order as the
    LDS-staged `eq_gemm16p_kernel` (mode 2): equal bit for bit, and within 5e-6 of float64 per row.  Shapes: config 4's
    three orders (N = 1024 / 768 / 640: the last one splits into two eight-wave column tiles and a four-wave remainder),
    a ragged last row tile, N below one tile."""
    import ctypes as C

    from adsorbdiff_amd import lib as L

    lib = L.load()
    torch.manual_seed(1)
    M, N, K = shape
    A = torch.randn(M, K, device=DEV)
    A[::7] *= 1e-3
    A[5] = 0.0
    W = torch.randn(N, K, device=DEV) * 0.05
    b = torch.randn(N, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = A.double() @ W.double().T + b.double()
    for act in (0, 2):
        o2 = torch.empty(M, N, device=DEV)
        o3 = torch.full((M, N), float("nan"), device=DEV)
        L.check(lib.adf_eqv2_linear_forward(A.data_ptr(), W.data_ptr(), b.data_ptr(), o2.data_ptr(), M, N, K, act, 2, 1, st))
        L.check(lib.adf_eqv2_linear_forward(A.data_ptr(), W.data_ptr(), b.data_ptr(), o3.data_ptr(), M, N, K, act, 3, 1, st))
        assert torch.equal(o2, o3)
        r = torch.nn.functional.silu(ref) if act == 2 else ref
        err = ((o3.double() - r).norm(dim=1) / r.norm(dim=1).clamp(min=1e-30))
        assert float(err.max()) < 5e-6, float(err.max())


@pytest.mark.parametrize("exact", [False, True])
def test_eqv2_config4_width_vs_reference_fixture(exact):
    """BASELINE config 4 AT ITS STATED SHAPE (configs/denoising/eqv2_so3.yml:40-75 with lmax_list [6]: C=128, 8 heads,
    attn_hidden 64, alpha 64, value 16, ffn 128, edge_channels 128, 8 blocks, L=6/M=2, K=20, 12 A — bench.py's EQV2_HP)
    on one 200-atom system, against the REFERENCE model's recordings (tests/golden/eqv2_cfg4.npz, e3nn stand-in: S2-grid
    normalisation unpinned): (f1, f2) at 1e-4, per atom against the largest atom, and the node embedding after the
    edge-degree embedding and after each of the 8 blocks per block and per degree (strided sample + norms over all
    atoms).  Both arithmetics; on the reference's edge list and on the device-built graph (no ties in this cell)."""
    from tests.helpers import cfg4_model_and_fixture, check_cfg4_blocks

    m, fx = cfg4_model_and_fixture()
    m = m.to(DEV)
    b = batch_from_fixture(fx, device=DEV)
    eng = m.engine()
    eng.set_arithmetic(exact)
    eng.set_edges(torch.from_numpy(fx["edge_index"]), torch.from_numpy(fx["edge_vec"]))
    f1, f2, xb = eng.forward(b, return_blocks=True)
    e1, e2 = rel_err(f1.cpu(), fx["f1"]), rel_err(f2.cpu(), fx["f2"])
    worst = check_cfg4_blocks(xb, fx, REL_TOL)
    print(f"config-4 width, {'exact f32' if exact else 'f16x3'}: rel err f1 {e1:.2e} f2 {e2:.2e}, worst block/degree {worst:.2e}")
    assert e1 < REL_TOL and e2 < REL_TOL
    for got, ref in ((f1.cpu().numpy(), fx["f1"]), (f2.cpu().numpy(), fx["f2"])):
        assert np.abs(got - ref).max() < REL_TOL * np.linalg.norm(ref, axis=1).max()
    # the device-built graph (strict top-K) gives the same outputs
    m._engine.close()
    m._engine = None
    eng = m.engine()
    eng.set_arithmetic(exact)
    g1, g2 = m(b)
    assert rel_err(g1.cpu(), fx["f1"]) < REL_TOL and rel_err(g2.cpu(), fx["f2"]) < REL_TOL


def test_eqv2_config4_width_vs_oracle():
    """The same shape against oracle/eqv2_oracle.py evaluated here (≈15 s of CPU) on ANOTHER 200-atom system than the
    fixture's, on the device-built graph: outputs at 1e-4 (Frobenius and per atom against the largest atom)."""
    from oracle import eqv2_oracle as Q
    from tests.helpers import CFG4_ORACLE_HP, cfg4_model_and_fixture

    m, _ = cfg4_model_and_fixture()
    b = safe_batch(1, 196, seed=23)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        r1, r2 = Q.eqv2_forward(sd, CFG4_ORACLE_HP, b.pos, b.atomic_numbers, b.cell, b.natoms)
    f1, f2 = m.to(DEV)(b.to(DEV))
    e1, e2 = rel_err(f1.cpu(), r1), rel_err(f2.cpu(), r2)
    print(f"config-4 width vs oracle: rel err {e1:.2e} {e2:.2e}")
    assert e1 < REL_TOL and e2 < REL_TOL
    for got, ref in ((f1.cpu(), r1), (f2.cpu(), r2)):
        assert float((got - ref).abs().max()) < REL_TOL * float(ref.norm(dim=1).max())


def test_eqv2_sampling_with_trajectory_sink(tmp_path):
    """adf_eqv2_sample_traj: the EquiformerV2 sampler with the asynchronous trajectory sink (csrc/frames.hip + trajectory.py)
    ends at the same positions as without it and writes one frame per applied step, the last one equal to the result."""
    from adsorbdiff_amd.denoising_torch import Denoiser, DiffTorchCalc
    from adsorbdiff_amd.trainer import DenoisingTrainer

    m = make_model(4, 2, C=8, hidden=8, heads=2, alpha=4, value=4, ffn=16, ec=8, layers=1, cutoff=12.0)
    m.so3_denoising = True
    b = safe_batch(2, 196, seed=21)
    params = dict(num_steps=4, ads_std_low=0.1, ads_std_high=10, rot_std_low=0.01, rot_std_high=1.55, ode=True, early_stop=False)
    torch.manual_seed(5)
    noise = torch.rand(2, 3)
    trainer = DenoisingTrainer(m.to(DEV), device=DEV)
    outs = []
    for traj in (None, tmp_path):
        den = Denoiser(b.clone().to(DEV), DiffTorchCalc(trainer), dict(params, placement_noise=noise), device=DEV,
                       traj_dir=traj, traj_names=[str(s) for s in b.sid], save_full_traj=True)
        outs.append(den.run().pos.cpu())
        assert den.steps_applied == 4
    assert torch.equal(outs[0], outs[1])
    n0 = int(b.natoms[0])
    z = np.load(tmp_path / f"{b.sid[0]}.npz")
    assert z["positions"].shape == (4, n0, 3)
    assert np.array_equal(z["positions"][-1], outs[1][:n0].numpy())
    assert np.array_equal(np.load(tmp_path / f"batch_{b.sid[0]}.frames.npy")[-1], outs[1].numpy())
