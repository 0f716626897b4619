"""CPU tests of the EquiformerV2 host side: the constant SO(3) tables the product derives (adsorbdiff_amd/so3_math.py)
against data recorded from the reference run (tests/golden/eqv2_*.npz, oracle/make_golden.py), properties of the S2 grid
that pin the e3nn stand-in as far as it can be pinned without e3nn, and the mirror class's parameter names."""
import math

import numpy as np
import pytest
import torch

from adsorbdiff_amd import so3_math as S
from tests.helpers import load_npz, state_dict_from_fixture


def test_j_matrices_equal_the_vendored_table():
    """J_l solved from the real harmonics = the reference's Jd.pt (recorded in eqv2_jd.npz) to 1e-12; symmetric
    involutions."""
    jd = load_npz("eqv2_jd.npz")
    for l, j in enumerate(S.j_matrices(6)):
        assert np.abs(j - jd[f"J{l}"]).max() < 1e-12
        assert np.abs(j @ j - np.eye(2 * l + 1)).max() < 1e-12 and np.abs(j - j.T).max() < 1e-12


@pytest.mark.parametrize("name,lmax", [("eqv2_l4m2.npz", 4), ("eqv2_l6m2.npz", 6)])
def test_grid_matrices_equal_the_reference_buffers(name, lmax):
    """to_grid / from_grid of SO3_Grid(lmax, mmax=2, resolution 18) as the reference model held them (float32)."""
    fx = load_npz(name)
    to, fr = S.s2_grid_matrices(lmax, 2, 18)
    assert np.abs(to - fx["to_grid_mat"].reshape(324, -1)).max() < 2e-6
    assert np.abs(fr - fx["from_grid_mat"].reshape(324, -1)).max() < 2e-6


def test_wigner_product_form_equals_the_definition():
    """D_l(Rx(b) Ry(g)) = J Z(b) J Z(g): the form the device kernel evaluates (eq_wigner_kernel) against
    Y(R x) = D Y(x) solved directly."""
    rng = np.random.default_rng(3)
    J = S.j_matrices(6)
    for _ in range(4):
        n = rng.standard_normal(3)
        n /= np.linalg.norm(n)
        rho = math.hypot(n[0], n[2])
        cg, sg, cb, sb = n[2] / rho, -n[0] / rho, n[1], -rho
        R = np.array([[1, 0, 0], [0, cb, -sb], [0, sb, cb]]) @ np.array([[cg, 0, sg], [0, 1, 0], [-sg, 0, cg]])
        assert np.allclose(R @ n, [0, 1, 0], atol=1e-12)
        ref = S.wigner_from_matrix(6, R)
        b, g = math.atan2(sb, cb), math.atan2(sg, cg)
        for l in range(7):
            D = J[l] @ S.y_rotation_matrix(l, b) @ J[l] @ S.y_rotation_matrix(l, g)
            assert np.abs(D - ref[l]).max() < 1e-12


@pytest.mark.parametrize("lmax", [4, 6])
def test_s2_grid_properties(lmax):
    """What can be pinned about the S2 grid without e3nn: from_grid o to_grid = identity on band-limited signals; the
    18 x 18 quadrature integrates products of two degree-<= lmax harmonics exactly (orthonormality); a rotated signal
    sampled on the grid equals the signal with Wigner-rotated coefficients."""
    to, fr = S.s2_grid_matrices(lmax, lmax, 18)
    n = (lmax + 1) ** 2
    assert np.abs(fr.T @ to - np.eye(n)).max() < 1e-10
    # quadrature exactness: Y^T diag(w) Y = I with the weights implied by from_grid = diag(w) Y (component normalisation)
    betas = (np.arange(18) + 0.5) / 18 * math.pi
    alphas = np.arange(18) / 18 * 2 * math.pi
    pts = np.stack([np.outer(np.sin(betas), np.sin(alphas)), np.outer(np.cos(betas), np.ones(18)),
                    np.outer(np.sin(betas), np.cos(alphas))], -1).reshape(-1, 3)
    Y = S.real_sh(lmax, pts)
    qw = np.repeat(S._quadrature_weights(9) * 18.0 ** 2 / 18.0, 18) * 4 * math.pi
    assert np.abs(Y.T @ (qw[:, None] * Y) - np.eye(n)).max() < 1e-10
    # rotated grids: f(R^-1 p) on the grid <-> D(R) c
    rng = np.random.default_rng(0)
    c = rng.standard_normal(n)
    A = rng.standard_normal((3, 3))
    R, _ = np.linalg.qr(A)
    if np.linalg.det(R) < 0:
        R[:, 0] = -R[:, 0]
    D = S.wigner_from_matrix(lmax, R)
    Dfull = np.zeros((n, n))
    for l in range(lmax + 1):
        Dfull[l * l:(l + 1) ** 2, l * l:(l + 1) ** 2] = D[l]
    # Y(R x) = D Y(x)  =>  sum_i c_i Y_i(R x) = (D^T c) . Y(x)
    assert np.abs(S.real_sh(lmax, pts @ R.T) @ c - Y @ (Dfull.T @ c)).max() < 1e-10


def test_mirror_has_the_reference_parameter_names():
    from adsorbdiff_amd.equiformer_v2_denoising import EquiformerV2S_OC20_DenoisingPos
    from adsorbdiff_amd.eqv2_engine import weight_names

    fx = load_npz("eqv2_l4m2.npz")
    sd = state_dict_from_fixture(fx)
    m = EquiformerV2S_OC20_DenoisingPos(
        None, None, None, max_neighbors=20, max_radius=6.0, max_num_elements=90, num_layers=2, sphere_channels=8,
        attn_hidden_channels=8, num_heads=2, attn_alpha_channels=4, attn_value_channels=4, ffn_hidden_channels=16,
        norm_type="layer_norm_sh", lmax_list=[4], mmax_list=[2], grid_resolution=18, edge_channels=8,
        attn_activation="silu", ffn_activation="silu", use_grid_mlp=True, use_sep_s2_act=True, weight_init="uniform",
        FOR_denoising=True)
    mine = {k: tuple(v.shape) for k, v in m.named_parameters() if k != "atom_radii"}
    assert mine == {k: tuple(v.shape) for k, v in sd.items()}
    m.load_state_dict(sd)
    assert all(torch.equal(dict(m.named_parameters())[k], v) for k, v in sd.items())
    # every tensor the library binds exists; the unused energy head does not reach it
    names = weight_names(2, 2)
    assert set(names) <= set(mine) | {"atom_radii"} and not any(n.startswith("energy_block") for n in names)
    with pytest.raises(ValueError):
        EquiformerV2S_OC20_DenoisingPos(None, None, None, lmax_list=[4], mmax_list=[2], grid_resolution=18,
                                        norm_type="rms_norm_sh", FOR_denoising=True)
    with pytest.raises(RuntimeError):  # no CPU fallback
        m(None if False else type("B", (), {"pos": torch.zeros(1, 3)})())


def test_load_state_dict_is_strict_about_everything_but_the_constant_buffers():
    """a checkpoint of another configuration must not load silently.  Ignored: only the reference's constant
    buffers (grid matrices, index tables, Gaussian offsets); raised under strict: other unexpected keys, missing
    parameters, shape mismatches."""
    import pytest

    from adsorbdiff_amd.equiformer_v2_denoising import EquiformerV2S_OC20_DenoisingPos as M

    kw = dict(max_neighbors=20, max_radius=6.0, max_num_elements=90, num_layers=1, sphere_channels=8,
              attn_hidden_channels=8, num_heads=2, attn_alpha_channels=4, attn_value_channels=4, ffn_hidden_channels=16,
              norm_type="layer_norm_sh", lmax_list=[4], mmax_list=[2], grid_resolution=18, edge_channels=8,
              num_distance_basis=16, attn_activation="silu", ffn_activation="silu", use_grid_mlp=True,
              use_sep_s2_act=True, weight_init="uniform", FOR_denoising=True)
    m = M(None, None, None, **kw)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    ok = dict(sd)
    ok["SO3_grid.4.2.to_grid_mat"] = torch.zeros(3)            # constant buffers of the reference: ignored
    ok["blocks.0.ga.so2_conv_1.mappingReduced.m_complex"] = torch.zeros(3)
    ok["distance_expansion.offset"] = torch.zeros(16)
    ok["blocks.0.ffn.so3_linear_1.expand_index"] = torch.zeros(25)
    del ok["atom_radii"]                                        # a constant table: may be absent
    res = m.load_state_dict(ok)
    assert not res.missing_keys and not res.unexpected_keys
    bad = dict(sd)
    bad["blocks.1.ga.alpha_dot"] = torch.zeros(2, 4)            # a deeper checkpoint
    with pytest.raises(RuntimeError, match="unexpected"):
        m.load_state_dict(bad)
    bad = dict(sd)
    bad["energy_embedding.weight"] = torch.zeros(4, 4)          # a conditional checkpoint
    with pytest.raises(RuntimeError, match="unexpected"):
        m.load_state_dict(bad)
    assert "energy_embedding.weight" in m.load_state_dict(bad, strict=False).unexpected_keys
    bad = {k: v for k, v in sd.items() if k != "blocks.0.ga.alpha_dot"}
    with pytest.raises(RuntimeError, match="missing"):
        m.load_state_dict(bad)
    bad = dict(sd)
    bad["blocks.0.ga.alpha_dot"] = torch.zeros(3, 4)
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_state_dict(bad)
