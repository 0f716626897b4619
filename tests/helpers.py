"""Shared helpers for the parity tests."""
from pathlib import Path

import numpy as np
import torch

from adsorbdiff_amd.data import Batch

GOLDEN = Path(__file__).resolve().parent / "golden"


def load_npz(name):
    with np.load(GOLDEN / name, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def batch_from_fixture(fx, pos_key="pos", device="cpu"):
    b = Batch()
    b.pos = torch.from_numpy(fx[pos_key]).float().clone()
    b.atomic_numbers = torch.from_numpy(fx["atomic_numbers"]).float()
    b.tags = torch.from_numpy(fx["tags"]).long()
    b.fixed = torch.from_numpy(fx["fixed"]).long()
    b.cell = torch.from_numpy(fx["cell"]).float()
    b.natoms = torch.from_numpy(fx["natoms"]).long()
    b.batch = torch.from_numpy(fx["batch"]).long()
    b.sid = [str(i) for i in range(len(b.natoms))]
    return b.to(device)


def state_dict_from_fixture(fx):
    return {k[4:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("sd::")}


def canon_edges(src, dst, dist, vec):
    """Canonical order for an edge multiset: by (dst, src, dist, vec)."""
    src, dst = np.asarray(src, np.int64), np.asarray(dst, np.int64)
    dist, vec = np.asarray(dist, np.float64), np.asarray(vec, np.float64)
    order = np.lexsort((np.round(vec[:, 2], 4), np.round(vec[:, 1], 4), np.round(vec[:, 0], 4), np.round(dist, 4), src, dst))
    return src[order], dst[order], dist[order], vec[order]


def rel_err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def row_rel_err(a, b, floor=1e-7):
    """max over rows of |a_i - b_i| / max(|b_i|, floor): every system / atom bounded against its OWN magnitude (the
    Frobenius ratio of rel_err lets large rows hide small ones)."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    return float(((a - b).norm(dim=1) / b.norm(dim=1).clamp(min=floor)).max())


def max_abs_err_rel_to_max(a, b):
    """max |a - b| / max row norm of b: the per-atom heads bounded element-wise against the largest atom."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.reshape(b.shape[0], -1).norm(dim=1).max().clamp(min=1e-30))


def refill_parameters_by_name(model, emb_scale=300.0):
    """Deterministic weights that depend on (parameter name, shape) only — not on the order in which a class draws its
    initial values — so that the reference model in oracle/make_golden.py and the mirror class in a test hold the SAME
    31 M weights without storing them.  Magnitudes follow the reference's `weight_init: uniform`
    (equiformer_v2_oc20.py `_uniform_init_linear_weights`: U(-1/sqrt(fan_in), 1/sqrt(fan_in))); the atom edge embeddings
    (U(-1e-3, 1e-3) in the reference) are lifted by `emb_scale` to trained-like magnitudes; parameters with fewer than two
    dimensions (biases, norm gains: constants in both classes) are left as constructed."""
    import math
    import zlib

    with torch.no_grad():
        for name, p in sorted(model.named_parameters()):
            if p.dim() < 2 or name == "atom_radii":
                continue
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            u = torch.rand(p.shape, generator=g, dtype=torch.float32) * 2.0 - 1.0
            if name.endswith("source_embedding.weight") or name.endswith("target_embedding.weight"):
                a = 1e-3 * emb_scale
            elif name == "sphere_embedding.weight":
                a = math.sqrt(3.0)
            else:
                a = 1.0 / math.sqrt(p.shape[-1])
            p.copy_(u * a)
    return model


# ---- BASELINE config 4 (EquiformerV2 at its stated width): shared by the CPU oracle test and the GPU tests
CFG4_KW = dict(max_neighbors=20, max_radius=12.0, max_num_elements=90, num_layers=8, sphere_channels=128,
               attn_hidden_channels=64, num_heads=8, attn_alpha_channels=64, attn_value_channels=16,
               ffn_hidden_channels=128, norm_type="layer_norm_sh", lmax_list=[6], mmax_list=[2], grid_resolution=18,
               edge_channels=128, attn_activation="silu", ffn_activation="silu", use_grid_mlp=True, use_sep_s2_act=True,
               alpha_drop=0.0, drop_path_rate=0.0, weight_init="uniform", FOR_denoising=True)
CFG4_ORACLE_HP = dict(lmax=6, mmax=2, num_layers=8, sphere_channels=128, attn_hidden_channels=64, num_heads=8,
                      attn_alpha_channels=64, attn_value_channels=16, ffn_hidden_channels=128, grid_resolution=18,
                      max_radius=12.0, max_neighbors=20)


def cfg4_model_and_fixture():
    """The mirror class at the BASELINE config-4 width with the fixture's weights (rebuilt from parameter names, not
    stored: tests/helpers.py::refill_parameters_by_name — oracle/make_golden.py asserts the reference model holds the
    same values)."""
    from adsorbdiff_amd.equiformer_v2_denoising import EquiformerV2S_OC20_DenoisingPos

    fx = load_npz("eqv2_cfg4.npz")
    torch.manual_seed(0)
    m = refill_parameters_by_name(EquiformerV2S_OC20_DenoisingPos(None, None, None, **CFG4_KW).eval(), float(fx["emb_scale"]))
    assert sum(p.numel() for p in m.parameters()) == int(fx["n_params"])
    return m, fx


def check_cfg4_blocks(xb, fx, tol):
    """xb [9, N, 49, 128] (after the edge-degree embedding and after each of the 8 blocks) against the fixture's strided
    sample, per block and per degree, and against the reference's per-degree norms over ALL atoms and channels."""
    sa, sc = int(fx["atom_stride"]), int(fx["channel_stride"])
    xb = torch.as_tensor(xb).float().cpu()
    ref = torch.from_numpy(fx["x_blocks_sample"])
    got = xb[:, ::sa, :, ::sc]
    assert got.shape == ref.shape, (got.shape, ref.shape)
    worst = 0.0
    for k in range(ref.shape[0]):
        for l in range(7):
            e = rel_err(got[k, :, l * l:(l + 1) ** 2], ref[k, :, l * l:(l + 1) ** 2])
            worst = max(worst, e)
            assert e < tol, (k, l, e)
            n = float(xb[k, :, l * l:(l + 1) ** 2].double().norm())
            assert abs(n - float(fx["x_blocks_degree_norms"][k, l])) < tol * float(fx["x_blocks_degree_norms"][k, l]), (k, l)
    return worst
