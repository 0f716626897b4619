"""TEST INFRASTRUCTURE — CPU restatement of the EquiformerV2 denoiser's forward (BASELINE config 4).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.

PARITY STATUS — UNPINNED (capped at "partial"): the reference model needs ``e3nn==0.4.4`` (spherical-harmonic grids,
rotation angle conventions), which is neither installed nor vendored; both the reference run that produced
tests/golden/eqv2_l*m*.npz and this file use the stand-in oracle/refshim/e3nn_standin.py (see its docstring for what
could and could not be checked).  WITHIN that stand-in this restatement reproduces the reference forward to 1e-5
(oracle/make_golden.py section 7 asserts it function by function against the imported reference).

Formulation (one resolution; L = lmax, M = mmax, C = sphere channels; S = (L+1)^2 coefficients in (l, m) order,
S_r = sum_l min(2l+1, 2M+1) "reduced" coefficients kept in an edge's own frame):

  graph            strict top-K periodic radius graph, NOT symmetrised      models/base.py:33-123
  edge scalars     Gaussian basis of d - r[Z_j] - r[Z_i] (radii in pm: the basis is identically 0 for every real
                   edge), source / target element embeddings                equiformer_v2_denoising.py:207-213
  edge frame       rotation taking the edge direction to the polar axis; any roll about it gives the same outputs
                   (edge_rot_mat.py:6-63 draws it at random: 1e-6 spread)   so3.py:495-521
  Wigner D         here: D_l(R) solved from Y_l(R x) = D_l(R) Y_l(x) on fixed sample points with the stand-in's real
                   harmonics — no table of J matrices needed (the reference multiplies z-rotations with the vendored
                   Jd.pt, wigner.py:18-40; make_golden.py checks both agree to 1e-10)
  SO(2) conv       per order m a dense map on the (l >= m) coefficients; (+m, -m) pairs mix like complex numbers
                                                                            so2_ops.py:12-79, 82-262
  attention block  rotate [x_j | x_i] to the edge frame, SO(2) conv with radial weights, separable S2 activation on a
                   grid, second SO(2) conv, per-target softmax over heads, rotate back with the m-truncation rescale,
                   scatter-add, SO(3) linear                                transformer_block.py:226-372
  feed forward     SO(3) linear, point-wise MLP on the S2 grid, scalar gate, SO(3) linear      transformer_block.py:473-531
  norm             LayerNorm on l = 0, one degree-balanced RMS over l > 0   layer_norm.py:129-250
  outputs          l = 1 coefficients of two single-channel attention blocks   equiformer_v2_denoising.py:300-318
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .painn_oracle import pbc_distances, radius_graph_pbc
from .refshim import e3nn_standin as E3

AVG_DEGREE = 23.395238876342773  # rescale of the edge-degree embedding (equiformer_v2_denoising.py:22-24)


# --------------------------------------------------------------------------------------------------- index bookkeeping
def lm_list(L: int, M: int) -> List[Tuple[int, int]]:
    """(l, m) of the coefficients kept with |m| <= min(l, M), degree-major (so3.py:56-72)."""
    return [(l, m) for l in range(L + 1) for m in range(-min(l, M), min(l, M) + 1)]


def reduced_mask(L: int, M: int) -> torch.Tensor:
    """Positions, inside the full (L+1)^2 list, of the coefficients with |m| <= M (so3.py:137-155)."""
    full = lm_list(L, L)
    return torch.tensor([k for k, (l, m) in enumerate(full) if abs(m) <= M])


def order_major_perm(L: int, M: int) -> torch.Tensor:
    """perm[k] = position in the degree-major reduced list of the k-th coefficient of the order-major layout:
    m = 0 for all l, then for m = 1..M the +m entries (l = m..L) followed by the -m entries (so3.py:83-103)."""
    red = lm_list(L, M)
    out = [red.index((l, 0)) for l in range(L + 1)]
    for m in range(1, M + 1):
        out += [red.index((l, m)) for l in range(m, L + 1)]
        out += [red.index((l, -m)) for l in range(m, L + 1)]
    return torch.tensor(out)


def truncation_rescale(L: int, M: int) -> torch.Tensor:
    """sqrt((2l+1)/(2M+1)) for the reduced coefficients of degree l > M, 1 otherwise (so3.py:160-186, 572-613)."""
    return torch.tensor([math.sqrt((2 * l + 1) / (2 * M + 1)) if l > M else 1.0 for l, _ in lm_list(L, M)])


# --------------------------------------------------------------------------------------------------- geometry
def edge_frames(vec: torch.Tensor) -> torch.Tensor:
    """[E,3,3] rotation whose second row is the edge direction (edge_rot_mat.py:51-63: rows = (z', x', -y') with
    x' = v / |v|).  The helper direction is the coordinate axis least aligned with the edge instead of a random vector."""
    x = vec / vec.norm(dim=1, keepdim=True)
    helper = torch.eye(3, dtype=vec.dtype)[x.abs().argmin(dim=1)]
    z = torch.cross(x, helper, dim=1)
    z = z / z.norm(dim=1, keepdim=True)
    y = torch.cross(x, z, dim=1)
    y = y / y.norm(dim=1, keepdim=True)
    return torch.stack([z, x, -y], dim=1)


_SAMPLES: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}


def wigner_from_rotation(L: int, R: torch.Tensor) -> torch.Tensor:
    """Block-diagonal [E, S, S] with D_l(R): Y_l(R x) = D_l(R) Y_l(x), solved on a fixed generic point set in fp64."""
    if L not in _SAMPLES:
        g = torch.Generator().manual_seed(1234)
        pts = torch.randn(4 * (L + 1) ** 2, 3, generator=g, dtype=torch.float64)
        pts = pts / pts.norm(dim=1, keepdim=True)
        Y = E3.real_sh(L, pts).double()          # [P, S]
        _SAMPLES[L] = (pts, torch.linalg.pinv(Y))  # pinv [S, P]
    pts, pinv = _SAMPLES[L]
    S = (L + 1) ** 2
    rp = torch.einsum("eij,pj->epi", R.double(), pts)            # rotated points
    Yr = E3.real_sh(L, rp.reshape(-1, 3)).double().reshape(R.shape[0], pts.shape[0], S)
    D = torch.einsum("sp,epk->eks", pinv, Yr)                     # Y(Rx)^T = Y(x)^T D^T  ->  D[k, s]
    mask = torch.zeros(S, S, dtype=torch.bool)
    for l in range(L + 1):
        mask[l * l:(l + 1) ** 2, l * l:(l + 1) ** 2] = True
    return (D * mask).float()


# --------------------------------------------------------------------------------------------------- small layers
def layer_norm(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def radial_mlp(sd, pre: str, x: torch.Tensor) -> torch.Tensor:
    """Linear, LayerNorm, SiLU, Linear, LayerNorm, SiLU, Linear (radial_function.py:11-32)."""
    x = F.linear(x, sd[pre + "net.0.weight"], sd[pre + "net.0.bias"])
    x = F.silu(layer_norm(x, sd[pre + "net.1.weight"], sd[pre + "net.1.bias"]))
    x = F.linear(x, sd[pre + "net.3.weight"], sd[pre + "net.3.bias"])
    x = F.silu(layer_norm(x, sd[pre + "net.4.weight"], sd[pre + "net.4.bias"]))
    return F.linear(x, sd[pre + "net.6.weight"], sd[pre + "net.6.bias"])


def so3_linear(sd, pre: str, x: torch.Tensor, L: int) -> torch.Tensor:
    """One weight matrix per degree, bias on l = 0 only (so3.py:694-745)."""
    deg = torch.tensor([l for l, _ in lm_list(L, L)])
    out = torch.einsum("nsi,soi->nso", x, sd[pre + "weight"][deg])
    out[:, 0] = out[:, 0] + sd[pre + "bias"]
    return out


def norm_sh(sd, pre: str, x: torch.Tensor, L: int, eps: float = 1e-5) -> torch.Tensor:
    """layer_norm_sh (layer_norm.py:129-250)."""
    out0 = layer_norm(x[:, :1], sd[pre + "norm_l0.weight"], sd[pre + "norm_l0.bias"], eps)
    if L == 0:
        return out0
    deg = torch.tensor([l for l, _ in lm_list(L, L)][1:])
    w = (1.0 / (2.0 * deg + 1.0) / L).to(x.dtype)                       # every degree weighs the same
    ms = torch.einsum("nsc,s->nc", x[:, 1:] ** 2, w).mean(dim=1)         # [N]
    scale = (ms + eps).pow(-0.5)[:, None, None] * sd[pre + "affine_weight"][deg - 1][None]
    return torch.cat([out0, x[:, 1:] * scale], dim=1)


class Grids:
    """to-grid / from-grid matrices [beta, alpha, coefficient] of the stand-in S2 grid with the reference's
    m-truncation rescale (so3.py:566-613)."""

    def __init__(self, L: int, M: int, res: int) -> None:
        to, fr = E3.ToS2Grid(L, (res, res), normalization="component"), E3.FromS2Grid((res, res), L, normalization="component")
        full_to = torch.einsum("mbi,am->bai", to.shb, to.sha)
        full_fr = torch.einsum("am,mbi->bai", fr.sha, fr.shb)
        self.to_full, self.from_full = full_to, full_fr
        deg = torch.tensor([l for l, _ in lm_list(L, L)])
        scale = torch.where(deg > M, torch.sqrt((2.0 * deg + 1.0) / (2 * M + 1)), torch.ones_like(deg, dtype=torch.float32))
        rm = reduced_mask(L, M)
        self.to_red = (full_to * scale)[:, :, rm] if M != L else full_to
        self.from_red = (full_fr * scale)[:, :, rm] if M != L else full_fr


def so2_conv(sd, pre: str, x: torch.Tensor, L: int, M: int, cout: int, radial: Optional[torch.Tensor], extra: int = 0):
    """SO(2) convolution on reduced, degree-major coefficients x [E, S_r, Cin] -> [E, S_r, cout] (+ `extra` scalar
    outputs of the m = 0 map).  radial [E, sum_m (L-m+1) Cin]: per-edge weights of the inputs (so2_ops.py:188-262)."""
    E, _, cin = x.shape
    perm = order_major_perm(L, M)
    xm = x[:, perm]
    n0 = L + 1
    x0 = xm[:, :n0].reshape(E, n0 * cin)
    off_r = 0
    if radial is not None:
        x0 = x0 * radial[:, :n0 * cin]
        off_r = n0 * cin
    y0 = F.linear(x0, sd[pre + "fc_m0.weight"], sd[pre + "fc_m0.bias"])
    ex = y0[:, :extra] if extra else None
    pieces = [y0[:, extra:].reshape(E, n0, cout)]
    off = n0
    for m in range(1, M + 1):
        nm = L - m + 1
        blk = xm[:, off:off + 2 * nm].reshape(E, 2, nm * cin)          # [+m | -m]
        if radial is not None:
            blk = blk * radial[:, None, off_r:off_r + nm * cin]
            off_r += nm * cin
        y = F.linear(blk, sd[pre + f"so2_m_conv.{m - 1}.fc.weight"])   # [E, 2, 2 nm cout]
        half = nm * cout
        yr, yi = y[..., :half], y[..., half:]
        plus = yr[:, 0] - yi[:, 1]                                     # complex product (a + i b)(w_r + i w_i)
        minus = yr[:, 1] + yi[:, 0]
        pieces += [plus.reshape(E, nm, cout), minus.reshape(E, nm, cout)]
        off += 2 * nm
    ym = torch.cat(pieces, dim=1)
    out = torch.empty_like(ym)
    out[:, perm] = ym                                                   # back to degree-major
    return (out, ex) if extra else out


def smooth_leaky_relu(x, alpha=0.2):
    return (1 + alpha) / 2 * x + (1 - alpha) / 2 * x * (2 * torch.sigmoid(x) - 1)


def segment_softmax(a: torch.Tensor, index: torch.Tensor, n: int) -> torch.Tensor:
    """softmax over the edges that share a target (torch_geometric.utils.softmax)."""
    mx = torch.full((n, a.shape[1]), -float("inf"), dtype=a.dtype).scatter_reduce(0, index[:, None].expand_as(a), a, "amax")
    e = torch.exp(a - mx[index])
    den = torch.zeros(n, a.shape[1], dtype=a.dtype).index_add_(0, index, e)
    return e / (den[index] + 1e-16)


# --------------------------------------------------------------------------------------------------- blocks
def attention_block(sd, pre, x, Z, basis, src, dst, D, hp, grids: Grids, out_channels: int) -> torch.Tensor:
    """SO2EquivariantGraphAttention (transformer_block.py:226-372) with use_atom_edge_embedding, separable S2
    activation, attention re-normalisation, no dropout."""
    L, M = hp["lmax"], hp["mmax"]
    H, A, V, hid = hp["num_heads"], hp["attn_alpha_channels"], hp["attn_value_channels"], hp["attn_hidden_channels"]
    N = x.shape[0]
    rm = reduced_mask(L, M)
    scal = torch.cat([basis, sd[pre + "source_embedding.weight"][Z[src]], sd[pre + "target_embedding.weight"][Z[dst]]], dim=1)
    msg = torch.cat([x[src], x[dst]], dim=2)                          # [E, S, 2C]
    msg = torch.bmm(D[:, rm, :], msg)                                 # edge frame, |m| <= M
    radial = radial_mlp(sd, pre + "so2_conv_1.rad_func.", scal)
    msg, ex = so2_conv(sd, pre + "so2_conv_1.", msg, L, M, hid, radial, extra=H * A + hid)
    alpha_in, gate = ex[:, :H * A], ex[:, H * A:]
    # separable S2 activation: SiLU on the scalar gate, point-wise SiLU on the grid for l > 0 (activation.py:176-202)
    g = torch.einsum("bai,zic->zbac", grids.to_red, msg)
    act = torch.einsum("bai,zbac->zic", grids.from_red, F.silu(g))
    msg = torch.cat([F.silu(gate)[:, None, :], act[:, 1:]], dim=1)
    msg = so2_conv(sd, pre + "so2_conv_2.", msg, L, M, H * V, None)
    a = layer_norm(alpha_in.reshape(-1, H, A), sd[pre + "alpha_norm.weight"], sd[pre + "alpha_norm.bias"])
    a = torch.einsum("ehk,hk->eh", smooth_leaky_relu(a), sd[pre + "alpha_dot"])
    a = segment_softmax(a, dst, N)
    msg = (msg.reshape(msg.shape[0], -1, H, V) * a[:, None, :, None]).reshape(msg.shape[0], -1, H * V)
    back = D.transpose(1, 2)[:, :, rm] * truncation_rescale(L, M)[None, None, :]
    msg = torch.bmm(back, msg)                                        # [E, S, H V]
    agg = torch.zeros(N, msg.shape[1], msg.shape[2], dtype=msg.dtype).index_add_(0, dst, msg)
    return so3_linear(sd, pre + "proj.", agg, L)


def feed_forward(sd, pre, x, hp, grids: Grids) -> torch.Tensor:
    """FeedForwardNetwork with use_grid_mlp and use_sep_s2_act (transformer_block.py:473-531)."""
    L = hp["lmax"]
    gate = F.silu(F.linear(x[:, :1], sd[pre + "scalar_mlp.0.weight"], sd[pre + "scalar_mlp.0.bias"]))
    h = so3_linear(sd, pre + "so3_linear_1.", x, L)
    g = torch.einsum("bai,zic->zbac", grids.to_full, h)
    g = F.linear(F.silu(F.linear(F.silu(F.linear(g, sd[pre + "grid_mlp.0.weight"])), sd[pre + "grid_mlp.2.weight"])),
                 sd[pre + "grid_mlp.4.weight"])
    h = torch.einsum("bai,zbac->zic", grids.from_full, g)
    h = torch.cat([gate, h[:, 1:]], dim=1)
    return so3_linear(sd, pre + "so3_linear_2.", h, L)


def eqv2_forward(sd: Dict[str, torch.Tensor], hp: dict, pos, atomic_numbers, cell, natoms,
                 atom_radii: Optional[torch.Tensor] = None, graph: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """(f1 [N,3], f2 [N,3]) of EquiformerV2S_OC20_DenoisingPos.forward with FOR_denoising (equiformer_v2_denoising.py:
    184-318).  hp: lmax, mmax, num_layers, sphere_channels, attn_hidden_channels, num_heads, attn_alpha_channels,
    attn_value_channels, ffn_hidden_channels, grid_resolution, max_radius, max_neighbors.  atom_radii: [101] in the
    reference's units (pm); None = the basis is taken as exactly zero, which is what any finite tabulated radius gives.
    graph = (edge_index [2,E] (source, target), edge_vec [E,3]): use this edge list instead of building one — in small
    cells the +a / -a images of an atom tie exactly at the K-th place and the reference's pick is implementation-defined
    (exact ties), so fixtures carry the reference's list."""
    L, M, C = hp["lmax"], hp["mmax"], hp["sphere_channels"]
    Z = atomic_numbers.long()
    N = Z.shape[0]
    if graph is None:
        ei, sh, nb = radius_graph_pbc(pos, cell, natoms, hp["max_radius"], hp["max_neighbors"])
        ei, d, v, _ = pbc_distances(pos, ei, cell, sh, nb)
    else:
        ei, v = graph
        d = v.norm(dim=1)
    src, dst = ei[0], ei[1]
    nbasis = 600
    if atom_radii is None:
        basis = torch.zeros(d.shape[0], nbasis)
    else:
        offs = torch.linspace(0.0, hp["max_radius"], nbasis)
        coeff = -0.5 / (2.0 * (offs[1] - offs[0]).item()) ** 2
        dd = d - atom_radii[Z[src]] - atom_radii[Z[dst]]
        basis = torch.exp(coeff * (dd[:, None] - offs[None, :]) ** 2)
    D = wigner_from_rotation(L, edge_frames(v))
    g_red, S = Grids(L, M, hp["grid_resolution"]), (L + 1) ** 2
    # node embedding: element embedding on l = 0 + edge-degree embedding (input_block.py:84-138)
    x = torch.zeros(N, S, C)
    x[:, 0] = sd["sphere_embedding.weight"][Z]
    pre = "edge_degree_embedding."
    scal = torch.cat([basis, sd[pre + "source_embedding.weight"][Z[src]], sd[pre + "target_embedding.weight"][Z[dst]]], dim=1)
    m0 = radial_mlp(sd, pre + "rad_func.", scal).reshape(-1, L + 1, C)              # m = 0 coefficients, l = 0..L
    red = lm_list(L, M)
    cols = torch.tensor([red.index((l, 0)) for l in range(L + 1)])
    rm = reduced_mask(L, M)
    back = (D.transpose(1, 2)[:, :, rm] * truncation_rescale(L, M)[None, None, :])[:, :, cols]   # [E, S, L+1]
    x = x + torch.zeros(N, S, C).index_add_(0, dst, torch.bmm(back, m0)) / AVG_DEGREE
    for i in range(hp["num_layers"]):
        p = f"blocks.{i}."
        x = x + attention_block(sd, p + "ga.", norm_sh(sd, p + "norm_1.", x, L), Z, basis, src, dst, D, hp, g_red, C)
        x = x + feed_forward(sd, p + "ffn.", norm_sh(sd, p + "norm_2.", x, L), hp, g_red)
    x = norm_sh(sd, "norm.", x, L)
    f1 = attention_block(sd, "force_block.", x, Z, basis, src, dst, D, hp, g_red, 1)[:, 1:4, 0]
    f2 = attention_block(sd, "force_block2.", x, Z, basis, src, dst, D, hp, g_red, 1)[:, 1:4, 0]
    return f1, f2
