"""TEST INFRASTRUCTURE — CPU restatement (plain PyTorch fp32) of AdsorbDiff's sampling hot path.

This file is the *oracle* for the HIP implementation.  It is a from-scratch
restatement of the reference algorithm, NOT a copy of the reference code; each
function cites the reference lines it follows.  It is pinned against the real
reference by ``oracle/make_golden.py`` (which imports /root/reference in the
build container and asserts equality / closeness function by function) and by
the committed fixtures under ``tests/golden/`` (``tests/test_oracle_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``adsorbdiff_amd``) never does.

Conventions (same as the reference):
  * edge_index[0] = source / neighbour j, edge_index[1] = target / centre i
    (utils/utils.py:728); messages flow j -> i and are summed at i.
  * graph uses row-vector lattice convention ``shift @ cell`` (utils.py:529);
    the stepper's COM wrap uses ``cell @ f`` (denoising_torch.py:298-310).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

F = torch.nn.functional


# --------------------------------------------------------------------------- graph
def cell_repeats(cell: torch.Tensor, radius: float, pbc=(True, True, True)) -> List[int]:
    """Number of periodic images needed per lattice direction, max over the batch.
    Reference: utils/utils.py:634-662."""
    a1, a2, a3 = cell[:, 0], cell[:, 1], cell[:, 2]
    c23 = torch.cross(a2, a3, dim=-1)
    vol = torch.sum(a1 * c23, dim=-1, keepdim=True)
    reps = []
    for k, cr in enumerate((c23, torch.cross(a3, a1, dim=-1), torch.cross(a1, a2, dim=-1))):
        if pbc[k]:
            inv_d = torch.norm(cr / vol, p=2, dim=-1)
            reps.append(int(torch.ceil(radius * inv_d).max().item()))
        else:
            reps.append(0)
    return reps


def shift_table(reps: List[int]) -> torch.Tensor:
    """Lexicographic list of integer lattice shifts (a,b,c), a slowest.  utils.py:665-669."""
    axes = [torch.arange(-r, r + 1, dtype=torch.float32) for r in reps]
    return torch.cartesian_prod(*axes).reshape(-1, 3)


def radius_graph_pbc(
    pos: torch.Tensor,
    cell: torch.Tensor,
    natoms: torch.Tensor,
    radius: float,
    max_neighbors: int,
    pbc=(True, True, True),
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """All (centre i, neighbour j, lattice shift) with 1e-4 < d^2 <= radius^2, then the
    ``max_neighbors`` nearest per centre (strict).  Candidates are ordered by
    (i, j, shift index); survivors keep that order.  Ties at the K-th place are broken by
    candidate order here; the reference's ``torch.sort`` (utils.py:806) is not stable, so
    for *exact* ties at the boundary its choice is implementation-defined.
    Reference: utils/utils.py:556-730 (radius_graph_pbc) + 733-853 (get_max_neighbors_mask)."""
    B = natoms.shape[0]
    reps = cell_repeats(cell, radius, pbc)
    shifts = shift_table(reps)  # [C,3]
    C = shifts.shape[0]
    # Cartesian offset of every shift in every image: cell^T @ shift  (utils.py:680-681)
    offs = torch.bmm(cell.transpose(1, 2), shifts.t().reshape(1, 3, C).expand(B, -1, -1))  # [B,3,C]
    r2 = radius * radius
    src_all, dst_all, sh_all, cnt_img = [], [], [], []
    start = 0
    for b in range(B):
        n = int(natoms[b])
        P = pos[start : start + n]
        p_i = P.reshape(n, 1, 3, 1)
        p_j = P.reshape(1, n, 3, 1) + offs[b].reshape(1, 1, 3, C)
        d2 = torch.sum((p_i - p_j) ** 2, dim=2)  # [n(i), n(j), C]
        ok = (d2 <= r2) & (d2 > 0.0001)
        d2f = d2.reshape(n, n * C)
        okf = ok.reshape(n, n * C)
        n_img = 0
        for i in range(n):
            cand = torch.nonzero(okf[i]).reshape(-1)  # ascending = (j, shift) order
            if cand.numel() > max_neighbors > 0:
                order = torch.sort(d2f[i, cand], stable=True).indices[:max_neighbors]
                cand = cand[torch.sort(order).values]
            j = torch.div(cand, C, rounding_mode="floor")
            c = cand % C
            src_all.append(j + start)
            dst_all.append(torch.full_like(j, i + start))
            sh_all.append(shifts[c])
            n_img += int(cand.numel())
        cnt_img.append(n_img)
        start += n
    edge_index = torch.stack([torch.cat(src_all), torch.cat(dst_all)])
    return edge_index, torch.cat(sh_all), torch.tensor(cnt_img, dtype=torch.long)


def pbc_distances(pos, edge_index, cell, cell_offsets, neighbors):
    """v = pos[j] - pos[i] + shift @ cell ; d = |v| ; drop d == 0.  utils/utils.py:513-553."""
    j, i = edge_index
    cell_e = torch.repeat_interleave(cell, neighbors, dim=0)
    off = torch.bmm(cell_offsets.float().reshape(-1, 1, 3), cell_e.float()).reshape(-1, 3)
    v = pos[j] - pos[i] + off
    d = v.norm(dim=-1)
    keep = torch.nonzero(d != 0).reshape(-1)
    return edge_index[:, keep], d[keep], v[keep], off[keep]


def symmetrize_edges(edge_index, cell_offsets, neighbors, dist, unit_vec):
    """Keep j<i edges (or same-atom edges whose shift is lexicographically negative), append
    their reversals (same d, negated unit vector and shift), order per image as
    [kept..., reversed...].  Reference: painn_denoising.py:262-327 (non-symmetric branch)
    with repeat_blocks(sizes=n_kept, repeats=2, repeat_inc=E_kept) (:304-309)."""
    j, i = edge_index
    s = cell_offsets
    earlier = (s[:, 0] < 0) | ((s[:, 0] == 0) & (s[:, 1] < 0)) | ((s[:, 0] == 0) & (s[:, 1] == 0) & (s[:, 2] < 0))
    keep = (j < i) | ((j == i) & earlier)
    img = torch.repeat_interleave(torch.arange(neighbors.shape[0]), neighbors)
    kept_idx = torch.nonzero(keep).reshape(-1)
    img_k = img[kept_idx]
    n_kept = torch.bincount(img_k, minlength=neighbors.shape[0])
    E_k = kept_idx.numel()
    # per image: first its kept edges, then the same edges reversed
    order = []
    start = 0
    for b in range(neighbors.shape[0]):
        nb = int(n_kept[b])
        blk = torch.arange(start, start + nb)
        order.append(blk)
        order.append(blk + E_k)
        start += nb
    order = torch.cat(order) if order else torch.zeros(0, dtype=torch.long)
    jk, ik = j[kept_idx], i[kept_idx]
    ei_cat = torch.stack([torch.cat([jk, ik]), torch.cat([ik, jk])])
    sk = s[kept_idx]
    dk, uk = dist[kept_idx], unit_vec[kept_idx]
    return (
        ei_cat[:, order],
        torch.cat([sk, -sk])[order],
        2 * n_kept,
        torch.cat([dk, dk])[order],
        torch.cat([uk, -uk])[order],
    )


def generate_graph_values(pos, cell, natoms, cutoff: float, max_neighbors: int):
    """Full graph pipeline of PaiNN.forward.  Reference: models/base.py:33-123 +
    painn_denoising.py:353-400.  Returns (edge_index[2,E], neighbors[B], dist[E], unit_vec[E,3])."""
    ei, sh, nb = radius_graph_pbc(pos, cell, natoms, cutoff, max_neighbors)
    ei, d, v, _ = pbc_distances(pos, ei, cell, sh, nb)
    # (d == 0 edges were dropped; shifts must follow)  painn_denoising.py:366-368 clamp:
    d = d.clone()
    d[torch.isclose(d, torch.tensor(0.0), atol=1e-3)] = 1.0e-3
    u = v / d[:, None]
    if torch.any(nb == 0):
        raise ValueError("An image has no neighbors")
    # NOTE: shifts of dropped (d==0) edges: the reference keeps `cell_offsets` unfiltered
    # (base.py:82-94 returns the input cell_offsets) — d==0 cannot occur after the d^2>1e-4
    # filter, so the two stay aligned.
    ei, sh, nb, d, u = symmetrize_edges(ei, sh, nb, d, u)
    return ei, nb, d, u


# --------------------------------------------------------------------------- model pieces
def radial_basis(d: torch.Tensor, cutoff: float, num_rbf: int = 128, p: int = 5) -> torch.Tensor:
    """env(d/rc) * exp(coeff * (d/rc - mu_k)^2).  radial_basis.py:18-43,64-82,235-244."""
    x = d * (1 / cutoff)
    pf = float(p)
    a, b, c = -(pf + 1) * (pf + 2) / 2, pf * (pf + 2), -pf * (pf + 1) / 2
    env = 1 + a * x**pf + b * x ** (pf + 1) + c * x ** (pf + 2)
    env = torch.where(x < 1, env, torch.zeros_like(x))
    mu = torch.linspace(0.0, 1.0, num_rbf)
    coeff = -0.5 / (1.0 / (num_rbf - 1)) ** 2
    return env[:, None] * torch.exp(coeff * torch.pow(x[:, None] - mu[None, :], 2))


def ssilu(x):
    """ScaledSiLU: silu(x) * (1/0.6).  gemnet_oc/layers/base_layers.py:65-72."""
    return F.silu(x) * (1 / 0.6)


def message_layer(sd, pre, x, vec, edge_index, rbf, unit_vec, H):
    """PaiNNMessage.forward/message/aggregate.  painn_denoising.py:530-567."""
    h = F.layer_norm(x, (H,), sd[pre + "x_layernorm.weight"], sd[pre + "x_layernorm.bias"])
    h = ssilu(F.linear(h, sd[pre + "x_proj.0.weight"], sd[pre + "x_proj.0.bias"]))
    xh = F.linear(h, sd[pre + "x_proj.2.weight"], sd[pre + "x_proj.2.bias"])
    rbfh = F.linear(rbf, sd[pre + "rbf_proj.weight"], sd[pre + "rbf_proj.bias"])
    src, dst = edge_index
    g = xh[src] * rbfh
    m_x, g2, g3 = g[:, :H], g[:, H : 2 * H], g[:, 2 * H :]
    g2 = g2 * (1 / math.sqrt(3.0))
    m_v = vec[src] * g2[:, None, :] + g3[:, None, :] * unit_vec[:, :, None]
    m_v = m_v * (1 / math.sqrt(H))
    dx = torch.zeros_like(x).index_add_(0, dst, m_x)
    dvec = torch.zeros_like(vec).index_add_(0, dst, m_v)
    return dx, dvec


def update_layer(sd, pre, x, vec, H):
    """PaiNNUpdate.forward.  painn_denoising.py:601-623."""
    vv = F.linear(vec, sd[pre + "vec_proj.weight"])
    v1, v2 = vv[..., :H], vv[..., H:]
    dot = (v1 * v2).sum(dim=1) * (1 / math.sqrt(H))
    vn = torch.sqrt(torch.sum(v2**2, dim=-2) + 1e-8)
    h = ssilu(F.linear(torch.cat([x, vn], dim=-1), sd[pre + "xvec_proj.0.weight"], sd[pre + "xvec_proj.0.bias"]))
    h = F.linear(h, sd[pre + "xvec_proj.2.weight"], sd[pre + "xvec_proj.2.bias"])
    h1, h2, h3 = h[:, :H], h[:, H : 2 * H], h[:, 2 * H :]
    dx = (h1 + h2 * dot) * (1 / math.sqrt(2.0))
    dvec = h3[:, None, :] * v1
    return dx, dvec


def gated_block(sd, pre, x, v, out_channels):
    """GatedEquivariantBlock.forward.  painn_denoising.py:688-697."""
    vec1 = torch.norm(F.linear(v, sd[pre + "vec1_proj.weight"]), dim=-2)
    vec2 = F.linear(v, sd[pre + "vec2_proj.weight"])
    h = ssilu(F.linear(torch.cat([x, vec1], dim=-1), sd[pre + "update_net.0.weight"], sd[pre + "update_net.0.bias"]))
    h = F.linear(h, sd[pre + "update_net.2.weight"], sd[pre + "update_net.2.bias"])
    xo, g = h[:, :out_channels], h[:, out_channels:]
    return ssilu(xo), g[:, None, :] * vec2


def output_head(sd, pre, x, vec, H):
    """PaiNNOutput.forward: two gated blocks then squeeze.  painn_denoising.py:647-650."""
    x, vec = gated_block(sd, pre + "output_network.0.", x, vec, H // 2)
    x, vec = gated_block(sd, pre + "output_network.1.", x, vec, 1)
    return vec.squeeze()


def painn_forward(
    sd: Dict[str, torch.Tensor],
    pos,
    atomic_numbers,
    cell,
    natoms,
    *,
    hidden_channels: int = 512,
    num_layers: int = 6,
    num_rbf: int = 128,
    cutoff: float = 12.0,
    max_neighbors: int = 50,
    scale_factors: Optional[List[float]] = None,
    so3_denoising: bool = True,
    graph=None,
    capture: Optional[dict] = None,
):
    """PaiNN.forward for the denoiser (painn_denoising.py:402-481).  ``sd`` is a reference
    ``state_dict``.  ``tag_based_Z`` is a no-op in the reference (operator precedence,
    :160-166) and is therefore absent here.  ``scale_factors[i]`` multiplies x after update
    layer i (:451); pass 1.0 for unfitted factors."""
    H = hidden_channels
    z = atomic_numbers.long()
    if graph is None:
        graph = generate_graph_values(pos, cell, natoms, cutoff, max_neighbors)
    edge_index, _, dist, unit_vec = graph
    rbf = radial_basis(dist, cutoff, num_rbf)
    x = sd["atom_emb.embeddings.weight"][z - 1]
    vec = torch.zeros(x.shape[0], 3, H)
    if scale_factors is None:
        scale_factors = [float(sd.get("upd_out_scalar_scale_%d.scale_factor" % i, 0.0)) or 1.0 for i in range(num_layers)]
    if capture is not None:
        capture["rbf"] = rbf
        capture["layers"] = []
    for i in range(num_layers):
        dx, dvec = message_layer(sd, "message_layers.%d." % i, x, vec, edge_index, rbf, unit_vec, H)
        if capture is not None:
            capture["layers"].append({"msg_dx": dx, "msg_dvec": dvec})
        x = (x + dx) * (1 / math.sqrt(2.0))
        vec = vec + dvec
        dx, dvec = update_layer(sd, "update_layers.%d." % i, x, vec, H)
        x = x + dx
        vec = vec + dvec
        x = x * scale_factors[i]
        if capture is not None:
            capture["layers"][-1].update({"x": x, "vec": vec})
    f1 = output_head(sd, "out_forces.", x, vec, H)
    if not so3_denoising:
        return f1
    f2 = output_head(sd, "out_forces2.", x, vec, H)
    return f1, f2


# --------------------------------------------------------------------------- stepper
def axis_angle_to_matrix(aa: torch.Tensor) -> torch.Tensor:
    """axis-angle -> unit quaternion (series below 1e-6) -> 3x3.  utils/rot_utils.py:18-98."""
    ang = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = 0.5 * ang
    small = ang.abs() < 1e-6
    k = torch.where(small, 0.5 - (ang * ang) / 48, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    q = torch.cat([torch.cos(half), aa * k], dim=-1)
    r, i, j, kk = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + kk * kk),
            two_s * (i * j - kk * r),
            two_s * (i * kk + j * r),
            two_s * (i * j + kk * r),
            1 - two_s * (i * i + kk * kk),
            two_s * (j * kk - i * r),
            two_s * (i * kk - j * r),
            two_s * (j * kk + i * r),
            1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(q.shape[:-1] + (3, 3))


def ads_mean(values: torch.Tensor, tags: torch.Tensor, batch: torch.Tensor, B: int) -> torch.Tensor:
    """Per-system mean over adsorbate (tag==2) atoms.  denoising_torch.py:460-467."""
    m = tags == 2
    tot = torch.zeros(B, values.shape[1], dtype=values.dtype).index_add_(0, batch[m], values[m])
    cnt = torch.zeros(B, dtype=values.dtype).index_add_(0, batch[m], torch.ones(int(m.sum()), dtype=values.dtype))
    return tot / cnt.clamp(min=1)[:, None]


def schedule_scalars(t_idx: int, num_steps: int, lo: float, hi: float, rlo: float, rhi: float):
    """(tr_g, rot_g, dt) for step t_idx with the dtypes the reference ends up with
    (float32 sigma from a float32 schedule; rot_g promoted to float64 by the 0-dim
    float64 ``torch.tensor(np.log(..))``).  denoising_torch.py:209-261."""
    sched = torch.tensor(np.linspace(1, 0, num_steps + 1)[:-1], dtype=torch.float32)
    t = sched[t_idx]
    tr_sigma = lo ** (1 - t) * hi**t
    rot_sigma = rlo ** (1 - t) * rhi**t
    tr_g = tr_sigma * (2 * np.log(hi / lo)) ** 0.5
    rot_g = 2 * rot_sigma * torch.sqrt(torch.tensor(np.log(rhi / rlo)))
    dt = sched[t_idx] - sched[t_idx + 1] if t_idx < num_steps - 1 else sched[t_idx]
    return tr_g, rot_g, dt


def initial_placement(pos, cell, tags, batch, noise):
    """Random xy placement of the adsorbate COM, z kept.  denoising_torch.py:215-232.
    ``noise`` = torch.rand(B,3) drawn by the caller from the CPU global generator."""
    B = cell.shape[0]
    com_noise = torch.einsum("bi,bij->bj", noise, cell.transpose(1, 2))
    m = tags == 2
    com0 = ads_mean(pos, tags, batch, B)
    com_noise[:, -1] = com0[:, -1]
    pos = pos.clone()
    pos[m] = pos[m] - com0[batch][m] + com_noise[batch][m]
    return pos


def reverse_step(pos, cell, tags, batch, f1, f2, fixed, t_idx, params, z_tr=None, z_rot=None):
    """One reverse step given the two model heads.  Returns (new_pos, dcom, drot, converged_flag).
    denoising_torch.py:237-353 (+ DiffTorchCalc.get_denoising_prediction :491-500)."""
    B = cell.shape[0]
    T = params["num_steps"]
    tr_g, rot_g, dt = schedule_scalars(
        t_idx, T, params["ads_std_low"], params["ads_std_high"], params["rot_std_low"], params["rot_std_high"]
    )
    f2 = f2.clone()
    f2[fixed == 1] = 0
    s_tr = ads_mean(f1, tags, batch, B)
    s_rot = ads_mean(f2, tags, batch, B)
    if params.get("ode", True):
        dcom = 0.5 * tr_g**2 * dt * s_tr
        drot = 0.5 * s_rot * dt * rot_g**2
    else:
        dcom = tr_g**2 * dt * s_tr + tr_g * np.sqrt(dt) * z_tr
        drot = s_rot * dt * rot_g**2 + rot_g * np.sqrt(dt) * z_rot
    com = ads_mean(pos, tags, batch, B)
    dcom[:, -1] = 0
    frac = torch.linalg.solve(cell, com + dcom)
    frac %= 1
    frac %= 1
    dcom = torch.einsum("bi,bij->bj", frac, cell.transpose(1, 2)) - com
    converged = bool(torch.allclose(dcom, torch.zeros_like(dcom), rtol=1e-3, atol=1e-3))
    R = axis_angle_to_matrix(drot.float()).float()  # [B,3,3]
    m = tags == 2
    bm = batch[m]
    rel = pos[m] - com[bm]
    new_ads = torch.einsum("nj,nij->ni", rel, R[bm]) + dcom[bm] + com[bm]
    new_pos = pos.clone()
    new_pos[m] = new_ads
    return new_pos, dcom, drot, converged


def reverse_sde_sampling_rot(
    pos,
    cell,
    tags,
    batch,
    fixed,
    model_fn: Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]],
    params: dict,
    noise: torch.Tensor,
    record: Optional[list] = None,
):
    """Whole reverse loop (ODE or SDE).  ``model_fn(pos) -> (f1, f2)``.  Early stop: cumulative
    count of steps whose wrapped COM update is allclose to 0 reaches 10 -> break *before*
    applying that step.  denoising_torch.py:198-367."""
    pos = initial_placement(pos, cell, tags, batch, noise)
    cvg = 0
    for t_idx in range(params["num_steps"]):
        f1, f2 = model_fn(pos)
        z_tr = z_rot = None
        if not params.get("ode", True):
            z_tr = torch.normal(mean=0, std=1, size=(cell.shape[0], 3))
            z_rot = torch.normal(mean=0, std=1, size=(cell.shape[0], 3))
        new_pos, dcom, drot, conv = reverse_step(pos, cell, tags, batch, f1, f2, fixed, t_idx, params, z_tr, z_rot)
        if conv:
            cvg += 1
            if cvg == 10:
                break
        pos = new_pos
        if record is not None:
            B = cell.shape[0]
            f2z = f2.clone()
            f2z[fixed == 1] = 0
            record.append({"dcom": dcom.clone(), "drot": drot.clone(), "pos": pos.clone(),
                           "s_tr": ads_mean(f1, tags, batch, B), "s_rot": ads_mean(f2z, tags, batch, B)})
    return pos
