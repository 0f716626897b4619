"""EquiformerV2 denoising score model — host-side mirror of the reference module (BASELINE config 4).

Drop-in for ``adsorbdiff.models.equiformer_v2.equiformer_v2_denoising.EquiformerV2S_OC20_DenoisingPos`` (reference:
models/equiformer_v2/equiformer_v2_denoising.py:28-318 on top of equiformer_v2_oc20.py:67-420): same constructor
signature, same parameter names and shapes in ``state_dict`` (reference checkpoints load; the reference's constant
buffers — S2 grid matrices, coefficient index tables, Gaussian offsets — are accepted and ignored by
``load_state_dict``: they are recomputed from the hyper-parameters), same ``forward(data) -> (forces[N,3],
forces2[N,3])``.  The sub-modules are *parameter containers only*; all arithmetic of ``forward`` runs in the HIP
library through the C ABI (``adf_eqv2_*`` in include/adsorbdiff_hip.h).  There is no CPU / eager fallback.

Supported configuration = what the repository ships (configs/denoising/eqv2_so3.yml): one resolution,
``norm_type='layer_norm_sh'``, ``attn_activation = ffn_activation = 'silu'``, ``use_s2_act_attn=False``,
``use_attn_renorm=True``, ``use_gate_act=False``, ``use_grid_mlp=True``, ``use_sep_s2_act=True``,
``use_atom_edge_embedding=True``, ``share_atom_edge_embedding=False``, ``use_m_share_rad=False``, Gaussian distance
expansion, ``grid_resolution`` given.  Anything else raises ``ValueError`` at construction.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
from torch import nn

_AVG_NUM_NODES = 77.81317
_AVG_DEGREE = 23.395238876342773  # equiformer_v2_denoising.py:22-25

# Empirical atomic radii in pm, Z = 0..100 (the table behind the reference's ``ATOMIC_RADII``,
# models/embeddings/atomic_radii.py; None = not tabulated -> NaN, as the reference builds it,
# equiformer_v2_denoising.py:165-169).  The reference subtracts these pm values from Angstrom distances (:209-213).
_ATOMIC_RADII_PM = [
    None, 25, 120, 145, 105, 85, 70, 65, 60, 50, 160, 180, 150, 125, 110, 100, 100, 100, 71, 220, 180, 160, 140, 135, 140,
    140, 140, 135, 135, 135, 135, 130, 125, 115, 115, 115, None, 235, 200, 180, 155, 145, 145, 135, 130, 135, 140, 160,
    155, 155, 145, 145, 140, 140, None, 260, 215, 195, 185, 185, 185, 185, 185, 185, 180, 175, 175, 175, 175, 175, 175,
    175, 155, 145, 135, 135, 130, 135, 135, 135, 150, 190, 180, 160, 190, None, None, None, 215, 195, 180, 180, 175, 175,
    175, 175, None, None, None, None, None,
]


class SiLU(nn.Module):
    """Parameter-free placeholder so that Sequential indices match the reference."""

    def forward(self, x):
        return torch.nn.functional.silu(x)


class RadialFunction(nn.Module):
    """Linear, LayerNorm, SiLU, ..., Linear (radial_function.py:11-32); weights U(-1/sqrt(in), 1/sqrt(in)), zero bias
    (equiformer_v2_oc20.py:585-595)."""

    def __init__(self, channels_list: List[int]) -> None:
        super().__init__()
        mods = []
        c_in = channels_list[0]
        for i in range(len(channels_list)):
            if i == 0:
                continue
            lin = nn.Linear(c_in, channels_list[i], bias=True)
            std = 1 / math.sqrt(c_in)
            nn.init.uniform_(lin.weight, -std, std)
            nn.init.constant_(lin.bias, 0)
            mods.append(lin)
            c_in = channels_list[i]
            if i == len(channels_list) - 1:
                break
            mods.append(nn.LayerNorm(channels_list[i]))
            mods.append(SiLU())
        self.net = nn.Sequential(*mods)


class SO3_LinearV2(nn.Module):
    """weight [lmax+1, out, in], bias [out] on l = 0 (so3.py:694-745)."""

    def __init__(self, in_features: int, out_features: int, lmax: int, normal: bool) -> None:
        super().__init__()
        self.in_features, self.out_features, self.lmax = in_features, out_features, lmax
        self.weight = nn.Parameter(torch.empty(lmax + 1, out_features, in_features))
        bound = 1 / math.sqrt(in_features)
        if normal:
            nn.init.normal_(self.weight, 0, bound)
        else:
            nn.init.uniform_(self.weight, -bound, bound)
        self.bias = nn.Parameter(torch.zeros(out_features))


def _linear(c_in: int, c_out: int, bias: bool, normal: bool, scale: float = 1.0) -> nn.Linear:
    lin = nn.Linear(c_in, c_out, bias=bias)
    if normal:
        nn.init.normal_(lin.weight, 0, 1 / math.sqrt(c_in))
    elif scale != 1.0:
        lin.weight.data.mul_(scale)
    if bias:
        nn.init.constant_(lin.bias, 0)
    return lin


class SO2_m_Convolution(nn.Module):
    """so2_ops.py:12-79: one bias-free map on the (l >= m) coefficients; 1/sqrt2 on the default initialisation."""

    def __init__(self, m: int, channels: int, out_channels: int, lmax: int, normal: bool) -> None:
        super().__init__()
        n = (lmax - m + 1) * channels
        self.fc = _linear(n, 2 * out_channels * (lmax - m + 1), False, normal, 1 / math.sqrt(2))


class SO2_Convolution(nn.Module):
    """so2_ops.py:82-262."""

    def __init__(self, channels: int, out_channels: int, lmax: int, mmax: int, normal: bool, extra_m0: int = 0,
                 rad_channels: Optional[List[int]] = None) -> None:
        super().__init__()
        n0 = (lmax + 1) * channels
        self.fc_m0 = _linear(n0, (lmax + 1) * out_channels + extra_m0, True, normal)
        self.so2_m_conv = nn.ModuleList([SO2_m_Convolution(m, channels, out_channels, lmax, normal) for m in range(1, mmax + 1)])
        if rad_channels is not None:
            n_in = n0 + sum((lmax - m + 1) * channels for m in range(1, mmax + 1))
            self.rad_func = RadialFunction(list(rad_channels) + [n_in])


class EquivariantLayerNormArraySphericalHarmonics(nn.Module):
    """layer_norm.py:129-250."""

    def __init__(self, lmax: int, num_channels: int) -> None:
        super().__init__()
        self.norm_l0 = nn.LayerNorm(num_channels, eps=1e-5)
        self.affine_weight = nn.Parameter(torch.ones(lmax, num_channels))


class SO2EquivariantGraphAttention(nn.Module):
    """transformer_block.py:38-224 (parameters only)."""

    def __init__(self, sphere_channels, hidden_channels, num_heads, alpha_channels, value_channels, output_channels, lmax,
                 mmax, max_num_elements, edge_channels_list, normal: bool) -> None:
        super().__init__()
        self.alpha_dot = nn.Parameter(torch.empty(num_heads, alpha_channels))
        std = 1.0 / math.sqrt(alpha_channels)
        nn.init.uniform_(self.alpha_dot, -std, std)
        self.source_embedding = nn.Embedding(max_num_elements, edge_channels_list[-1])
        self.target_embedding = nn.Embedding(max_num_elements, edge_channels_list[-1])
        nn.init.uniform_(self.source_embedding.weight.data, -0.001, 0.001)
        nn.init.uniform_(self.target_embedding.weight.data, -0.001, 0.001)
        rad = list(edge_channels_list)
        rad[0] = rad[0] + 2 * rad[-1]
        self.so2_conv_1 = SO2_Convolution(2 * sphere_channels, hidden_channels, lmax, mmax, normal,
                                          extra_m0=num_heads * alpha_channels + hidden_channels, rad_channels=rad)
        self.alpha_norm = nn.LayerNorm(alpha_channels)
        self.so2_conv_2 = SO2_Convolution(hidden_channels, num_heads * value_channels, lmax, mmax, normal)
        self.proj = SO3_LinearV2(num_heads * value_channels, output_channels, lmax, normal)


class FeedForwardNetwork(nn.Module):
    """transformer_block.py:375-471 with use_grid_mlp and use_sep_s2_act (parameters only)."""

    def __init__(self, sphere_channels, hidden_channels, output_channels, lmax, normal: bool) -> None:
        super().__init__()
        self.so3_linear_1 = SO3_LinearV2(sphere_channels, hidden_channels, lmax, normal)
        self.scalar_mlp = nn.Sequential(_linear(sphere_channels, hidden_channels, True, normal), SiLU())
        self.grid_mlp = nn.Sequential(_linear(hidden_channels, hidden_channels, False, normal), SiLU(),
                                      _linear(hidden_channels, hidden_channels, False, normal), SiLU(),
                                      _linear(hidden_channels, hidden_channels, False, normal))
        self.so3_linear_2 = SO3_LinearV2(hidden_channels, output_channels, lmax, normal)


class TransBlockV2(nn.Module):
    """transformer_block.py:534-728 (parameters only)."""

    def __init__(self, C, hidden, heads, alpha, value, ffn_hidden, lmax, mmax, max_num_elements, edge_channels_list, normal):
        super().__init__()
        self.norm_1 = EquivariantLayerNormArraySphericalHarmonics(lmax, C)
        self.ga = SO2EquivariantGraphAttention(C, hidden, heads, alpha, value, C, lmax, mmax, max_num_elements,
                                               edge_channels_list, normal)
        self.norm_2 = EquivariantLayerNormArraySphericalHarmonics(lmax, C)
        self.ffn = FeedForwardNetwork(C, ffn_hidden, C, lmax, normal)


class EdgeDegreeEmbedding(nn.Module):
    """input_block.py:11-83 (parameters only)."""

    def __init__(self, sphere_channels, lmax, max_num_elements, edge_channels_list) -> None:
        super().__init__()
        self.source_embedding = nn.Embedding(max_num_elements, edge_channels_list[-1])
        self.target_embedding = nn.Embedding(max_num_elements, edge_channels_list[-1])
        nn.init.uniform_(self.source_embedding.weight.data, -0.001, 0.001)
        nn.init.uniform_(self.target_embedding.weight.data, -0.001, 0.001)
        rad = list(edge_channels_list)
        rad[0] = rad[0] + 2 * rad[-1]
        self.rad_func = RadialFunction(rad + [(lmax + 1) * sphere_channels])


class EquiformerV2S_OC20_DenoisingPos(nn.Module):
    """See module docstring.  ``num_atoms, bond_feat_dim, num_targets`` are accepted and ignored like the reference."""

    NUM_GAUSSIANS = 600  # equiformer_v2_oc20.py:251-262: fixed, whatever ``num_distance_basis`` says

    def __init__(
        self,
        num_atoms=None, bond_feat_dim=None, num_targets=None,
        use_pbc=True, regress_forces=True, otf_graph=True, max_neighbors=500, max_radius=5.0, max_num_elements=110,
        num_layers=12, sphere_channels=128, attn_hidden_channels=128, num_heads=8, attn_alpha_channels=32,
        attn_value_channels=16, ffn_hidden_channels=512, norm_type="rms_norm_sh", lmax_list=[6], mmax_list=[2],
        grid_resolution=None, num_sphere_samples=128, edge_channels=128, use_atom_edge_embedding=True,
        share_atom_edge_embedding=False, use_m_share_rad=False, distance_function="gaussian", num_distance_basis=512,
        attn_activation="scaled_silu", use_s2_act_attn=False, use_attn_renorm=True, ffn_activation="scaled_silu",
        use_gate_act=False, use_grid_mlp=False, use_sep_s2_act=True, alpha_drop=0.1, drop_path_rate=0.05, proj_drop=0.0,
        weight_init="normal", enforce_max_neighbors_strictly=True, so3_denoising=False, FOR_denoising=False,
        energy_encoding=None, sampling=False,
    ) -> None:
        super().__init__()
        bad = []
        if len(lmax_list) != 1 or len(mmax_list) != 1: bad.append("one resolution (len(lmax_list) == 1)")
        if norm_type != "layer_norm_sh": bad.append("norm_type='layer_norm_sh'")
        if attn_activation != "silu" or ffn_activation != "silu": bad.append("attn_activation = ffn_activation = 'silu'")
        if use_s2_act_attn or not use_attn_renorm or use_gate_act or not use_grid_mlp or not use_sep_s2_act:
            bad.append("use_s2_act_attn=False, use_attn_renorm=True, use_gate_act=False, use_grid_mlp=True, use_sep_s2_act=True")
        if not use_atom_edge_embedding or share_atom_edge_embedding or use_m_share_rad:
            bad.append("use_atom_edge_embedding=True, share_atom_edge_embedding=False, use_m_share_rad=False")
        if distance_function != "gaussian": bad.append("distance_function='gaussian'")
        if grid_resolution is None: bad.append("an explicit grid_resolution")
        if not (use_pbc and otf_graph and regress_forces and enforce_max_neighbors_strictly):
            bad.append("use_pbc = otf_graph = regress_forces = enforce_max_neighbors_strictly = True")
        if not FOR_denoising: bad.append("FOR_denoising=True (two force blocks)")
        if energy_encoding is not None:
            bad.append("energy_encoding=None (the reference's conditional EquiformerV2 only runs under CUDA autocast, "
                       "equiformer_v2_denoising.py:263)")
        if weight_init not in ("normal", "uniform"): bad.append("weight_init in {'normal', 'uniform'}")
        if bad:
            raise ValueError("the HIP EquiformerV2 path implements the shipped configuration only; needs " + "; ".join(bad))
        self.use_pbc, self.regress_forces, self.otf_graph = use_pbc, regress_forces, otf_graph
        self.direct_forces = True
        self.max_neighbors, self.max_radius, self.cutoff = max_neighbors, max_radius, max_radius
        self.max_num_elements, self.num_layers, self.sphere_channels = max_num_elements, num_layers, sphere_channels
        self.attn_hidden_channels, self.num_heads = attn_hidden_channels, num_heads
        self.attn_alpha_channels, self.attn_value_channels = attn_alpha_channels, attn_value_channels
        self.ffn_hidden_channels, self.norm_type = ffn_hidden_channels, norm_type
        self.lmax_list, self.mmax_list, self.grid_resolution = list(lmax_list), list(mmax_list), grid_resolution
        self.edge_channels, self.num_distance_basis = edge_channels, num_distance_basis
        self.weight_init = weight_init
        self.avg_num_nodes, self.avg_degree = _AVG_NUM_NODES, _AVG_DEGREE
        self.so3_denoising, self.FOR_denoising, self.sampling = so3_denoising, FOR_denoising, sampling
        self.enforce_max_neighbors_strictly = enforce_max_neighbors_strictly
        lmax, mmax = self.lmax_list[0], self.mmax_list[0]
        normal = weight_init == "normal"
        ecl = [self.NUM_GAUSSIANS, edge_channels, edge_channels]
        self.edge_channels_list = ecl

        self.sphere_embedding = nn.Embedding(max_num_elements, sphere_channels)
        self.edge_degree_embedding = EdgeDegreeEmbedding(sphere_channels, lmax, max_num_elements, ecl)
        self.blocks = nn.ModuleList([
            TransBlockV2(sphere_channels, attn_hidden_channels, num_heads, attn_alpha_channels, attn_value_channels,
                         ffn_hidden_channels, lmax, mmax, max_num_elements, ecl, normal)
            for _ in range(num_layers)])
        self.norm = EquivariantLayerNormArraySphericalHarmonics(lmax, sphere_channels)
        # present in every checkpoint, unused by the denoiser's outputs (equiformer_v2_denoising.py:300-318)
        self.energy_block = FeedForwardNetwork(sphere_channels, ffn_hidden_channels, 1, lmax, normal)
        self.force_block = SO2EquivariantGraphAttention(sphere_channels, attn_hidden_channels, num_heads,
                                                        attn_alpha_channels, attn_value_channels, 1, lmax, mmax,
                                                        max_num_elements, ecl, normal)
        self.force_block2 = SO2EquivariantGraphAttention(sphere_channels, attn_hidden_channels, num_heads,
                                                         attn_alpha_channels, attn_value_channels, 1, lmax, mmax,
                                                         max_num_elements, ecl, normal)
        radii = torch.tensor([float("nan") if v is None else float(v) for v in _ATOMIC_RADII_PM])
        self.atom_radii = nn.Parameter(radii, requires_grad=False)  # in pm, like the reference (:165-169)
        self._engine = None
        self._engine_key = None

    # ------------------------------------------------------------------ API
    @property
    def num_params(self) -> int:
        return sum(p.numel() for p in self.parameters())

    def no_weight_decay(self) -> set:
        """Reference: equiformer_v2_oc20.py:597-621: biases and norm weights of Linear / SO3_LinearV2 / LayerNorm / the
        equivariant norm."""
        out = []
        for mname, mod in self.named_modules():
            if isinstance(mod, (nn.Linear, SO3_LinearV2, nn.LayerNorm, EquivariantLayerNormArraySphericalHarmonics)):
                for pname, _ in mod.named_parameters(recurse=False):
                    if isinstance(mod, (nn.Linear, SO3_LinearV2)) and "weight" in pname:
                        continue
                    out.append(mname + "." + pname)
        return set(out)

    # constant buffers of the reference's modules (functions of the hyper-parameters only; rebuilt by the engine):
    # S2 grid matrices (so3.py:566-599), CoefficientMapping tables (so3.py:22-115), SO3_LinearV2.expand_index
    # (so3.py:694-745), the equivariant norm's balance_degree_weight (layer_norm.py), GaussianSmearing.offset
    _CONST_BUFFER_SUFFIXES = ("to_grid_mat", "from_grid_mat", "expand_index", "balance_degree_weight",
                              "distance_expansion.offset", "l_harmonic", "m_harmonic", "m_complex", "res_size", "m_size",
                              "to_m")

    def load_state_dict(self, state_dict, strict: bool = True):
        """Reference checkpoints also carry constant buffers (S2 grid matrices, index tables, Gaussian offsets): they are
        functions of the hyper-parameters and are rebuilt here, so exactly THOSE keys are ignored.  Under ``strict`` any
        other unexpected key (a checkpoint of another configuration: more blocks, ``energy_embedding.*`` ...), any missing
        parameter and any shape mismatch raises, like ``nn.Module.load_state_dict``.  ``atom_radii`` (a frozen Parameter
        in the reference, equiformer_v2_denoising.py:165-169) is loaded when present and may be absent: it is a constant
        table."""
        own = self.state_dict()
        const = {k for k in state_dict if k not in own and k.endswith(self._CONST_BUFFER_SUFFIXES)}
        unexpected = sorted(k for k in state_dict if k not in own and k not in const)
        missing = sorted(k for k in own if k not in state_dict and k != "atom_radii")
        mismatched = sorted(k for k in own if k in state_dict and tuple(state_dict[k].shape) != tuple(own[k].shape))
        if strict and (unexpected or missing or mismatched):
            def head(v):
                return f"{v[:6]}{'...' if len(v) > 6 else ''}"
            raise RuntimeError("Error(s) in loading state_dict for EquiformerV2S_OC20_DenoisingPos: "
                               + (f"unexpected keys {head(unexpected)}; " if unexpected else "")
                               + (f"missing keys {head(missing)}; " if missing else "")
                               + (f"size mismatch for {head(mismatched)}" if mismatched else ""))
        kept = {k: v for k, v in state_dict.items() if k in own and k not in mismatched}
        res = super().load_state_dict(kept, strict=False)
        from torch.nn.modules.module import _IncompatibleKeys

        return _IncompatibleKeys([k for k in res.missing_keys if k != "atom_radii"] + mismatched, unexpected)

    def engine(self, device=None):
        from .eqv2_engine import EqV2Engine

        if device is None:
            device = self.sphere_embedding.weight.device
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self._engine is not None and self._engine.device != device:
            self._engine.close()
            self._engine = None
        version = self._weights_version()
        if self._engine is None:
            self._engine = EqV2Engine(self, device)
            self._engine_key = version
        elif self._engine_key != version:
            self._engine.bind_weights()
            self._engine_key = version
        return self._engine

    def _weights_version(self):
        """Key of the weight images the engine derives (transposed first radial layers, fp16 hi/lo splits):
        (data_ptr, _version) per tensor plus a content fingerprint that also catches writes through ``param.data`` —
        what the reference's EMA copy_to / restore do (modules/exponential_moving_average.py:113,147)."""
        tensors = list(self.parameters())
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        flat = [t.detach().reshape(-1) for t in tensors if t.is_cuda and t.dtype == torch.float32 and t.numel() > 0]
        if not flat:
            return key
        with torch.no_grad():
            bits = torch.cat(flat).view(torch.int32)
            fp = int(bits.sum(dtype=torch.int64).item()) ^ int((bits[::7].sum(dtype=torch.int64) * 31).item())
        return key + (fp,)

    def forward(self, data):
        """data: pos [N,3] f32, atomic_numbers [N], batch [N], natoms [B], cell [B,3,3] -> (forces [N,3], forces2 [N,3]):
        the l = 1 coefficients (m = -1, 0, 1) of the two force blocks (equiformer_v2_denoising.py:307-318)."""
        eng = self.engine(data.pos.device)
        return eng.forward(data)
