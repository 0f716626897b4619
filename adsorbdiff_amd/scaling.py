"""ScaleFactor: forward-time scalar multiply + scale-file loader.

Mirrors the reference's ``adsorbdiff.modules.scaling`` for the part the
sampling path needs (reference: adsorbdiff/modules/scaling/scale_factor.py:29-172,
compat.py:52-77): a 0-dim ``scale_factor`` buffer that multiplies only when it
has been fitted (value != 0).  The interactive fitter (fit.py) is out of scope.
"""
from __future__ import annotations

import json
import logging
from pathlib import Path
from typing import Dict, Optional, Union

import torch
from torch import nn


class ScaleFactor(nn.Module):
    scale_factor: torch.Tensor

    def __init__(self, name: Optional[str] = None) -> None:
        super().__init__()
        self.name = name
        # a frozen 0-dim Parameter like the reference (scale_factor.py:47-49), so
        # parameter counts and checkpoint keys agree
        self.scale_factor = nn.Parameter(torch.tensor(0.0), requires_grad=False)

    @property
    def fitted(self) -> bool:
        return bool((self.scale_factor != 0.0).item())

    @torch.no_grad()
    def set_(self, scale: Union[float, torch.Tensor]) -> None:
        self.scale_factor.fill_(float(scale))

    @torch.no_grad()
    def reset_(self) -> None:
        self.scale_factor.zero_()

    def forward(self, x: torch.Tensor, *, ref: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.fitted:
            x = x * self.scale_factor
        return x


def _load_scale_dict(scale_file: Union[str, Path, Dict[str, float]]):
    if isinstance(scale_file, dict):
        return scale_file
    path = Path(scale_file)
    if not path.exists():
        raise ValueError(f"Scale file '{path}' does not exist.")
    if path.suffix == ".pt":
        scale_dict = torch.load(path, map_location="cpu")
    elif path.suffix == ".json":
        with open(path) as f:
            scale_dict = json.load(f)
        if isinstance(scale_dict, dict):
            scale_dict.pop("comment", None)
    else:
        raise ValueError(f"Unsupported scale file extension '{path.suffix}'")
    return scale_dict


def load_scales_compat(module: nn.Module, scale_file: Union[str, Path, Dict[str, float], None]) -> None:
    """Set every ``ScaleFactor`` sub-module named in ``scale_file`` (name -> value)."""
    if not scale_file:
        return
    scale_dict = _load_scale_dict(scale_file)
    if not scale_dict:
        logging.warning("No scale factors found in the scale file.")
        return
    factors = {name: m for name, m in module.named_modules() if isinstance(m, ScaleFactor)}
    for name, scale in scale_dict.items():
        if name not in factors:
            logging.warning(f"Scale factor '{name}' in the scale file has no matching module; ignored.")
            continue
        value = scale.item() if torch.is_tensor(scale) else float(scale)
        factors[name].set_(value)


def ensure_fitted(module: nn.Module, warn: bool = False) -> None:
    """Reference: adsorbdiff/modules/scaling/util.py:8-23."""
    unfitted = [name for name, m in module.named_modules() if isinstance(m, ScaleFactor) and not m.fitted]
    if unfitted:
        msg = "Unfitted scale factors: " + ", ".join(unfitted)
        if warn:
            logging.warning(msg)
        else:
            raise ValueError(msg)


# values of configs/scaling_factors/painn_nb6_scaling_factors.pt in the reference tree.
PAINN_NB6_SCALE_FACTORS = {
    "upd_out_scalar_scale_0": 1.0364354848861694,
    "upd_out_scalar_scale_1": 0.8951448202133179,
    "upd_out_scalar_scale_2": 0.8934778571128845,
    "upd_out_scalar_scale_3": 0.8899308443069458,
    "upd_out_scalar_scale_4": 0.8886106610298157,
    "upd_out_scalar_scale_5": 0.8822302222251892,
}
