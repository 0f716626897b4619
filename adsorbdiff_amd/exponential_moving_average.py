"""Exponential moving average of the trainable parameters.

Same object protocol as the reference's ``ExponentialMovingAverage``
(adsorbdiff/modules/exponential_moving_average.py:19-175): ``update / store / copy_to / restore /
state_dict / load_state_dict``, ``shadow_params`` in ``model.parameters()`` order restricted to
``requires_grad`` ones (that order is what a checkpoint's ``ema.shadow_params`` list follows,
base_trainer.py:456-533), decay optionally warmed up with ``(1+n)/(10+n)``.

Written for the device: the shadow update is one fused multi-tensor ``lerp`` (s += (1-d)(p-s)) instead of a Python
loop of two kernels per tensor.  Like the reference, ``copy_to``/``restore`` write through ``param.data`` — which does
NOT bump the parameters' autograd version counters; ``PaiNN.engine()`` therefore fingerprints the weight contents
instead of trusting ``_version`` (painn_denoising.py).
"""
from __future__ import annotations

import copy
import weakref
from typing import Iterable, List, Optional

import torch


class ExponentialMovingAverage:
    def __init__(self, parameters: Iterable[torch.nn.Parameter], decay: float, use_num_updates: bool = False) -> None:
        if not 0.0 <= decay <= 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay = decay
        self.num_updates: Optional[int] = 0 if use_num_updates else None
        trainable = [p for p in parameters if p.requires_grad]
        self.shadow_params: List[torch.Tensor] = [p.detach().clone() for p in trainable]
        self.collected_params: List[torch.Tensor] = []
        self._refs = [weakref.ref(p) for p in trainable]  # no strong reference to the model

    def _resolve(self, parameters) -> List[torch.nn.Parameter]:
        if parameters is not None:
            return [p for p in parameters if p.requires_grad]
        out = [r() for r in self._refs]
        if any(p is None for p in out):
            raise RuntimeError("a parameter tracked by this ExponentialMovingAverage no longer exists; "
                               "pass `parameters` explicitly or keep the model alive")
        return out

    @torch.no_grad()
    def update(self, parameters=None) -> None:
        params = self._resolve(parameters)
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        if self.shadow_params:
            torch._foreach_lerp_(self.shadow_params, [p.detach() for p in params], 1.0 - decay)

    def copy_to(self, parameters=None) -> None:
        for s, p in zip(self.shadow_params, self._resolve(parameters)):
            p.data.copy_(s.data)

    def store(self, parameters=None) -> None:
        self.collected_params = [p.clone() for p in self._resolve(parameters)]

    def restore(self, parameters=None) -> None:
        for c, p in zip(self.collected_params, self._resolve(parameters)):
            p.data.copy_(c.data)

    def state_dict(self) -> dict:
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": self.shadow_params,
                "collected_params": self.collected_params}

    def load_state_dict(self, state_dict: dict) -> None:
        state_dict = copy.deepcopy(state_dict)
        if not 0.0 <= state_dict["decay"] <= 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay = state_dict["decay"]
        self.num_updates = state_dict["num_updates"]
        if self.num_updates is not None and not isinstance(self.num_updates, int):
            raise ValueError("Invalid num_updates")
        shadow = state_dict["shadow_params"]
        if not isinstance(shadow, list) or not all(isinstance(t, torch.Tensor) for t in shadow):
            raise ValueError("shadow_params must be a list of tensors")
        if len(shadow) != len(self.shadow_params):
            raise ValueError("shadow_params has a different length than the tracked parameters")
        self.shadow_params = [t.to(s.device, s.dtype) for t, s in zip(shadow, self.shadow_params)]
        self.collected_params = list(state_dict.get("collected_params") or [])
