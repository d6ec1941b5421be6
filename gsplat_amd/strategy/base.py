"""Strategy interface (reference ``gsplat/strategy/base.py:24-65``)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Union

import torch


@dataclass
class Strategy:
    """Callbacks around ``loss.backward()``. Convention shared by all strategies: ``params`` maps names to
    ``nn.Parameter`` whose FIRST dimension indexes Gaussians; every trainable parameter has its own optimizer with
    exactly one parameter group."""

    def check_sanity(self, params: Union[Dict[str, torch.nn.Parameter], torch.nn.ParameterDict],
                     optimizers: Dict[str, torch.optim.Optimizer]):
        trainable = {name for name, p in params.items() if p.requires_grad}
        assert trainable == set(optimizers.keys()), (
            f"trainable parameters and optimizers must have the same keys, got {trainable} and {set(optimizers.keys())}")
        for name, opt in optimizers.items():
            assert len(opt.param_groups) == 1, f"optimizer '{name}' must have exactly one param_group"

    def step_pre_backward(self, *args, **kwargs):
        pass

    def step_post_backward(self, *args, **kwargs):
        pass
