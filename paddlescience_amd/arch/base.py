"""ppsci.arch.Arch (/root/reference/ppsci/arch/base.py:28-279): dict-in / dict-out networks with
named inputs and outputs, optional input / output transforms, freeze / unfreeze."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch


class Arch:
    input_keys: Tuple[str, ...]
    output_keys: Tuple[str, ...]

    def __init__(self):
        self._input_transform: Optional[Callable] = None
        self._output_transform: Optional[Callable] = None
        self.training = True

    # ---- reference API
    def forward(self, *args, **kwargs):
        raise NotImplementedError("Arch.forward is not implemented")

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    @property
    def num_params(self) -> int:
        return int(sum(int(np.prod(p.shape)) for p in self.parameters()))

    @staticmethod
    def concat_to_tensor(data_dict: Dict[str, torch.Tensor], keys: Tuple[str, ...], axis=-1):  # base.py:78-112
        if len(keys) == 1:
            return data_dict[keys[0]]
        return torch.cat([data_dict[k] for k in keys], dim=axis)

    @staticmethod
    def split_to_dict(data_tensor: torch.Tensor, keys: Tuple[str, ...], axis=-1):  # base.py:114-148
        if len(keys) == 1:
            return {keys[0]: data_tensor}
        parts = torch.split(data_tensor, 1, dim=axis)
        return {k: parts[i] for i, k in enumerate(keys)}

    def register_input_transform(self, transform: Callable):  # base.py:150-183
        self._input_transform = transform

    def register_output_transform(self, transform: Callable):  # base.py:185-218
        self._output_transform = transform

    def parameters(self) -> List[torch.Tensor]:
        return []

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {}

    def set_state_dict(self, state: Dict[str, torch.Tensor]):
        raise NotImplementedError

    def freeze(self):  # base.py:220-244
        self.training = False
        self._frozen = True

    def unfreeze(self):
        self.training = True
        self._frozen = False

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    def __str__(self):
        return f"{self.__class__.__name__}(input_keys: {self.input_keys}, output_keys: {self.output_keys}, " \
               f"num_params: {self.num_params})"
