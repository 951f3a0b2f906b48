"""ppsci.arch.ModelList (/root/reference/ppsci/arch/model_list.py:24-72): several networks that share the input
dict; the outputs are merged.  On the fused path every member keeps its own Taylor-mode kernels, and their
parameters are re-homed into ONE flat buffer (members' `flat_params` become slices of it), so that the optimizer,
the gradient all-reduce and the checkpoint see a single tensor as they do for one MLP."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from .base import Arch
from .mlp import MLP


class ModelList(Arch):
    def __init__(self, model_list: Tuple[Arch, ...]):
        super().__init__()
        model_list = tuple(model_list)
        from .piratenet import PirateNet

        for m in model_list:
            if not isinstance(m, (MLP, PirateNet)):
                raise NotImplementedError("ModelList members must be ppsci.arch.MLP / PirateNet on the fused HIP path")
        keys: List[str] = []
        for m in model_list:  # the reference keeps a set; a stable order is needed for the kernels' input arrays
            keys += [k for k in m.input_keys if k not in keys]
        self.input_keys = tuple(keys)
        seen = set()
        for m in model_list:
            dup = seen & set(m.output_keys)
            if dup:
                raise ValueError(f"output_keys of model from model_list should be unique,but got duplicate keys: {dup}")
            seen |= set(m.output_keys)
        self.output_keys = tuple(k for m in model_list for k in m.output_keys)
        self.model_list = list(model_list)
        pad = lambda n: (n + 63) // 64 * 64  # noqa: E731 -- every member starts on a 256-byte boundary (the
        # kernels use 16-byte vector loads of weights); the padding floats stay zero: their gradient is never written
        dev = model_list[0].flat_params.device
        self.reparam = any(m.reparam for m in model_list)
        self.flat_params = torch.zeros(sum(pad(m.flat_params.numel()) for m in model_list), dtype=torch.float32, device=dev)
        if self.reparam:
            # some member keeps factored / tied / activation parameters: the kernels read a second buffer in the
            # kernel layout (one slice per member, filled by materialize()); its gradient is pulled back per member
            self.kernel_params = torch.zeros(sum(pad(m.layout.n_params) for m in model_list), dtype=torch.float32, device=dev)
            self._grad_train = torch.zeros_like(self.flat_params)
        else:
            self.kernel_params = self.flat_params
        off = koff = 0
        for m in model_list:
            n, nk = m.flat_params.numel(), m.layout.n_params
            if m.reparam:
                m.rehome(self.flat_params[off:off + n], self._grad_train[off:off + n], self.kernel_params[koff:koff + nk])
            else:
                m.rehome(self.flat_params[off:off + n])
            m._train_offset = off
            m._param_offset = koff if self.reparam else off  # where the KERNELS find this member's parameters
            off += pad(n)
            koff += pad(nk)
        self.layout = None  # one layout per member: see compile.CompiledConstraint

    def materialize(self) -> torch.Tensor:
        if self.reparam:
            for m in self.model_list:
                if m.reparam:
                    m.materialize()
                else:  # same layout: a copy into its slice of the kernel buffer
                    n = m.flat_params.numel()
                    self.kernel_params[m._param_offset:m._param_offset + n].copy_(m.flat_params)
        return self.kernel_params

    def pull_back(self, grad_kernel: torch.Tensor) -> torch.Tensor:
        if not self.reparam:
            return grad_kernel
        for m in self.model_list:
            nk = m.layout.n_params
            gk = grad_kernel[m._param_offset:m._param_offset + nk]
            if m.reparam:
                m.pull_back(gk)  # writes its slice of self._grad_train
            else:
                self._grad_train[m._train_offset:m._train_offset + nk].copy_(gk)
        return self._grad_train

    def forward(self, x: Dict[str, object]) -> Dict[str, object]:
        y_all: Dict[str, object] = {}
        for m in self.model_list:
            y_all.update(m({k: x[k] for k in x}))
        if self._output_transform is not None:
            y_all = self._output_transform(x, y_all)
        return y_all

    def parameters(self) -> List[torch.Tensor]:
        return [p for m in self.model_list for p in m.parameters()]

    def named_parameters(self):
        return [(f"model_list.{i}.{n}", p) for i, m in enumerate(self.model_list) for n, p in m.named_parameters()]

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return dict(self.named_parameters())  # nn.LayerList naming: model_list.<i>.<param>

    def set_state_dict(self, state):
        missing, unexpected = [], []
        for i, m in enumerate(self.model_list):
            pre = f"model_list.{i}."
            mi, un = m.set_state_dict({k[len(pre):]: v for k, v in state.items() if k.startswith(pre)})
            missing += [pre + k for k in mi]
            unexpected += [pre + k for k in un]
        unexpected += [k for k in state if not k.startswith("model_list.")]
        return missing, unexpected
