"""ppsci.metric.base.Metric (/root/reference/ppsci/metric/base.py:20-30): the base class user-defined metrics derive from
(examples/pipe/poiseuille_flow.py:265-300): `forward(output_dict, label_dict) -> {key: value}`."""
import torch


class Metric:
    def __init__(self, keep_batch: bool = False):
        self.keep_batch = keep_batch

    def __call__(self, output_dict, label_dict):
        with torch.no_grad():
            return self.forward(output_dict, label_dict)

    def forward(self, output_dict, label_dict):
        raise NotImplementedError(f"{type(self).__name__}.forward")
